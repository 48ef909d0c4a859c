#include "../../../../shim_g2o.h"
