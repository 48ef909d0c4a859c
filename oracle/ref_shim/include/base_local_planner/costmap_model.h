#include "../../shim_ros.h"
