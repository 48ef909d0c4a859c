#include "../../../shim_boost_graph.h"
