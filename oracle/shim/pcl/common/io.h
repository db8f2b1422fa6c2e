// oracle/shim — nothing of pcl/common/io.h is used by the compiled reference sources
#include "../point_cloud.h"
#include "../point_types.h"
