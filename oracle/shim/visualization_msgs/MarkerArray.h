#ifndef LK_SHIM_VIS_MARKER_ARRAY
#define LK_SHIM_VIS_MARKER_ARRAY
#include <vector>
#include "Marker.h"
namespace visualization_msgs {
struct MarkerArray {
    std::vector<Marker> markers;
};
}  // namespace visualization_msgs
#endif
