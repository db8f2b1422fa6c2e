// oracle/shim — TEST INFRASTRUCTURE ONLY: field-only shell of visualization_msgs/Marker (voxel_map.cc:475-500)
#ifndef LK_SHIM_VIS_MARKER
#define LK_SHIM_VIS_MARKER
#include <string>
#include "../geometry_msgs/Quaternion.h"
#include "../ros/ros.h"
namespace std_msgs {
struct Header {
    unsigned int seq = 0;
    ros::Time stamp;
    std::string frame_id;
};
struct ColorRGBA {
    float r = 0, g = 0, b = 0, a = 0;
};
}  // namespace std_msgs
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3 };
    enum { ADD = 0, MODIFY = 0, DELETE = 2 };
    std_msgs::Header header;
    std::string ns;
    int id = 0, type = 0, action = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    ros::Duration lifetime;
};
}  // namespace visualization_msgs
#endif
