// oracle/shim — TEST INFRASTRUCTURE ONLY: field-only shell of sensor_msgs/Imu (common/sensor_types.hpp:4,31)
#ifndef LK_SHIM_SENSOR_IMU
#define LK_SHIM_SENSOR_IMU
#include <memory>
#include "../visualization_msgs/Marker.h"
namespace sensor_msgs {
struct Imu {
    std_msgs::Header header;
    geometry_msgs::Quaternion orientation;
    geometry_msgs::Vector3 angular_velocity, linear_acceleration;
};
typedef std::shared_ptr<Imu> ImuPtr;
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
#endif
