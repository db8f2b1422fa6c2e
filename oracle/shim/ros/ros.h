// oracle/shim — TEST INFRASTRUCTURE ONLY: inert shells of the roscpp types named by voxel_map.h / voxel_map.cc
// (visualisation code that is compiled but never called by the parity tests).
#ifndef LK_SHIM_ROS_H
#define LK_SHIM_ROS_H
#include <string>
namespace ros {
struct Time {
    double t = 0;
    Time() {}
    explicit Time(double s) : t(s) {}
    static Time now() { return Time(); }
    double toSec() const { return t; }
};
struct Duration {
    double d = 0;
    Duration() {}
    explicit Duration(double s) : d(s) {}
};
struct Rate {
    explicit Rate(double) {}
    void sleep() {}
};
struct Publisher {
    template <class M>
    void publish(const M&) const {}
};
}  // namespace ros
#endif
