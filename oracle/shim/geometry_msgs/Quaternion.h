#ifndef LK_SHIM_GEOMETRY_QUATERNION
#define LK_SHIM_GEOMETRY_QUATERNION
namespace geometry_msgs {
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
};
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
}  // namespace geometry_msgs
#endif
