// oracle/shim — TEST INFRASTRUCTURE ONLY: the PCL point type the reference names (common/pcl_types.h:11), fields only.
#ifndef LK_SHIM_PCL_POINT_TYPES
#define LK_SHIM_PCL_POINT_TYPES
// pcl/pcl_macros.h (PCL 1.8): the truncated constant is part of the reference's arithmetic (voxel_map.cc:27)
#ifndef DEG2RAD
#define DEG2RAD(x) ((x)*0.017453293)
#endif
#ifndef RAD2DEG
#define RAD2DEG(x) ((x)*57.29578)
#endif
namespace pcl {
struct PointXYZINormal {
    float x = 0, y = 0, z = 0;
    float normal_x = 0, normal_y = 0, normal_z = 0;
    float intensity = 0, curvature = 0;
};
struct PointXYZI {
    float x = 0, y = 0, z = 0, intensity = 0;
};
}  // namespace pcl
#endif
