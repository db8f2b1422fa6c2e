// oracle/shim — TEST INFRASTRUCTURE ONLY: restatement of the PCL pieces preprocess/lidar_processing.h uses
// (PCL 1.8 / pcl_conversions, third-party, absent here):
//   EIGEN_ALIGN16, PCL_ADD_POINT4D                  a 16-byte aligned x, y, z (+ padding) head of a point struct
//   POINT_CLOUD_REGISTER_POINT_STRUCT(T, (type, member, tag)...)   the field list of a custom point type
//   pcl::fromROSMsg(PointCloud2, PointCloud<T>)     per point, every REGISTERED field of T that the message carries
//                                                   under the same name and datatype is copied from its message
//                                                   offset; fields the message lacks stay value-initialised (PCL
//                                                   warns "Failed to find match for field" and does the same)
#ifndef LK_SHIM_PCL_CONVERSIONS
#define LK_SHIM_PCL_CONVERSIONS
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include "../pcl/point_cloud.h"
#include "../pcl/point_types.h"
#include "../sensor_msgs/PointCloud2.h"

#define EIGEN_ALIGN16 alignas(16)
#define PCL_ADD_POINT4D \
    float x = 0, y = 0, z = 0, data_pad_ = 1.0f;

namespace lk_shim {
struct FieldDesc {
    const char* name;
    size_t offset, size;
    uint8_t datatype;
};
template <class T>
constexpr uint8_t pcl_type_code() {
    using sensor_msgs::PointField;
    return std::is_same<T, int8_t>::value ? PointField::INT8 : std::is_same<T, uint8_t>::value ? PointField::UINT8
         : std::is_same<T, int16_t>::value ? PointField::INT16 : std::is_same<T, uint16_t>::value ? PointField::UINT16
         : std::is_same<T, int32_t>::value ? PointField::INT32 : std::is_same<T, uint32_t>::value ? PointField::UINT32
         : std::is_same<T, float>::value ? PointField::FLOAT32 : std::is_same<T, double>::value ? PointField::FLOAT64 : 0;
}
template <class P>
struct FieldList;  // specialised by POINT_CLOUD_REGISTER_POINT_STRUCT
}  // namespace lk_shim

// sequence iteration without Boost.Preprocessor: A and B call each other until the sequence is used up
#define LK_PP_CAT(a, b) LK_PP_CAT_I(a, b)
#define LK_PP_CAT_I(a, b) a##b
#define LK_REG_A(type, member, tag) LK_REG_FIELD(type, member) LK_REG_B
#define LK_REG_B(type, member, tag) LK_REG_FIELD(type, member) LK_REG_A
#define LK_REG_A_END
#define LK_REG_B_END
#define LK_REG_FIELD(type, member) v.push_back(::lk_shim::FieldDesc{#member, offsetof(P_, member), sizeof(type), ::lk_shim::pcl_type_code<type>()});
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, seq)                 \
    namespace lk_shim {                                              \
    template <>                                                      \
    struct FieldList<name> {                                         \
        static void get(std::vector<FieldDesc>& v) {                 \
            typedef name P_;                                         \
            LK_PP_CAT(LK_REG_A seq, _END)                            \
        }                                                            \
    };                                                               \
    }

namespace pcl {
template <class PointT>
void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointT>& cloud) {
    std::vector<lk_shim::FieldDesc> reg;
    lk_shim::FieldList<PointT>::get(reg);
    struct Map {
        size_t src, dst, size;
    };
    std::vector<Map> maps;
    for (const auto& f : reg)
        for (const auto& mf : msg.fields)
            if (mf.name == f.name && mf.datatype == f.datatype) {
                maps.push_back({mf.offset, f.offset, f.size});
                break;
            }
    const size_t n = (size_t)msg.width * msg.height;
    cloud.points.assign(n, PointT());
    cloud.width = msg.width, cloud.height = msg.height, cloud.is_dense = msg.is_dense;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* src = msg.data.data() + i * msg.point_step;
        uint8_t* dst = reinterpret_cast<uint8_t*>(&cloud.points[i]);
        for (const auto& m : maps) std::memcpy(dst + m.dst, src + m.src, m.size);
    }
}
}  // namespace pcl
#endif
