// oracle/ref_decode_capi.cc — TEST INFRASTRUCTURE ONLY.
// C API (lkd_*) over the REFERENCE's own legkilo::LidarProcessing (preprocess/lidar_processing.h/.cc:25-108), compiled
// unmodified from /root/reference by `make ref` against oracle/shim (PointCloud2 shell, restated pcl::fromROSMsg).
// The caller hands over a PointCloud2 payload in the layout of include/legkilo_hip.h's lk_cloud_layout; the message
// carries the fields x, y, z and the sensor's time field, as the C-ABI's lk_decode_scan does.
#include <cstring>
#include <memory>

#include "../include/legkilo_hip.h"
#include "preprocess/lidar_processing.h"

using namespace legkilo;

extern "C" int lkd_decode(const void* msg_data, size_t n_points, const lk_cloud_layout* layout, double time_scale, int filter_num,
                          float blind, double header_stamp, lk_point* out, size_t* n_out, double* begin_time, double* end_time) {
    if (n_points == 0) return -3;
    LidarProcessing::Config cfg;
    cfg.blind_ = blind;
    cfg.filter_num_ = filter_num;
    cfg.time_scale_ = time_scale;
    cfg.lidar_type_ = static_cast<common::LidarType>(layout->lidar_type);
    LidarProcessing lp(cfg);
    auto msg = std::make_shared<sensor_msgs::PointCloud2>();
    msg->header.stamp = ros::Time(header_stamp);
    msg->width = (uint32_t)n_points;
    msg->height = 1;
    msg->point_step = layout->point_step;
    msg->row_step = layout->point_step * (uint32_t)n_points;
    msg->data.assign((const uint8_t*)msg_data, (const uint8_t*)msg_data + n_points * layout->point_step);
    using sensor_msgs::PointField;
    auto add = [&](const char* name, uint32_t off, uint8_t type) {
        PointField f;
        f.name = name, f.offset = off, f.datatype = type, f.count = 1;
        msg->fields.push_back(f);
    };
    add("x", layout->off_x, PointField::FLOAT32);
    add("y", layout->off_y, PointField::FLOAT32);
    add("z", layout->off_z, PointField::FLOAT32);
    switch (layout->lidar_type) {
        case 1: add("time", layout->off_time, PointField::FLOAT32); break;
        case 2: add("t", layout->off_time, PointField::UINT32); break;
        case 3: add("timestamp", layout->off_time, PointField::FLOAT64); break;
        default: return -3;
    }
    common::LidarScan scan;
    lp.processing(msg, scan);
    const size_t n = scan.cloud_->points.size();
    for (size_t i = 0; i < n; ++i) {
        const PointType& p = scan.cloud_->points[i];
        out[i].x = p.x, out[i].y = p.y, out[i].z = p.z, out[i].curvature = p.curvature;
    }
    *n_out = n;
    *begin_time = scan.lidar_begin_time_;
    *end_time = scan.lidar_end_time_;
    return 0;
}
