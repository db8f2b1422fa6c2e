// ORACLE — TEST INFRASTRUCTURE ONLY (see smallmat.hpp header).  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
//
// CPU restatement of legkilo/src/core/slam/voxel_map.{h,cc} (lines 22-427, the live
// part: the viz / sliding code at :429-594 is dead in the reference and not restated)
// and of the hash/key helpers in legkilo/src/common/eigen_types.hpp:52-95.
#pragma once
#include <array>
#include <cstdint>
#include <unordered_map>
#include <vector>

#include "smallmat.hpp"

namespace lko {

using Vec3i = std::array<int, 3>;

// eigen_types.hpp:79-82 (Teschner hash in int arithmetic; wraps; cast to size_t)
struct hash_vec3 {
    size_t operator()(const Vec3i& v) const {
        unsigned a = (unsigned)v[0] * 73856093u, b = (unsigned)v[1] * 471943u, c = (unsigned)v[2] * 83492791u;
        int x = (int)(a ^ b ^ c);  // two's-complement wrap of the int products
        return size_t(x % 10000000);
    }
};
// eigen_types.hpp:89-95
inline Vec3i voxelKeyFloor(const Vec3& pt, double voxel_size) {
    return Vec3i{(int)std::floor(pt[0] / voxel_size), (int)std::floor(pt[1] / voxel_size),
                 (int)std::floor(pt[2] / voxel_size)};
}

// voxel_map.h:41-57
struct VoxelMapConfig {
    double max_voxel_size_;
    int max_layer_;
    int max_iterations_;  // unused (voxel_map.h:44)
    std::vector<int> layer_init_num_;
    int max_points_num_;
    double planner_threshold_;
    double beam_err_;
    double dept_err_;
    double sigma_num_;
    // local map sliding, voxel_map.h:53-56 (loaded at KILO.cc:68-70; the reference never calls mapSliding itself)
    double sliding_thresh = 8;
    bool map_sliding_en = false;
    int half_map_size = 100;
};

// voxel_map.h:59-78
struct pointWithVar {
    Vec3 point_b = Vec3::Zero(), point_i = Vec3::Zero(), point_w = Vec3::Zero();
    Mat3 var_nostate = Mat3::Zero(), body_var = Mat3::Zero(), var = Mat3::Zero(), point_crossmat = Mat3::Zero();
    Vec3 normal = Vec3::Zero();
};

// voxel_map.h:80-94
struct PointToPlane {
    Vec3 point_b_, point_w_, normal_, center_;
    Mat3 point_crossmat_;
    Mat6 plane_var_;
    Mat3 body_cov_;
    int layer_;
    double d_;
    float dis_to_plane_;
};

// voxel_map.h:96-119
struct VoxelPlane {
    Vec3 center_ = Vec3::Zero(), normal_ = Vec3::Zero(), y_normal_ = Vec3::Zero(), x_normal_ = Vec3::Zero();
    Mat3 covariance_ = Mat3::Zero();
    Mat6 plane_var_ = Mat6::Zero();
    float radius_ = 0, min_eigen_value_ = 1, mid_eigen_value_ = 1, max_eigen_value_ = 1, d_ = 0;
    int points_size_ = 0;
    bool is_plane_ = false, is_init_ = false;
    int id_ = 0;
    bool is_update_ = false;
};

// voxel_map.cc:22-40
void calcBodyCov(Vec3& pb, const float range_inc, const float degree_inc, Mat3& cov);

// voxel_map.h:129-176
class VoxelOctoTree {
   public:
    std::vector<pointWithVar> temp_points_;
    VoxelPlane* plane_ptr_;
    int layer_;
    int octo_state_;
    VoxelOctoTree* leaves_[8];
    double voxel_center_[3];
    std::vector<int> layer_init_num_;
    float quater_length_;
    float planer_threshold_;
    int points_size_threshold_;
    int update_size_threshold_;
    int max_points_num_;
    int max_layer_;
    int new_points_;
    bool init_octo_;
    bool update_enable_;

    VoxelOctoTree(int max_layer, int layer, int points_size_threshold, int max_points_num, float planer_threshold);
    ~VoxelOctoTree();
    void init_plane(const std::vector<pointWithVar>& points, VoxelPlane* plane);  // voxel_map.cc:42-117
    void init_octo_tree();                                                          // voxel_map.cc:119-137
    void cut_octo_tree();                                                           // voxel_map.cc:139-183
    void UpdateOctoTree(const pointWithVar& pv);                                    // voxel_map.cc:185-241
};

// voxel_map.h:180-244 (live members only)
class VoxelMapManager {
   public:
    explicit VoxelMapManager(const VoxelMapConfig& c) : config_setting_(c) {
        extR_ = Mat3::Identity();
        extT_ = Vec3::Zero();
    }
    ~VoxelMapManager();
    VoxelMapConfig config_setting_;
    std::unordered_map<Vec3i, VoxelOctoTree*, hash_vec3> voxel_map_;
    Mat3 extR_;
    Vec3 extT_;
    // feats_down_body_ / feats_down_world_ as xyz float triples (PCL clouds in the reference)
    std::vector<float> feats_down_body_, feats_down_world_;

    void BuildVoxelMap(const Mat3 rot, const Mat3 rot_cov, const Mat3 pos_cov);  // voxel_map.cc:287-334
    void UpdateVoxelMap(const std::vector<pointWithVar>& input_points);          // voxel_map.cc:336-361
    void build_single_residual(pointWithVar& pv, const VoxelOctoTree* current_octo, const int current_layer,
                               bool& is_success, double& prob, PointToPlane& single_ptpl);  // voxel_map.cc:363-427

    // local map sliding.  position_last_ is a public member the caller sets (voxel_map.h:199; nothing in the
    // reference writes it), last_slide_position starts at the origin (voxel_map.h:201).
    Vec3 position_last_ = Vec3::Zero();
    Vec3 last_slide_position = Vec3::Zero();
    bool mapSliding();                                                              // voxel_map.cc:552-569
    int clearMemOutOfMap(int x_max, int x_min, int y_max, int y_min, int z_max, int z_min);  // voxel_map.cc:571-594

    // TEST DIAGNOSTIC (not in the reference): when set, build_single_residual records the smallest relative distance of
    // any gate it evaluates from its threshold - [0] the range gate (float, voxel_map.cc:379-381), [1] the sigma gate
    // (:388) - so that a parity test can show that a decision flip sits within rounding of a threshold.
    double* gate_margin_probe_ = nullptr;
};

extern int voxel_plane_id;  // voxel_map.h:39

}  // namespace lko
