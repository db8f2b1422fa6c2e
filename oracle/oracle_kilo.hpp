// ORACLE — TEST INFRASTRUCTURE ONLY (see smallmat.hpp header).  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
//
// CPU restatement of the estimation part of legkilo/src/core/slam/KILO.cc
// (:86-399) and of preprocess/state_initial.hpp:34-118.  ROS / PCL / YAML types
// are replaced by the PODs of include/legkilo_hip.h (lk_point, lk_imu, lk_kin_imu).
#pragma once
#include <deque>
#include <memory>

#include "../include/legkilo_hip.h"
#include "oracle_eskf.hpp"
#include "oracle_voxel_map.hpp"

namespace lko {

struct ResidualRow {  // one ObsShared row + validity (KILO.cc:189-210)
    bool valid;
    double h[6];
    double z;
    double R;
    int layer;
};

class KILO {
   public:
    explicit KILO(const lk_config& cfg);  // KILO.cc:25-84 (values arrive pre-parsed)

    // KILO.cc:108-233.  cloud_down_body: n x lk_point (only idx_i..idx_j read);
    // cloud_down_world: n x 4 floats (x,y,z,intensity)
    bool predictUpdatePoint(double current_time, size_t idx_i, size_t idx_j, const lk_point* cloud_down_body,
                            float* cloud_down_world, size_t& success_pts_size_out);
    bool predictUpdateImu(const lk_imu& imu);             // KILO.cc:235-258
    bool predictUpdateKinImu(const lk_kin_imu& kin_imu);  // KILO.cc:260-314

    // First-frame branch of KILO::process (KILO.cc:332-353): StateInitial + BuildVoxelMap.
    void firstFrame(const lk_point* cloud_raw, size_t n, double end_time, const lk_imu* imus, size_t n_imu,
                    const lk_kin_imu* kins, size_t n_kin);
    // Bucket loop of KILO::process (KILO.cc:367-396) on an already time-sorted cloud
    // (the std::sort at :370 is unstable; callers sort once and feed both sides the same order).
    void processSorted(const lk_point* pts, size_t n, double begin_time, std::deque<lk_imu>& imus,
                       std::deque<lk_kin_imu>& kin_imus, float* cloud_down_world, size_t& success_pts_size_out,
                       uint32_t* n_buckets, uint32_t* n_updates);
    // Steps 2-3a of predictUpdatePoint only (KILO.cc:122-210) with the current state: per-point rows.
    void residualsOnly(const float* xyz_body, size_t n, std::vector<ResidualRow>& rows);

    std::unique_ptr<ESKF> eskf_;
    std::unique_ptr<VoxelMapManager> map_manager_;
    bool imu_mode_only_ = true;
    double gravity_ = 9.81;
    double acc_norm_ = 1.0;
    double last_state_predict_time_ = 0.0;
    double last_state_update_time_ = 0.0;
    Mat3 ext_rot_ = Mat3::Identity();
    Vec3 ext_t_ = Vec3::Zero();
    bool map_insert_enabled_ = true;  // false = frozen-map batch replay (config 5); not a reference switch

    // KILO.cc:122-183 for one point; returns is_success (public for the diagnostics of oracle_capi.cc)
    bool matchPoint(const lk_point& cur_pt, pointWithVar& cur_pt_var, PointToPlane& single_ptpl, float* world_xyzi);
    void rowFromPtpl(const PointToPlane& p, double* h6, double& z, double& R);  // KILO.cc:195-209
};

}  // namespace lko
