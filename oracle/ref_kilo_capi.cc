// oracle/ref_kilo_capi.cc — TEST INFRASTRUCTURE ONLY.
//
// A C API (lkk_*) over the REFERENCE's own legkilo::KILO (KILO.h, KILO.cc: first-frame initialisation, the bucket loop
// of KILO::process :316-399, predictUpdatePoint :108-233, predictUpdateImu / predictUpdateKinImu :235-314), compiled
// unmodified from /root/reference by `make ref` against the stand-in headers of oracle/shim.  The entry points mirror
// the lko_* functions of oracle_capi.cc that drive whole scans, so tests/test_reference_pin.py replays the same scans
// through the reference's KILO::process and through the oracle and compares states, match counts and maps.
//
// Two things about the build are part of the contract (oracle/shim/kilo_prefix.h, pcl/filters/voxel_grid.h):
//   * KILO.cc's std::sort by time (unstable; the reference's own result is defined only up to a permutation inside a
//     bucket) is compiled as its stable instance;
//   * pcl::VoxelGrid is a restatement; the tests feed clouds that are already down-sampled, on which it is the identity.
// KILO keeps its modules private; this translation unit (and only this one) includes KILO.h with the access specifier
// neutralised, which does not change the class layout.
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <vector>

#include "core/slam/eskf.h"
#include "core/slam/voxel_map.h"
#include "preprocess/state_initial.hpp"
#define private public
#include "core/slam/KILO.h"
#undef private
#include "export_blob.hpp"

using namespace legkilo;

struct lkk_handle {
    std::unique_ptr<KILO> k;
};

static sensor_msgs::ImuPtr to_imu(const lk_imu& m) {
    sensor_msgs::ImuPtr p(new sensor_msgs::Imu());
    p->header.stamp = ros::Time(m.stamp);
    p->linear_acceleration.x = m.acc[0], p->linear_acceleration.y = m.acc[1], p->linear_acceleration.z = m.acc[2];
    p->angular_velocity.x = m.gyr[0], p->angular_velocity.y = m.gyr[1], p->angular_velocity.z = m.gyr[2];
    return p;
}
static common::KinImuMeas to_kin(const lk_kin_imu& m) {
    common::KinImuMeas k;
    k.time_stamp_ = m.time_stamp;
    for (int l = 0; l < 4; ++l) {
        for (int c = 0; c < 3; ++c) k.foot_pos_[l][c] = m.foot_pos[l][c], k.foot_vel_[l][c] = m.foot_vel[l][c];
        k.contact_[l] = m.contact[l] != 0;
    }
    for (int c = 0; c < 3; ++c) k.acc_[c] = m.acc[c], k.gyr_[c] = m.gyr[c];
    return k;
}
static common::MeasGroup to_meas(const lk_point* pts, size_t n, double begin, double end, const lk_imu* imus, size_t n_imu,
                                 const lk_kin_imu* kins, size_t n_kin) {
    common::MeasGroup g;
    g.lidar_scan_.lidar_begin_time_ = begin;
    g.lidar_scan_.lidar_end_time_ = end;
    g.lidar_scan_.cloud_.reset(new PointCloudType());
    g.lidar_scan_.cloud_->points.resize(n);
    for (size_t i = 0; i < n; ++i) {
        PointType& p = g.lidar_scan_.cloud_->points[i];
        p.x = pts[i].x, p.y = pts[i].y, p.z = pts[i].z, p.curvature = pts[i].curvature;
    }
    for (size_t i = 0; i < n_imu; ++i) g.imus_.push_back(to_imu(imus[i]));
    for (size_t i = 0; i < n_kin; ++i) g.kin_imus_.push_back(to_kin(kins[i]));
    return g;
}

extern "C" {

// KILO(config_file): the reference parses its own flat YAML (KILO.cc:25-83)
lkk_handle* lkk_create(const char* yaml_path) {
    try {
        lkk_handle* h = new lkk_handle;
        h->k = std::make_unique<KILO>(std::string(yaml_path));
        h->k->eskf_->cov().setZero();
        h->k->eskf_->Q().setZero();
        return h;
    } catch (const std::exception&) {
        return nullptr;
    }
}
void lkk_destroy(lkk_handle* h) {
    if (!h) return;
    for (auto& kv : h->k->map_manager_->voxel_map_) delete kv.second;  // the reference never frees its trees
    delete h;
}
int lkk_imu_mode_only(lkk_handle* h) { return h->k->imu_mode_only_ ? 1 : 0; }

int lkk_set_state(lkk_handle* h, const double* x, const double* P900) {
    State& s = h->k->eskf_->state();
    if (x) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) s.rot_(i, j) = x[3 * i + j];
        Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
        for (int k = 0; k < 9; ++k)
            for (int c = 0; c < 3; ++c) (*v[k])[c] = x[9 + 3 * k + c];
    }
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) h->k->eskf_->cov()(i, j) = P900[30 * i + j];
    return 0;
}
int lkk_get_state(lkk_handle* h, double* x, double* P900) {
    const State& s = h->k->eskf_->state();
    if (x) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) x[3 * i + j] = s.rot_(i, j);
        const Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
        for (int k = 0; k < 9; ++k)
            for (int c = 0; c < 3; ++c) x[9 + 3 * k + c] = (*v[k])[c];
    }
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) P900[30 * i + j] = h->k->eskf_->cov()(i, j);
    return 0;
}
int lkk_get_Q(lkk_handle* h, double* Q900) {
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) Q900[30 * i + j] = h->k->eskf_->Q()(i, j);
    return 0;
}
int lkk_init_process_cov_q(lkk_handle* h) {
    h->k->eskf_->initProcessCovQ();
    return 0;
}
int lkk_set_times(lkk_handle* h, double last_predict_t, double last_update_t) {
    h->k->last_state_predict_time_ = last_predict_t;
    h->k->last_state_update_time_ = last_update_t;
    return 0;
}
int lkk_get_times(lkk_handle* h, double* last_predict_t, double* last_update_t) {
    *last_predict_t = h->k->last_state_predict_time_;
    *last_update_t = h->k->last_state_update_time_;
    return 0;
}
int lkk_set_acc_norm(lkk_handle* h, double a) {
    h->k->acc_norm_ = a;
    return 0;
}
double lkk_get_acc_norm(lkk_handle* h) { return h->k->acc_norm_; }

// the map of a first frame whose state was set by the caller (what scenes.first_frame does on every backend):
// BuildVoxelMap exactly as called at KILO.cc:339, then the first-frame flag is cleared
int lkk_map_build(lkk_handle* h, const float* xyz_world, const float* xyz_body, size_t n) {
    auto& m = *h->k->map_manager_;
    m.feats_down_world_.reset(new PointCloudType());
    m.feats_down_body_.reset(new PointCloudType());
    for (size_t i = 0; i < n; ++i) {
        PointType pw, pb;
        pw.x = xyz_world[3 * i], pw.y = xyz_world[3 * i + 1], pw.z = xyz_world[3 * i + 2];
        pb.x = xyz_body[3 * i], pb.y = xyz_body[3 * i + 1], pb.z = xyz_body[3 * i + 2];
        m.feats_down_world_->push_back(pw);
        m.feats_down_body_->push_back(pb);
    }
    m.BuildVoxelMap(h->k->eskf_->getRot(), h->k->eskf_->getRotCov(), h->k->eskf_->getPosCov());
    h->k->init_flag_ = false;
    return 0;
}
int lkk_map_export(lkk_handle* h, void* blob, size_t* bytes) {
    auto& m = *h->k->map_manager_;
    return lkx::export_map<VoxelOctoTree, VoxelPlane, pointWithVar>(m.voxel_map_, m.config_setting_.max_voxel_size_,
                                                                   m.config_setting_.max_layer_,
                                                                   m.config_setting_.max_points_num_, blob, bytes);
}
int lkk_map_stats(lkk_handle* h, uint32_t* n_roots) {
    *n_roots = (uint32_t)h->k->map_manager_->voxel_map_.size();
    return 0;
}

// KILO::process on the FIRST frame (init_flag_ still set): StateInitial + cloudLidarToWorld + BuildVoxelMap (KILO.cc:332-353)
int lkk_first_frame(lkk_handle* h, const lk_point* raw, size_t n, double end_time, const lk_imu* imus, size_t n_imu,
                    const lk_kin_imu* kins, size_t n_kin) {
    if (!h->k->init_flag_) return -5;
    common::MeasGroup g = to_meas(raw, n, end_time, end_time, imus, n_imu, kins, n_kin);
    CloudPtr body, world;
    size_t ok = 0;
    return h->k->process(g, body, world, ok) ? 0 : -3;
}

// KILO::process on a later scan: down-sampling (identity on the pre-filtered input), time sort, bucket loop.
// xyz_world_out (n x 3) is in the order process() leaves the cloud in (= sorted by time, stable).
int lkk_process_scan(lkk_handle* h, const lk_point* pts, size_t n, double t_begin, const lk_imu* imus, size_t n_imu,
                     const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out, int /*with_sort*/) {
    if (h->k->init_flag_) return -5;
    double end = t_begin;
    for (size_t i = 0; i < n; ++i) end = std::max(end, t_begin + (double)pts[i].curvature);
    common::MeasGroup g = to_meas(pts, n, t_begin, end, imus, n_imu, kins, n_kin);
    CloudPtr body, world;
    size_t ok = 0;
    if (!h->k->process(g, body, world, ok)) return -3;
    if (body->size() != n) return -6;  // the input was not one point per down-sampling cell
    if (xyz_world_out)
        for (size_t i = 0; i < n; ++i) {
            xyz_world_out[3 * i] = world->points[i].x, xyz_world_out[3 * i + 1] = world->points[i].y;
            xyz_world_out[3 * i + 2] = world->points[i].z;
        }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        const State& s = h->k->eskf_->state();
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) out->rot[3 * i + j] = s.rot_(i, j);
            out->pos[i] = s.pos_[i];
            out->vel[i] = s.vel_[i];
        }
        out->n_effect = ok;
    }
    return 0;
}

}  // extern "C"
