// tests/integration/refhip_capi.cc — TEST INFRASTRUCTURE ONLY.
//
// The same C API (lkk_*) that oracle/ref_kilo_capi.cc puts over the reference's own legkilo::KILO, here over the reference's KILO
// AFTER tests/integration/kilo_hip.ed: KILO.h / KILO.cc as a maintainer would patch them, core/slam/eskf.h and
// core/slam/voxel_map.h forwarding to leg-kilo_amd/host/legkilo_host_eigen.hpp, eskf.cc and voxel_map.cc NOT in the build, linked
// against liblegkilo_hip.so.  tests/test_integration.py drives this library with the inputs of tests/golden/ref_kilo_small.npz /
// ref_kilo_config4.npz - made by the unpatched build of the same KILO::process - and expects the golden match counts and states:
// the reference's own orchestrator (YAML parsing, first-frame initialisation through state_initial.hpp, voxel grid, time sort, the bucket
// loop with its message queues) on top of the C-ABI.
//
// KILO keeps its modules private; this translation unit (and only this one) includes KILO.h with the access specifier neutralised.
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

using namespace legkilo;

struct lkk_handle {
    std::unique_ptr<KILO> k;
};
static thread_local std::string g_err;

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
template <class F>
static int guarded(F&& f) {
    try {
        return f();
    } catch (const std::exception& e) {
        g_err = e.what();
        return -100;
    }
}

extern "C" {

const char* lkk_last_error() { return g_err.c_str(); }

// KILO(config_file): the reference parses its own flat YAML (KILO.cc:25-83); the patched initializeFromYaml creates the device handle.
// Without a gfx950 device lk_create fails (LK_ERR_NO_DEVICE) and so does this: there is no CPU fallback behind the patched KILO.
lkk_handle* lkk_create(const char* yaml_path) {
    try {
        lkk_handle* h = new lkk_handle;
        h->k = std::make_unique<KILO>(std::string(yaml_path));
        h->k->path_->eskf().cov().setZero();
        h->k->path_->eskf().Q().setZero();
        return h;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}
void lkk_destroy(lkk_handle* h) { delete h; }
int lkk_imu_mode_only(lkk_handle* h) { return h->k->imu_mode_only_ ? 1 : 0; }
// 1: KILO::process runs its bucket loop as ONE device call per scan (KiloPath::processSorted); 0 (default): the reference's loop, one
// predictUpdatePoint / predictUpdateImu / predictUpdateKinImu call per bucket / message
int lkk_set_fused_scan(lkk_handle* h, int on) {
    h->k->hip_fused_scan_ = on != 0;
    return 0;
}

int lkk_set_state(lkk_handle* h, const double* x, const double* P900) {
    return guarded([&] {
        ESKF& e = h->k->path_->eskf();
        if (x) {
            State s;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) s.rot_(i, j) = x[3 * i + j];
            Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
            for (int k = 0; k < 9; ++k)
                for (int c = 0; c < 3; ++c) (*v[k])[c] = x[9 + 3 * k + c];
            e.setState(s);
        }
        if (P900) {
            auto P = e.cov();   // write-back proxy
            for (int i = 0; i < 30; ++i)
                for (int j = 0; j < 30; ++j) P(i, j) = P900[30 * i + j];
        }
        return 0;
    });
}
int lkk_get_state(lkk_handle* h, double* x, double* P900) {
    return guarded([&] {
        const ESKF& e = h->k->path_->eskf();
        if (x) {
            const State s = e.state();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) x[3 * i + j] = s.rot_(i, j);
            const Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
            for (int k = 0; k < 9; ++k)
                for (int c = 0; c < 3; ++c) x[9 + 3 * k + c] = (*v[k])[c];
        }
        if (P900) {
            const StateCov P = e.cov();
            for (int i = 0; i < 30; ++i)
                for (int j = 0; j < 30; ++j) P900[30 * i + j] = P(i, j);
        }
        return 0;
    });
}
int lkk_get_Q(lkk_handle* h, double* Q900) {
    return guarded([&] {
        const StateQ Q = static_cast<const ESKF&>(h->k->path_->eskf()).Q();
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) Q900[30 * i + j] = Q(i, j);
        return 0;
    });
}
int lkk_init_process_cov_q(lkk_handle* h) {
    return guarded([&] {
        h->k->path_->eskf().initProcessCovQ();
        return 0;
    });
}
int lkk_set_times(lkk_handle* h, double last_predict_t, double last_update_t) {
    return guarded([&] {
        h->k->last_state_predict_time_ = last_predict_t;
        h->k->last_state_update_time_ = last_update_t;
        h->k->path_->setTimes(last_predict_t, last_update_t);
        return 0;
    });
}
int lkk_get_times(lkk_handle* h, double* last_predict_t, double* last_update_t) {
    return guarded([&] {
        h->k->path_->getTimes(last_predict_t, last_update_t);
        return 0;
    });
}
int lkk_set_acc_norm(lkk_handle* h, double a) {
    return guarded([&] {
        h->k->acc_norm_ = a;
        h->k->path_->setAccNorm(a);
        return 0;
    });
}
double lkk_get_acc_norm(lkk_handle* h) { return h->k->path_->accNorm(); }

// the map of a first frame whose state was set by the caller: BuildVoxelMap exactly as called at KILO.cc:339, first-frame flag cleared
int lkk_map_build(lkk_handle* h, const float* xyz_world, const float* xyz_body, size_t n) {
    return guarded([&] {
        auto& m = h->k->path_->map_manager();
        m.feats_down_world_.reset(new PointCloudType());
        m.feats_down_body_.reset(new PointCloudType());
        for (size_t i = 0; i < n; ++i) {
            PointType pw, pb;
            pw.x = xyz_world[3 * i], pw.y = xyz_world[3 * i + 1], pw.z = xyz_world[3 * i + 2];
            pb.x = xyz_body[3 * i], pb.y = xyz_body[3 * i + 1], pb.z = xyz_body[3 * i + 2];
            m.feats_down_world_->push_back(pw);
            m.feats_down_body_->push_back(pb);
        }
        ESKF& e = h->k->path_->eskf();
        m.BuildVoxelMap(e.getRot(), e.getRotCov(), e.getPosCov());
        h->k->init_flag_ = false;
        return 0;
    });
}
int lkk_map_export(lkk_handle* h, void* blob, size_t* bytes) { return lk_map_export(h->k->path_->device().h(), blob, bytes); }
int lkk_map_stats(lkk_handle* h, uint32_t* n_roots) { return lk_map_stats(h->k->path_->device().h(), n_roots, nullptr, nullptr); }

// KILO::process on the FIRST frame (init_flag_ still set): StateInitial + cloudLidarToWorld + BuildVoxelMap (KILO.cc:332-353)
int lkk_first_frame(lkk_handle* h, const lk_point* raw, size_t n, double end_time, const lk_imu* imus, size_t n_imu,
                    const lk_kin_imu* kins, size_t n_kin) {
    if (!h->k->init_flag_) return -5;
    return guarded([&] {
        common::MeasGroup g = to_meas(raw, n, end_time, end_time, imus, n_imu, kins, n_kin);
        CloudPtr body, world;
        size_t ok = 0;
        return h->k->process(g, body, world, ok) ? 0 : -3;
    });
}

// KILO::process on a later scan: down-sampling (identity on the pre-filtered input), time sort, bucket loop
int lkk_process_scan(lkk_handle* h, const lk_point* pts, size_t n, double t_begin, const lk_imu* imus, size_t n_imu,
                     const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out, int /*with_sort*/) {
    if (h->k->init_flag_) return -5;
    return guarded([&] {
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
            const State s = static_cast<const ESKF&>(h->k->path_->eskf()).state();
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) out->rot[3 * i + j] = s.rot_(i, j);
                out->pos[i] = s.pos_[i];
                out->vel[i] = s.vel_[i];
            }
            out->n_effect = ok;
        }
        return 0;
    });
}

}  // extern "C"
