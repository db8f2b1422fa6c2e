// legkilo_host.hpp — C++ mirror of the reference class surface that KILO consumes, implemented purely on
// the C-ABI of include/legkilo_hip.h (no HIP, no torch, no Eigen in this header).
//
//   legkilo::ESKF              <- legkilo/src/core/slam/eskf.h:46-109
//   legkilo::VoxelMapManager   <- legkilo/src/core/slam/voxel_map.h:180-244 (live members)
//   legkilo::KiloPath          <- KILO::predictUpdatePoint / predictUpdateImu / predictUpdateKinImu and the
//                                 bucket loop of KILO::process (KILO.cc:108-399)
//
// Same method names, argument meaning and (void / bool) error behaviour as the reference.  Differences forced
// by the state living in HBM: state()/cov()/Q() return COPIES (use setState/setCov/setQ to write back), and
// build_single_residual takes the voxel KEY where the reference takes a VoxelOctoTree* (a batch of points per call; a bucket of the path
// goes through BuildResidualList(), which never leaves the GPU).
// Configuration errors throw std::runtime_error like YamlHelper does (yaml_helper.hpp:42,50).
#pragma once
#include <algorithm>
#include <array>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "legkilo_hip.h"

// The mirror lives in namespace legkilo, like the classes it stands in for.  Inside the reference's own tree that namespace already
// holds the Eigen typedefs of common/eigen_types.hpp (Vec3D, Mat3D, ...): there legkilo_host_eigen.hpp includes this header under
// another name (LEGKILO_HOST_NAMESPACE = legkilo_hip) and puts Eigen-typed classes with the reference's names on top of it.
#ifndef LEGKILO_HOST_NAMESPACE
#define LEGKILO_HOST_NAMESPACE legkilo
#endif

namespace LEGKILO_HOST_NAMESPACE {

constexpr int DIM_STATE = LK_DIM_STATE;
using Vec3D = std::array<double, 3>;
struct Mat3D {
    double m[9];  // row-major
    double& operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
    static Mat3D Identity() { return Mat3D{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
};
using StateVec = std::array<double, DIM_STATE>;
struct StateCov {
    std::vector<double> d = std::vector<double>(DIM_STATE * DIM_STATE, 0.0);  // row-major
    double& operator()(int i, int j) { return d[i * DIM_STATE + j]; }
    double operator()(int i, int j) const { return d[i * DIM_STATE + j]; }
    Mat3D block3(int o) const {
        Mat3D b;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) b(i, j) = (*this)(o + i, o + j);
        return b;
    }
};
using StateF = StateCov;
using StateQ = StateCov;

// eskf.h:15-32
struct State {
    Mat3D rot_ = Mat3D::Identity();
    Vec3D pos_{}, vel_{}, ba_{}, bw_{}, grav_{{0.0, 0.0, -9.81}}, imu_a_{}, imu_w_{}, bv_{}, contact_{};
    void to_x36(double* x) const {
        std::memcpy(x, rot_.m, sizeof(rot_.m));
        const Vec3D* v[9] = {&pos_, &vel_, &ba_, &bw_, &grav_, &imu_a_, &imu_w_, &bv_, &contact_};
        for (int k = 0; k < 9; ++k)
            for (int c = 0; c < 3; ++c) x[9 + 3 * k + c] = (*v[k])[c];
    }
    void from_x36(const double* x) {
        std::memcpy(rot_.m, x, sizeof(rot_.m));
        Vec3D* v[9] = {&pos_, &vel_, &ba_, &bw_, &grav_, &imu_a_, &imu_w_, &bv_, &contact_};
        for (int k = 0; k < 9; ++k)
            for (int c = 0; c < 3; ++c) (*v[k])[c] = x[9 + 3 * k + c];
    }
};

// eskf.h:34-44; pt_h is N x 6 and ki_h is M x 30, both row-major
struct ObsShared {
    std::vector<double> pt_z, pt_h, pt_R, ki_z, ki_h, ki_R;
};

// voxel_map.h:41-57
struct VoxelMapConfig {
    double max_voxel_size_ = 0.5;
    int max_layer_ = 2;
    int max_iterations_ = 0;
    std::vector<int> layer_init_num_{5, 5, 5, 5, 5};
    int max_points_num_ = 50;
    double planner_threshold_ = 0.01, beam_err_ = 0.2, dept_err_ = 0.04, sigma_num_ = 3;
    bool is_pub_plane_map_ = false;
    double sliding_thresh = 8;      // loaded, never consulted by the reference (KILO.cc:68-70)
    bool map_sliding_en = false;
    int half_map_size = 100;
};

// voxel_map.h:59-78: the two fields UpdateVoxelMap reads
struct pointWithVar {
    Vec3D point_w{};
    Mat3D var{};
};

// PointToPlane (voxel_map.h:80-94): what build_single_residual writes into it and KILO.cc:160-210 reads of it
struct PointToPlane {
    Vec3D point_w_{}, normal_{}, center_{};
    int layer_ = -1;
    double d_ = 0.0;
    float dis_to_plane_ = 0.0f;
};

// PointType (pcl_types.h:11) fields the path touches
struct PointType {
    float x = 0, y = 0, z = 0, intensity = 0, curvature = 0;
};
using PointCloudType = std::vector<PointType>;

struct DeviceCaps {
    int device_id = 0;
    uint32_t n_slots = 1, max_roots = 1u << 18, max_nodes = 1u << 19, max_point_blocks = 1u << 18, max_scan_points = 1u << 17;
};

// One lk_handle shared by the ESKF and VoxelMapManager mirrors (they are two views of the same device state).
class Device {
   public:
    Device(const lk_config& cfg) {
        if (lk_create(&cfg, &h_) != LK_OK) throw std::runtime_error(std::string("lk_create: ") + lk_last_error(nullptr));
        cfg_ = cfg;
    }
    ~Device() { lk_destroy(h_); }
    Device(const Device&) = delete;
    Device& operator=(const Device&) = delete;
    lk_handle* h() const { return h_; }
    const lk_config& cfg() const { return cfg_; }
    void check(int rc) const {
        if (rc != LK_OK) throw std::runtime_error(std::string("liblegkilo_hip: ") + lk_last_error(h_));
    }

   private:
    lk_handle* h_ = nullptr;
    lk_config cfg_;
};

class ESKF {
   public:
    // eskf.h:49-65
    struct Config {
        double vel_process_cov, imu_acc_process_cov, imu_gyr_process_cov, contact_process_cov, acc_bias_process_cov,
            gyr_bias_process_cov, kin_bias_process_cov;
        double imu_acc_meas_noise, imu_acc_z_meas_noise, imu_gyr_meas_noise, kin_meas_noise, chd_meas_noise,
            contact_meas_noise, lidar_point_meas_ratio;
    };
    ESKF(const Config& config, std::shared_ptr<Device> dev) : config_(config), dev_(std::move(dev)) {}

    State state() const {
        double x[LK_STATE_DOUBLES];
        dev_->check(lk_get_state(dev_->h(), 0, x, nullptr));
        State s;
        s.from_x36(x);
        return s;
    }
    void setState(const State& s) {
        double x[LK_STATE_DOUBLES];
        s.to_x36(x);
        dev_->check(lk_set_state(dev_->h(), 0, x, nullptr));
    }
    Mat3D getRot() const { return state().rot_; }
    Vec3D getPos() const { return state().pos_; }
    Vec3D getVel() const { return state().vel_; }
    Mat3D getRotCov() const { return cov().block3(0); }
    Mat3D getPosCov() const { return cov().block3(3); }
    Mat3D getVelCov() const { return cov().block3(6); }
    StateQ Q() const {
        StateQ q;
        dev_->check(lk_get_Q(dev_->h(), q.d.data()));
        return q;
    }
    void setQ(const StateQ& q) { dev_->check(lk_set_Q(dev_->h(), q.d.data())); }
    StateCov cov() const {
        StateCov c;
        dev_->check(lk_get_state(dev_->h(), 0, nullptr, c.d.data()));
        return c;
    }
    void setCov(const StateCov& c) { dev_->check(lk_set_state(dev_->h(), 0, nullptr, c.d.data())); }
    const Config& config() const { return config_; }

    void initProcessCovQ() { dev_->check(lk_init_process_cov_q(dev_->h())); }
    StateVec getFunctionf(double dt) {
        StateVec f;
        dev_->check(lk_get_function_f(dev_->h(), 0, dt, f.data()));
        return f;
    }
    StateF getFx(double dt) {
        StateF F;
        dev_->check(lk_get_fx(dev_->h(), 0, dt, F.d.data()));
        return F;
    }
    void predict(double dt, bool prop_state, bool prop_cov) { dev_->check(lk_predict(dev_->h(), 0, dt, prop_state, prop_cov)); }
    void updateByPoints(ObsShared& o) {
        dev_->check(lk_update_by_points(dev_->h(), 0, o.pt_h.data(), o.pt_z.data(), o.pt_R.data(), o.pt_z.size()));
    }
    void updateByImu(ObsShared& o) { dev_->check(lk_update_by_imu(dev_->h(), 0, o.ki_z.data(), o.ki_R.data())); }
    void updateByKinImu(ObsShared& o) {
        dev_->check(lk_update_by_kin_imu(dev_->h(), 0, o.ki_h.data(), o.ki_z.data(), o.ki_R.data(), o.ki_z.size()));
    }

   private:
    Config config_;
    std::shared_ptr<Device> dev_;
};

class VoxelMapManager {
   public:
    VoxelMapManager(VoxelMapConfig& config_setting, std::shared_ptr<Device> dev) : config_setting_(config_setting), dev_(std::move(dev)) {}
    VoxelMapConfig config_setting_;
    Mat3D extR_ = Mat3D::Identity();
    Vec3D extT_{};
    std::shared_ptr<PointCloudType> feats_down_body_, feats_down_world_;

    // voxel_map.cc:287-334.  rot / rot_cov / pos_cov are those of the filter (KILO.cc:339) and already live on
    // the device; the arguments are kept for source compatibility.
    void BuildVoxelMap(const Mat3D&, const Mat3D&, const Mat3D&) {
        if (!feats_down_body_ || !feats_down_world_ || feats_down_body_->size() != feats_down_world_->size())
            throw std::runtime_error("BuildVoxelMap: feats_down_body_/feats_down_world_ not set");
        std::vector<float> w, b;
        for (size_t i = 0; i < feats_down_world_->size(); ++i) {
            const PointType &pw = (*feats_down_world_)[i], &pb = (*feats_down_body_)[i];
            w.insert(w.end(), {pw.x, pw.y, pw.z});
            b.insert(b.end(), {pb.x, pb.y, pb.z});
        }
        dev_->check(lk_map_build(dev_->h(), w.data(), b.data(), feats_down_world_->size()));
    }
    // voxel_map.cc:336-361
    void UpdateVoxelMap(const std::vector<pointWithVar>& input_points) {
        std::vector<double> pw, var;
        for (const auto& p : input_points) {
            pw.insert(pw.end(), p.point_w.begin(), p.point_w.end());
            var.insert(var.end(), p.var.m, p.var.m + 9);
        }
        dev_->check(lk_map_update(dev_->h(), pw.data(), var.data(), input_points.size()));
    }
    // voxel_map.cc:552-569.  position_last_ is the public member the caller sets (voxel_map.h:199); last_slide_position
    // (voxel_map.h:201) lives in the device handle.
    Vec3D position_last_{};
    bool mapSliding() {
        int32_t slid = 0;
        dev_->check(lk_map_slide(dev_->h(), position_last_.data(), config_setting_.sliding_thresh, config_setting_.half_map_size, &slid, nullptr));
        return slid != 0;
    }
    // voxel_map.cc:571-594
    void clearMemOutOfMap(const int& x_max, const int& x_min, const int& y_max, const int& y_min, const int& z_max, const int& z_min) {
        dev_->check(lk_map_clear_outside(dev_->h(), x_max, x_min, y_max, y_min, z_max, z_min, nullptr));
    }
    // voxel_map.cc:363-427 for points the caller holds as pointWithVar, each started on the root voxel at keys[i] with
    // is_success = false, prob = 0 (KILO.cc:149-155).  found[i]: the find of KILO.cc:149 hit.  The reference's argument
    // `const VoxelOctoTree* current_octo` has no host-side counterpart (the octrees live in HBM): the key stands in for it.
    void build_single_residual(const std::vector<pointWithVar>& pv, const std::vector<std::array<int32_t, 3>>& keys, std::vector<uint8_t>& found,
                               std::vector<uint8_t>& is_success, std::vector<double>& prob, std::vector<PointToPlane>& single_ptpl) {
        const size_t n = pv.size();
        if (keys.size() != n) throw std::runtime_error("build_single_residual: one key per point");
        std::vector<int32_t> k(3 * n);
        std::vector<double> pw(3 * n), var(9 * n), nrm(3 * n), ctr(3 * n), d(n);
        std::vector<float> dis(n);
        std::vector<int32_t> layer(n);
        for (size_t i = 0; i < n; ++i) {
            std::copy(keys[i].begin(), keys[i].end(), k.begin() + 3 * i);
            std::copy(pv[i].point_w.begin(), pv[i].point_w.end(), pw.begin() + 3 * i);
            std::copy(pv[i].var.m, pv[i].var.m + 9, var.begin() + 9 * i);
        }
        found.assign(n, 0), is_success.assign(n, 0), prob.assign(n, 0.0), single_ptpl.assign(n, PointToPlane());
        dev_->check(lk_match_points(dev_->h(), n, k.data(), pw.data(), var.data(), found.data(), is_success.data(), prob.data(), nrm.data(), ctr.data(),
                                    d.data(), dis.data(), layer.data()));
        for (size_t i = 0; i < n; ++i) {
            PointToPlane& t = single_ptpl[i];
            t.point_w_ = pv[i].point_w;
            for (int c = 0; c < 3; ++c) t.normal_[c] = nrm[3 * i + c], t.center_[c] = ctr[3 * i + c];
            t.layer_ = layer[i], t.d_ = d[i], t.dis_to_plane_ = dis[i];
        }
    }
    // Residual build of KILO.cc:122-210 for a whole bucket (replaces the per-point build_single_residual calls).
    void BuildResidualList(const PointCloudType& body, size_t i0, size_t i1, ObsShared& obs, std::vector<uint8_t>& valid) {
        size_t n = i1 - i0;
        std::vector<float> b;
        for (size_t i = i0; i < i1; ++i) b.insert(b.end(), {body[i].x, body[i].y, body[i].z});
        std::vector<double> h(6 * n), z(n), R(n);
        valid.assign(n, 0);
        dev_->check(lk_residuals(dev_->h(), b.data(), n, h.data(), z.data(), R.data(), valid.data()));
        obs.pt_h.clear(), obs.pt_z.clear(), obs.pt_R.clear();
        for (size_t k = 0; k < n; ++k)
            if (valid[k]) {
                obs.pt_h.insert(obs.pt_h.end(), h.begin() + 6 * k, h.begin() + 6 * k + 6);
                obs.pt_z.push_back(z[k]);
                obs.pt_R.push_back(R[k]);
            }
    }

   private:
    std::shared_ptr<Device> dev_;
};

// The path of KILO (KILO.cc:108-399) with the reference's private method names.
class KiloPath {
   public:
    KiloPath(const ESKF::Config& ec, VoxelMapConfig& vc, const Mat3D& ext_rot, const Vec3D& ext_t, double gravity,
             const DeviceCaps& caps = DeviceCaps()) {
        lk_config c;
        std::memset(&c, 0, sizeof(c));
        std::memcpy(&c.vel_process_cov, &ec, sizeof(ec));  // same 14 doubles, same order (eskf.h:49-65)
        c.max_voxel_size = vc.max_voxel_size_;
        c.planner_threshold = vc.planner_threshold_;
        c.beam_err = vc.beam_err_;
        c.dept_err = vc.dept_err_;
        c.sigma_num = vc.sigma_num_;
        c.max_layer = vc.max_layer_;
        c.max_iterations = vc.max_iterations_;
        if (vc.layer_init_num_.size() < 5) throw std::runtime_error("layer_init_num needs 5 entries");
        for (int i = 0; i < 5; ++i) c.layer_init_num[i] = vc.layer_init_num_[i];
        c.max_points_num = vc.max_points_num_;
        std::memcpy(c.ext_R, ext_rot.m, sizeof(c.ext_R));
        for (int i = 0; i < 3; ++i) c.ext_T[i] = ext_t[i];
        c.gravity = gravity;
        c.device_id = caps.device_id;
        c.n_slots = caps.n_slots;
        c.max_roots = caps.max_roots;
        c.max_nodes = caps.max_nodes;
        c.max_point_blocks = caps.max_point_blocks;
        c.max_scan_points = caps.max_scan_points;
        dev_ = std::make_shared<Device>(c);
        eskf_ = std::make_unique<ESKF>(ec, dev_);
        map_manager_ = std::make_unique<VoxelMapManager>(vc, dev_);
        map_manager_->extR_ = ext_rot;
        map_manager_->extT_ = ext_t;
    }
    ESKF& eskf() { return *eskf_; }
    VoxelMapManager& map_manager() { return *map_manager_; }
    Device& device() { return *dev_; }   // the lk_handle behind both mirrors (multi-GPU helpers: legkilo_rccl.hpp)
    void setTimes(double last_predict, double last_update) { dev_->check(lk_set_times(dev_->h(), 0, last_predict, last_update)); }
    void setAccNorm(double a) { dev_->check(lk_set_acc_norm(dev_->h(), a)); }
    double accNorm() const {
        double a = 0.0;
        dev_->check(lk_get_acc_norm(dev_->h(), &a));
        return a;
    }

    // KILO.cc:108-233
    bool predictUpdatePoint(double current_time, size_t idx_i, size_t idx_j, const PointCloudType& cloud_down_body,
                            PointCloudType& cloud_down_world, size_t& success_pts_size_out) {
        size_t n = idx_j - idx_i;
        std::vector<float> b, w(3 * n), inten(n);
        for (size_t i = idx_i; i < idx_j; ++i) b.insert(b.end(), {cloud_down_body[i].x, cloud_down_body[i].y, cloud_down_body[i].z});
        size_t before = success_pts_size_out;
        dev_->check(lk_update_points(dev_->h(), current_time, b.data(), n, w.data(), inten.data(), &success_pts_size_out));
        for (size_t i = 0; i < n; ++i) {
            PointType& p = cloud_down_world[idx_i + i];
            p.x = w[3 * i], p.y = w[3 * i + 1], p.z = w[3 * i + 2], p.intensity = inten[i];
        }
        return success_pts_size_out > before;
    }
    bool predictUpdateImu(const lk_imu& imu) {  // KILO.cc:235-258
        dev_->check(lk_update_imu(dev_->h(), &imu));
        return true;
    }
    bool predictUpdateKinImu(const lk_kin_imu& kin) {  // KILO.cc:260-314
        dev_->check(lk_update_kin_imu(dev_->h(), &kin));
        return true;
    }
    // bucket loop of KILO::process (KILO.cc:367-396), fused on the device stream: one call per scan
    bool processSorted(const PointCloudType& sorted_body, double begin_time, const std::vector<lk_imu>& imus,
                       const std::vector<lk_kin_imu>& kins, PointCloudType* world_out, lk_pose* pose) {
        std::vector<lk_point> pts(sorted_body.size());
        for (size_t i = 0; i < pts.size(); ++i) pts[i] = lk_point{sorted_body[i].x, sorted_body[i].y, sorted_body[i].z, sorted_body[i].curvature};
        std::vector<float> w(world_out ? 3 * pts.size() : 0);
        dev_->check(lk_process_scan(dev_->h(), pts.data(), pts.size(), begin_time, imus.data(), imus.size(), kins.data(), kins.size(),
                                    world_out ? w.data() : nullptr, pose));
        if (world_out) {
            world_out->resize(pts.size());
            for (size_t i = 0; i < pts.size(); ++i) (*world_out)[i].x = w[3 * i], (*world_out)[i].y = w[3 * i + 1], (*world_out)[i].z = w[3 * i + 2];
        }
        return true;
    }

    // Many recorded scans at once against the CURRENT map, frozen (lk_batch_replay_scans_dev): scan s runs the bucket
    // loop of KILO::process (KILO.cc:375-395) on filter slot s from its own prior; buckets are the runs of equal curvature
    // of each time-sorted scan (KILO.cc:376-378), t_begin[s] its start time; `imus` (optional, one time-sorted vector per
    // scan) are applied between the buckets as in only_imu_use mode (KILO.cc:379-383), `kins` as in the default leg-fusion
    // mode (KILO.cc:384-390); at most one of the two.  n scans need DeviceCaps::n_slots >= n.
    std::vector<lk_pose> replayRecordedRun(const std::vector<PointCloudType>& sorted_scans, const std::vector<double>& t_begin,
                                           const std::vector<State>& prior_states, const std::vector<StateCov>& prior_covs,
                                           const std::vector<std::vector<lk_imu>>* imus = nullptr,
                                           const std::vector<std::vector<lk_kin_imu>>* kins = nullptr) {
        const size_t S = sorted_scans.size();
        if (t_begin.size() != S || prior_states.size() != S || prior_covs.size() != S || (imus && imus->size() != S) ||
            (kins && kins->size() != S))
            throw std::runtime_error("replayRecordedRun: one start time, prior state and prior covariance per scan");
        if (imus && kins) throw std::runtime_error("replayRecordedRun: IMU messages or kinematic + IMU messages, not both");
        std::vector<lk_kin_imu> kin_flat;
        // the scans go to HBM back to back; their time buckets (runs of equal curvature, KILO.cc:375-378) are found on the device
        std::vector<lk_point> pts;
        std::vector<uint64_t> scan_off(1, 0);
        std::vector<uint32_t> n_msg;
        std::vector<double> x36(S * LK_STATE_DOUBLES), P900(S * DIM_STATE * DIM_STATE);
        std::vector<lk_imu> imu_flat;
        for (size_t s = 0; s < S; ++s) {
            const PointCloudType& sc = sorted_scans[s];
            for (size_t i = 0; i < sc.size(); ++i) pts.push_back(lk_point{sc[i].x, sc[i].y, sc[i].z, sc[i].curvature});
            scan_off.push_back(pts.size());
            prior_states[s].to_x36(&x36[s * LK_STATE_DOUBLES]);
            std::memcpy(&P900[s * DIM_STATE * DIM_STATE], prior_covs[s].d.data(), sizeof(double) * DIM_STATE * DIM_STATE);
            if (imus) {
                n_msg.push_back((uint32_t)(*imus)[s].size());
                imu_flat.insert(imu_flat.end(), (*imus)[s].begin(), (*imus)[s].end());
            }
            if (kins) {
                n_msg.push_back((uint32_t)(*kins)[s].size());
                kin_flat.insert(kin_flat.end(), (*kins)[s].begin(), (*kins)[s].end());
            }
        }
        void* d_pts = nullptr;
        dev_->check(lk_device_malloc(dev_->h(), &d_pts, sizeof(lk_point) * std::max<size_t>(pts.size(), 1)));
        std::vector<lk_pose> out(S);
        int rc = lk_memcpy_h2d(dev_->h(), d_pts, pts.data(), sizeof(lk_point) * pts.size());
        if (!rc) rc = lk_batch_set_priors(dev_->h(), x36.data(), P900.data(), S);
        if (!rc)
            rc = lk_batch_replay_scans_dev(dev_->h(), static_cast<const lk_point*>(d_pts), S, scan_off.data(), t_begin.data(), kins ? 2 : (imus ? 1 : 0),
                                           (imus || kins) ? n_msg.data() : nullptr,
                                           kins ? static_cast<const void*>(kin_flat.data()) : static_cast<const void*>(imu_flat.data()), out.data());
        lk_device_free(dev_->h(), d_pts);
        dev_->check(rc);
        return out;
    }

    // Many equally shaped scans at once WITH the map insert of KILO::process (KILO.cc:216-233 after every bucket): scan s runs the whole
    // bucket loop on filter slot s from its own prior and inserts into its OWN copy-on-write overlay of the current map
    // (lk_batch_replay_overlay_dev); the map itself is not changed.  All scans share their bucket bounds (runs of equal curvature of
    // scan 0, KILO.cc:375-378) and start time.  Per scan the result equals processSorted() on a private copy of the map.
    std::vector<lk_pose> replayWithInsert(const std::vector<PointCloudType>& sorted_scans, double t_begin, const std::vector<State>& prior_states,
                                          const std::vector<StateCov>& prior_covs) {
        const size_t S = sorted_scans.size();
        if (S == 0 || prior_states.size() != S || prior_covs.size() != S) throw std::runtime_error("replayWithInsert: one prior state and covariance per scan");
        const size_t n = sorted_scans[0].size();
        std::vector<uint32_t> off(1, 0);
        std::vector<double> dt;
        for (size_t i = 0; i < n;) {
            size_t j = i + 1;
            while (j < n && sorted_scans[0][i].curvature == sorted_scans[0][j].curvature) ++j;
            dt.push_back((double)sorted_scans[0][i].curvature);
            off.push_back((uint32_t)j);
            i = j;
        }
        std::vector<lk_point> pts;
        pts.reserve(S * n);
        std::vector<double> x36(S * LK_STATE_DOUBLES), P900(S * DIM_STATE * DIM_STATE);
        for (size_t s = 0; s < S; ++s) {
            const PointCloudType& sc = sorted_scans[s];
            if (sc.size() != n) throw std::runtime_error("replayWithInsert: the scans of a batch have one size");
            for (size_t b = 0; b + 1 < off.size(); ++b)
                if (sc[off[b]].curvature != sorted_scans[0][off[b]].curvature || sc[off[b + 1] - 1].curvature != sc[off[b]].curvature)
                    throw std::runtime_error("replayWithInsert: the scans of a batch share their time buckets");
            for (size_t i = 0; i < n; ++i) pts.push_back(lk_point{sc[i].x, sc[i].y, sc[i].z, sc[i].curvature});
            prior_states[s].to_x36(&x36[s * LK_STATE_DOUBLES]);
            std::memcpy(&P900[s * DIM_STATE * DIM_STATE], prior_covs[s].d.data(), sizeof(double) * DIM_STATE * DIM_STATE);
        }
        void* d_pts = nullptr;
        dev_->check(lk_device_malloc(dev_->h(), &d_pts, sizeof(lk_point) * std::max<size_t>(pts.size(), 1)));
        std::vector<lk_pose> out(S);
        int rc = lk_memcpy_h2d(dev_->h(), d_pts, pts.data(), sizeof(lk_point) * pts.size());
        if (!rc) rc = lk_batch_set_priors(dev_->h(), x36.data(), P900.data(), S);
        if (!rc) rc = lk_batch_replay_overlay_dev(dev_->h(), static_cast<const lk_point*>(d_pts), S, n, t_begin, off.data(), dt.data(), dt.size(), out.data());
        lk_device_free(dev_->h(), d_pts);
        dev_->check(rc);
        return out;
    }

    // A recorded run's scans WITH the map insert (lk_batch_replay_overlay_ragged_dev): every scan its own size, its own time buckets (runs of
    // equal curvature, KILO.cc:375-378), its own start time and - optionally - its own IMU (only_imu_use, KILO.cc:379-383) or kinematic + IMU
    // (leg fusion, KILO.cc:384-390) messages between the buckets; each scan inserts into its own copy-on-write overlay of the current map,
    // the map itself is not changed.  Per scan the result equals processSorted() on a private copy of the map.
    std::vector<lk_pose> replayRecordedRunWithInsert(const std::vector<PointCloudType>& sorted_scans, const std::vector<double>& t_begin,
                                                     const std::vector<State>& prior_states, const std::vector<StateCov>& prior_covs,
                                                     const std::vector<std::vector<lk_imu>>* imus = nullptr,
                                                     const std::vector<std::vector<lk_kin_imu>>* kins = nullptr) {
        const size_t S = sorted_scans.size();
        if (S == 0 || t_begin.size() != S || prior_states.size() != S || prior_covs.size() != S || (imus && imus->size() != S) || (kins && kins->size() != S))
            throw std::runtime_error("replayRecordedRunWithInsert: one start time, prior state and prior covariance per scan");
        if (imus && kins) throw std::runtime_error("replayRecordedRunWithInsert: IMU messages or kinematic + IMU messages, not both");
        std::vector<lk_point> pts;
        std::vector<uint64_t> scan_off(1, 0);
        std::vector<uint32_t> n_buckets, bucket_off, n_msg;
        std::vector<double> bucket_dt, x36(S * LK_STATE_DOUBLES), P900(S * DIM_STATE * DIM_STATE);
        std::vector<lk_imu> imu_flat;
        std::vector<lk_kin_imu> kin_flat;
        for (size_t s = 0; s < S; ++s) {
            const PointCloudType& sc = sorted_scans[s];
            if (sc.size() == 0) throw std::runtime_error("replayRecordedRunWithInsert: empty scan");
            uint32_t nb = 0;
            bucket_off.push_back(0);
            for (size_t i = 0; i < sc.size();) {   // KILO.cc:375-378
                size_t j = i + 1;
                while (j < sc.size() && sc[i].curvature == sc[j].curvature) ++j;
                bucket_dt.push_back((double)sc[i].curvature);
                bucket_off.push_back((uint32_t)j);
                ++nb;
                i = j;
            }
            n_buckets.push_back(nb);
            for (size_t i = 0; i < sc.size(); ++i) pts.push_back(lk_point{sc[i].x, sc[i].y, sc[i].z, sc[i].curvature});
            scan_off.push_back(pts.size());
            prior_states[s].to_x36(&x36[s * LK_STATE_DOUBLES]);
            std::memcpy(&P900[s * DIM_STATE * DIM_STATE], prior_covs[s].d.data(), sizeof(double) * DIM_STATE * DIM_STATE);
            if (imus) n_msg.push_back((uint32_t)(*imus)[s].size()), imu_flat.insert(imu_flat.end(), (*imus)[s].begin(), (*imus)[s].end());
            if (kins) n_msg.push_back((uint32_t)(*kins)[s].size()), kin_flat.insert(kin_flat.end(), (*kins)[s].begin(), (*kins)[s].end());
        }
        void* d_pts = nullptr;
        dev_->check(lk_device_malloc(dev_->h(), &d_pts, sizeof(lk_point) * std::max<size_t>(pts.size(), 1)));
        std::vector<lk_pose> out(S);
        int rc = lk_memcpy_h2d(dev_->h(), d_pts, pts.data(), sizeof(lk_point) * pts.size());
        if (!rc) rc = lk_batch_set_priors(dev_->h(), x36.data(), P900.data(), S);
        if (!rc)
            rc = lk_batch_replay_overlay_ragged_dev(dev_->h(), static_cast<const lk_point*>(d_pts), S, scan_off.data(), n_buckets.data(), bucket_off.data(), bucket_dt.data(),
                                                    t_begin.data(), (imus || kins) ? n_msg.data() : nullptr,
                                                    kins ? static_cast<const void*>(kin_flat.data()) : static_cast<const void*>(imu_flat.data()), kins ? 2 : (imus ? 1 : 0),
                                                    out.data());
        lk_device_free(dev_->h(), d_pts);
        dev_->check(rc);
        return out;
    }

   private:
    std::shared_ptr<Device> dev_;
    std::unique_ptr<ESKF> eskf_;
    std::unique_ptr<VoxelMapManager> map_manager_;
};

}  // namespace LEGKILO_HOST_NAMESPACE
