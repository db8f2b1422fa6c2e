// legkilo_host_eigen.hpp — the class surface KILO consumes, with the REFERENCE'S OWN TYPES (Eigen matrices, pcl clouds, ROS messages),
// for use inside the reference's source tree.  It is what tests/integration/kilo_hip.ed makes core/slam/eskf.h and core/slam/voxel_map.h
// forward to: KILO.cc, KILO.h and preprocess/state_initial.hpp then compile against
//
//   legkilo::State, ObsShared, ESKF            <- legkilo/src/core/slam/eskf.h:15-109
//   legkilo::VoxelMapConfig, pointWithVar,
//            VoxelMapManager                   <- legkilo/src/core/slam/voxel_map.h:41-78,180-244
//   legkilo::KiloPath                          <- KILO::predictUpdatePoint / predictUpdateImu / predictUpdateKinImu (KILO.cc:108-314)
//
// and eskf.cc / voxel_map.cc leave the build: everything below is a thin layer over legkilo_host.hpp (std-only mirror, included under the
// name legkilo_hip) and through it over the C-ABI of include/legkilo_hip.h.  No arithmetic of the path happens in this header.
//
// Needs, from the including tree: <Eigen/Dense> and the reference's common/eigen_types.hpp, common/pcl_types.h,
// common/sensor_types.hpp (found on the reference's own include path).  Only element access, sizes and resize are used of Eigen,
// so that the header does not depend on an Eigen version.
//
// What differs from the reference's classes - all of it because the state lives in HBM:
//   * ESKF::state(), cov(), Q() hand out write-back proxies instead of references: `eskf.state().grav_ = g;`, `eskf.cov() = P0;`
//     (state_initial.hpp:66-70) work as written - the proxy IS a State / StateCov and stores itself to the device when it goes out of
//     scope, if it was changed.  A proxy bound to a long-lived name keeps the value it was created with: read again after a call
//     that moves the filter.
//   * VoxelMapManager::voxel_map_ (the unordered_map of octrees) does not exist on the host; build_single_residual takes the voxel's
//     key instead of its octree, and a bucket goes through one call (BuildResidualList).
//   * errors of the C-ABI become std::runtime_error, like YamlHelper's configuration errors (yaml_helper.hpp:42,50).
#pragma once
#define LEGKILO_HOST_NAMESPACE legkilo_hip
#include "legkilo_host.hpp"

#include <sensor_msgs/Imu.h>

#include "common/eigen_types.hpp"
#include "common/pcl_types.h"
#include "common/sensor_types.hpp"

namespace legkilo {

constexpr int DIM_STATE = LK_DIM_STATE;
using StateVec = Eigen::Matrix<double, DIM_STATE, 1>;
using StateCov = Eigen::Matrix<double, DIM_STATE, DIM_STATE>;
using StateF = Eigen::Matrix<double, DIM_STATE, DIM_STATE>;
using StateQ = Eigen::Matrix<double, DIM_STATE, DIM_STATE>;

namespace hip_glue {
inline legkilo_hip::Mat3D to_hip(const Mat3D& m) {
    legkilo_hip::Mat3D r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = m(i, j);
    return r;
}
inline legkilo_hip::Vec3D to_hip(const Vec3D& v) { return legkilo_hip::Vec3D{{v(0), v(1), v(2)}}; }
inline Mat3D to_eigen(const legkilo_hip::Mat3D& m) {
    Mat3D r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = m(i, j);
    return r;
}
inline Vec3D to_eigen(const legkilo_hip::Vec3D& v) {
    Vec3D r;
    for (int c = 0; c < 3; ++c) r(c) = v[c];
    return r;
}
template <class M>
inline void to_hip30(const M& m, legkilo_hip::StateCov& c) {
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) c(i, j) = m(i, j);
}
template <class M>
inline void to_eigen30(const legkilo_hip::StateCov& c, M& m) {
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) m(i, j) = c(i, j);
}
}  // namespace hip_glue

// eskf.h:15-32 (the ⊞ / ⊟ operators of eskf.cc:18-45 run on the device: ESKF::predict and the three updates)
struct State {
    Mat3D rot_;
    Vec3D pos_, vel_, ba_, bw_, grav_, imu_a_, imu_w_, bv_, contact_;
    State() { from_hip(legkilo_hip::State()); }   // identity, zeros, grav_ = (0, 0, -9.81): eskf.cc:5-16
    void from_hip(const legkilo_hip::State& s) {
        rot_ = hip_glue::to_eigen(s.rot_);
        pos_ = hip_glue::to_eigen(s.pos_), vel_ = hip_glue::to_eigen(s.vel_), ba_ = hip_glue::to_eigen(s.ba_);
        bw_ = hip_glue::to_eigen(s.bw_), grav_ = hip_glue::to_eigen(s.grav_), imu_a_ = hip_glue::to_eigen(s.imu_a_);
        imu_w_ = hip_glue::to_eigen(s.imu_w_), bv_ = hip_glue::to_eigen(s.bv_), contact_ = hip_glue::to_eigen(s.contact_);
    }
    legkilo_hip::State to_hip() const {
        legkilo_hip::State s;
        s.rot_ = hip_glue::to_hip(rot_);
        s.pos_ = hip_glue::to_hip(pos_), s.vel_ = hip_glue::to_hip(vel_), s.ba_ = hip_glue::to_hip(ba_);
        s.bw_ = hip_glue::to_hip(bw_), s.grav_ = hip_glue::to_hip(grav_), s.imu_a_ = hip_glue::to_hip(imu_a_);
        s.imu_w_ = hip_glue::to_hip(imu_w_), s.bv_ = hip_glue::to_hip(bv_), s.contact_ = hip_glue::to_hip(contact_);
        return s;
    }
};

// eskf.h:34-44
struct ObsShared {
    Eigen::Matrix<double, Eigen::Dynamic, 1> pt_z;
    Eigen::Matrix<double, Eigen::Dynamic, 6> pt_h;
    Eigen::Matrix<double, Eigen::Dynamic, 1> pt_R;
    Eigen::Matrix<double, Eigen::Dynamic, 1> ki_z;
    Eigen::Matrix<double, Eigen::Dynamic, DIM_STATE> ki_h;
    Eigen::Matrix<double, Eigen::Dynamic, 1> ki_R;
};

class ESKF {
   public:
    // eskf.h:49-65, same fields, same order (= the first 14 doubles of lk_config)
    struct Config {
        double vel_process_cov, imu_acc_process_cov, imu_gyr_process_cov, contact_process_cov, acc_bias_process_cov,
            gyr_bias_process_cov, kin_bias_process_cov;
        double imu_acc_meas_noise, imu_acc_z_meas_noise, imu_gyr_meas_noise, kin_meas_noise, chd_meas_noise,
            contact_meas_noise, lidar_point_meas_ratio;
    };
    static legkilo_hip::ESKF::Config to_hip(const Config& c) {
        static_assert(sizeof(Config) == sizeof(legkilo_hip::ESKF::Config), "ESKF::Config is 14 doubles on both sides");
        legkilo_hip::ESKF::Config h;
        std::memcpy(&h, &c, sizeof(h));
        return h;
    }
    ESKF(const Config& config, legkilo_hip::ESKF* dev) : config_(config), dev_(dev) {}

    // write-back proxies (see the header comment)
    class StateRef : public State {
       public:
        StateRef(ESKF* e, const State& s) : State(s), e_(e) { s.to_hip().to_x36(was_); }
        StateRef(const StateRef&) = delete;
        ~StateRef() noexcept(false) {
            double now[LK_STATE_DOUBLES];
            this->to_hip().to_x36(now);
            if (std::memcmp(now, was_, sizeof(now)) != 0) e_->setState(*this);
        }

       private:
        ESKF* e_;
        double was_[LK_STATE_DOUBLES];
    };
    template <bool IS_Q>
    class MatRef : public StateCov {
       public:
        MatRef(ESKF* e, const StateCov& m) : StateCov(m), e_(e), was_(m) {}
        MatRef(const MatRef&) = delete;
        using StateCov::operator=;
        ~MatRef() noexcept(false) {
            bool same = true;
            for (int i = 0; i < DIM_STATE && same; ++i)
                for (int j = 0; j < DIM_STATE; ++j)
                    if (!((*this)(i, j) == was_(i, j))) {   // !(==): a NaN written by the caller is a change
                        same = false;
                        break;
                    }
            if (!same) IS_Q ? e_->setQ(*this) : e_->setCov(*this);
        }

       private:
        ESKF* e_;
        StateCov was_;
    };
    using CovRef = MatRef<false>;
    using QRef = MatRef<true>;

    StateRef state() { return StateRef(this, static_cast<const ESKF*>(this)->state()); }
    State state() const {
        State s;
        s.from_hip(dev_->state());
        return s;
    }
    void setState(const State& s) { dev_->setState(s.to_hip()); }
    Mat3D getRot() const { return state().rot_; }
    Vec3D getPos() const { return state().pos_; }
    Vec3D getVel() const { return state().vel_; }
    Mat3D getRotCov() const { return block3(0); }
    Mat3D getPosCov() const { return block3(3); }
    Mat3D getVelCov() const { return block3(6); }

    QRef Q() { return QRef(this, static_cast<const ESKF*>(this)->Q()); }
    StateQ Q() const {
        StateQ q;
        hip_glue::to_eigen30(dev_->Q(), q);
        return q;
    }
    void setQ(const StateQ& Q) {
        legkilo_hip::StateQ q;
        hip_glue::to_hip30(Q, q);
        dev_->setQ(q);
    }
    CovRef cov() { return CovRef(this, static_cast<const ESKF*>(this)->cov()); }
    StateCov cov() const {
        StateCov c;
        hip_glue::to_eigen30(dev_->cov(), c);
        return c;
    }
    void setCov(const StateCov& P) {
        legkilo_hip::StateCov c;
        hip_glue::to_hip30(P, c);
        dev_->setCov(c);
    }
    Config& config() { return config_; }
    const Config& config() const { return config_; }

    void initProcessCovQ() { dev_->initProcessCovQ(); }   // eskf.cc:47-62, from the Config the handle was created with
    StateVec getFunctionf(double dt) {
        legkilo_hip::StateVec f = dev_->getFunctionf(dt);
        StateVec r;
        for (int i = 0; i < DIM_STATE; ++i) r(i) = f[i];
        return r;
    }
    StateF getFx(double dt) {
        StateF F;
        hip_glue::to_eigen30(dev_->getFx(dt), F);
        return F;
    }
    void predict(double dt, bool prop_state, bool prop_cov) { dev_->predict(dt, prop_state, prop_cov); }
    void updateByPoints(ObsShared& o) {
        legkilo_hip::ObsShared h;
        const size_t n = (size_t)o.pt_z.rows();
        h.pt_h.resize(6 * n), h.pt_z.resize(n), h.pt_R.resize(n);
        for (size_t k = 0; k < n; ++k) {
            for (int c = 0; c < 6; ++c) h.pt_h[6 * k + c] = o.pt_h(k, c);
            h.pt_z[k] = o.pt_z(k), h.pt_R[k] = o.pt_R(k);
        }
        dev_->updateByPoints(h);
    }
    void updateByImu(ObsShared& o) {
        legkilo_hip::ObsShared h;
        h.ki_z.resize(6), h.ki_R.resize(6);
        for (int k = 0; k < 6; ++k) h.ki_z[k] = o.ki_z(k), h.ki_R[k] = o.ki_R(k);
        dev_->updateByImu(h);
    }
    void updateByKinImu(ObsShared& o) {
        legkilo_hip::ObsShared h;
        const size_t m = (size_t)o.ki_z.rows();
        h.ki_h.resize(DIM_STATE * m), h.ki_z.resize(m), h.ki_R.resize(m);
        for (size_t k = 0; k < m; ++k) {
            for (int c = 0; c < DIM_STATE; ++c) h.ki_h[DIM_STATE * k + c] = o.ki_h(k, c);
            h.ki_z[k] = o.ki_z(k), h.ki_R[k] = o.ki_R(k);
        }
        dev_->updateByKinImu(h);
    }

   private:
    Mat3D block3(int o) const { return hip_glue::to_eigen(dev_->cov().block3(o)); }
    Config config_;
    legkilo_hip::ESKF* dev_;
};

// voxel_map.h:41-57
typedef struct VoxelMapConfig {
    double max_voxel_size_;
    int max_layer_;
    int max_iterations_;
    std::vector<int> layer_init_num_;
    int max_points_num_;
    double planner_threshold_;
    double beam_err_;
    double dept_err_;
    double sigma_num_;
    bool is_pub_plane_map_;
    double sliding_thresh;
    bool map_sliding_en;
    int half_map_size;
} VoxelMapConfig;

// voxel_map.h:59-78
typedef struct pointWithVar {
    Eigen::Vector3d point_b, point_i, point_w;
    Eigen::Matrix3d var_nostate, body_var, var, point_crossmat;
    Eigen::Vector3d normal;
    pointWithVar() {
        var_nostate = Eigen::Matrix3d::Zero(), var = Eigen::Matrix3d::Zero(), body_var = Eigen::Matrix3d::Zero();
        point_crossmat = Eigen::Matrix3d::Zero();
        point_b = Eigen::Vector3d::Zero(), point_i = Eigen::Vector3d::Zero(), point_w = Eigen::Vector3d::Zero();
        normal = Eigen::Vector3d::Zero();
    }
} pointWithVar;

// voxel_map.h:80-94
typedef struct PointToPlane {
    Eigen::Vector3d point_b_, point_w_, normal_, center_;
    Eigen::Matrix<double, 3, 3> point_crossmat_;
    Eigen::Matrix<double, 6, 6> plane_var_;   // not fetched from the device: zero (KILO.cc reads it only through sigma_l, which the device has evaluated)
    Eigen::Matrix3d body_cov_;
    int layer_;
    double d_, eigen_value_;
    bool is_valid_;
    float dis_to_plane_;
    double dis_r;
    PointToPlane() : layer_(-1), d_(0), eigen_value_(0), is_valid_(false), dis_to_plane_(0), dis_r(0) {
        point_b_.setZero(), point_w_.setZero(), normal_.setZero(), center_.setZero(), point_crossmat_.setZero(), plane_var_.setZero(), body_cov_.setZero();
    }
} PointToPlane;

class VoxelMapManager {
   public:
    VoxelMapManager(VoxelMapConfig& config_setting, legkilo_hip::VoxelMapManager* dev) : config_setting_(config_setting), dev_(dev) {
        feats_undistort_.reset(new PointCloudType());
        feats_down_body_.reset(new PointCloudType());
        feats_down_world_.reset(new PointCloudType());
    }
    static legkilo_hip::VoxelMapConfig to_hip(const VoxelMapConfig& c) {
        legkilo_hip::VoxelMapConfig h;
        h.max_voxel_size_ = c.max_voxel_size_, h.max_layer_ = c.max_layer_, h.max_iterations_ = 0;   // max_iterations_ is never loaded (KILO.cc:57-71)
        h.layer_init_num_ = c.layer_init_num_, h.max_points_num_ = c.max_points_num_, h.planner_threshold_ = c.planner_threshold_;
        h.beam_err_ = c.beam_err_, h.dept_err_ = c.dept_err_, h.sigma_num_ = c.sigma_num_, h.is_pub_plane_map_ = c.is_pub_plane_map_;
        h.sliding_thresh = c.sliding_thresh, h.map_sliding_en = c.map_sliding_en, h.half_map_size = c.half_map_size;
        return h;
    }
    VoxelMapConfig config_setting_;
    CloudPtr feats_undistort_, feats_down_body_, feats_down_world_;
    Eigen::Matrix3d extR_;
    Eigen::Vector3d extT_;
    Eigen::Vector3d position_last_;

    // voxel_map.cc:287-334; rot / rot_cov / pos_cov are the filter's (KILO.cc:339) and already live on the device
    void BuildVoxelMap(const Eigen::Matrix3d, const Eigen::Matrix3d, const Eigen::Matrix3d) {
        dev_->feats_down_body_ = std::make_shared<legkilo_hip::PointCloudType>(to_hip(*feats_down_body_));
        dev_->feats_down_world_ = std::make_shared<legkilo_hip::PointCloudType>(to_hip(*feats_down_world_));
        dev_->BuildVoxelMap(legkilo_hip::Mat3D::Identity(), legkilo_hip::Mat3D::Identity(), legkilo_hip::Mat3D::Identity());
        dev_->feats_down_body_.reset(), dev_->feats_down_world_.reset();
    }
    // voxel_map.cc:336-361
    void UpdateVoxelMap(const std::vector<pointWithVar>& input_points) {
        std::vector<legkilo_hip::pointWithVar> v(input_points.size());
        for (size_t i = 0; i < v.size(); ++i) v[i].point_w = hip_glue::to_hip(Vec3D(input_points[i].point_w)), v[i].var = hip_glue::to_hip(Mat3D(input_points[i].var));
        dev_->UpdateVoxelMap(v);
    }
    // voxel_map.cc:552-594
    bool mapSliding() {
        dev_->config_setting_.sliding_thresh = config_setting_.sliding_thresh, dev_->config_setting_.half_map_size = config_setting_.half_map_size;
        dev_->position_last_ = hip_glue::to_hip(Vec3D(position_last_));
        return dev_->mapSliding();
    }
    void clearMemOutOfMap(const int& x_max, const int& x_min, const int& y_max, const int& y_min, const int& z_max, const int& z_min) {
        dev_->clearMemOutOfMap(x_max, x_min, y_max, y_min, z_max, z_min);
    }
    // voxel_map.cc:363-427 on the root voxel at `position` (the key of voxel_map_, voxel_map.h:186), entered like KILO.cc:149-155 with the caller's
    // is_success / prob; returns whether a root voxel exists there (the find of KILO.cc:149).  `position` stands in for the reference's
    // `const VoxelOctoTree* current_octo, const int current_layer`: the octrees live in HBM.
    bool build_single_residual(pointWithVar& pv, const Eigen::Vector3i& position, bool& is_success, double& prob, PointToPlane& single_ptpl) {
        std::vector<legkilo_hip::pointWithVar> v(1);
        v[0].point_w = hip_glue::to_hip(Vec3D(pv.point_w)), v[0].var = hip_glue::to_hip(Mat3D(pv.var));
        std::vector<std::array<int32_t, 3>> key(1, std::array<int32_t, 3>{{position[0], position[1], position[2]}});
        std::vector<uint8_t> found, ok;
        std::vector<double> pr;
        std::vector<legkilo_hip::PointToPlane> pl;
        dev_->build_single_residual(v, key, found, ok, pr, pl);
        if (!found[0]) return false;
        if (ok[0]) is_success = true;
        if (ok[0] && pr[0] > prob) {   // voxel_map.cc:389-406
            prob = pr[0];
            for (int c = 0; c < 3; ++c) single_ptpl.normal_[c] = pl[0].normal_[c], single_ptpl.center_[c] = pl[0].center_[c];
            pv.normal = single_ptpl.normal_;
            single_ptpl.body_cov_ = pv.body_var, single_ptpl.point_b_ = pv.point_b, single_ptpl.point_w_ = pv.point_w;
            single_ptpl.point_crossmat_ = pv.point_crossmat;
            single_ptpl.d_ = pl[0].d_, single_ptpl.layer_ = pl[0].layer_, single_ptpl.dis_to_plane_ = pl[0].dis_to_plane_;
        }
        return true;
    }
    // the residual build of KILO.cc:122-210 for one bucket [i0, i1) of a cloud (stands in for the per-point build_single_residual calls)
    void BuildResidualList(const PointCloudType& body, size_t i0, size_t i1, ObsShared& obs, std::vector<uint8_t>& valid) {
        legkilo_hip::ObsShared h;
        dev_->BuildResidualList(to_hip(body), i0, i1, h, valid);
        const size_t n = h.pt_z.size();
        obs.pt_h.resize(n, 6), obs.pt_z.resize(n), obs.pt_R.resize(n);
        for (size_t k = 0; k < n; ++k) {
            for (int c = 0; c < 6; ++c) obs.pt_h(k, c) = h.pt_h[6 * k + c];
            obs.pt_z(k) = h.pt_z[k], obs.pt_R(k) = h.pt_R[k];
        }
    }
    static legkilo_hip::PointCloudType to_hip(const PointCloudType& c) {
        legkilo_hip::PointCloudType v(c.points.size());
        for (size_t i = 0; i < v.size(); ++i) {
            const PointType& p = c.points[i];
            v[i].x = p.x, v[i].y = p.y, v[i].z = p.z, v[i].intensity = p.intensity, v[i].curvature = p.curvature;
        }
        return v;
    }

   private:
    legkilo_hip::VoxelMapManager* dev_;
};

// KILO's three per-sensor handlers (KILO.cc:108-314) with the reference's argument types, plus the whole bucket loop as one call.
class KiloPath {
   public:
    KiloPath(const ESKF::Config& ec, VoxelMapConfig& vc, const Mat3D& ext_rot, const Vec3D& ext_t, double gravity,
             const legkilo_hip::DeviceCaps& caps = legkilo_hip::DeviceCaps())
        : hip_vc_(VoxelMapManager::to_hip(vc)),
          impl_(ESKF::to_hip(ec), hip_vc_, hip_glue::to_hip(ext_rot), hip_glue::to_hip(ext_t), gravity, caps),
          eskf_(ec, &impl_.eskf()),
          map_manager_(vc, &impl_.map_manager()) {
        map_manager_.extR_ = ext_rot;
        map_manager_.extT_ = ext_t;
    }
    ESKF& eskf() { return eskf_; }
    const ESKF& eskf() const { return eskf_; }
    VoxelMapManager& map_manager() { return map_manager_; }
    legkilo_hip::Device& device() { return impl_.device(); }
    void setTimes(double last_predict, double last_update) { impl_.setTimes(last_predict, last_update); }   // KILO.cc:350-351
    void getTimes(double* last_predict, double* last_update) { impl_.device().check(lk_get_times(impl_.device().h(), 0, last_predict, last_update)); }
    void setAccNorm(double a) { impl_.setAccNorm(a); }                                                       // KILO.cc:349
    double accNorm() const { return impl_.accNorm(); }

    // KILO.cc:108-233
    bool predictUpdatePoint(double current_time, size_t idx_i, size_t idx_j, const PointCloudType& cloud_down_body, PointCloudType& cloud_down_world,
                            size_t& success_pts_size_out) {
        const size_t n = idx_j - idx_i;
        std::vector<float> b(3 * n), w(3 * n), inten(n);
        for (size_t i = 0; i < n; ++i) {
            const PointType& p = cloud_down_body.points[idx_i + i];
            b[3 * i] = p.x, b[3 * i + 1] = p.y, b[3 * i + 2] = p.z;
        }
        const size_t before = success_pts_size_out;
        impl_.device().check(lk_update_points(impl_.device().h(), current_time, b.data(), n, w.data(), inten.data(), &success_pts_size_out));
        for (size_t i = 0; i < n; ++i) {
            PointType& p = cloud_down_world.points[idx_i + i];
            p.x = w[3 * i], p.y = w[3 * i + 1], p.z = w[3 * i + 2], p.intensity = inten[i];
        }
        return success_pts_size_out > before;
    }
    static lk_imu to_lk(const sensor_msgs::Imu& m) {
        lk_imu r;
        r.stamp = m.header.stamp.toSec();
        r.acc[0] = m.linear_acceleration.x, r.acc[1] = m.linear_acceleration.y, r.acc[2] = m.linear_acceleration.z;
        r.gyr[0] = m.angular_velocity.x, r.gyr[1] = m.angular_velocity.y, r.gyr[2] = m.angular_velocity.z;
        return r;
    }
    static lk_kin_imu to_lk(const common::KinImuMeas& k) {
        lk_kin_imu r;
        r.time_stamp = k.time_stamp_;
        for (int l = 0; l < 4; ++l) {
            for (int c = 0; c < 3; ++c) r.foot_pos[l][c] = k.foot_pos_[l][c], r.foot_vel[l][c] = k.foot_vel_[l][c];
            r.contact[l] = k.contact_[l] ? 1 : 0;
        }
        for (int c = 0; c < 3; ++c) r.acc[c] = k.acc_[c], r.gyr[c] = k.gyr_[c];
        return r;
    }
    bool predictUpdateImu(const sensor_msgs::ImuPtr& imu) { return impl_.predictUpdateImu(to_lk(*imu)); }                // KILO.cc:235-258
    bool predictUpdateKinImu(const common::KinImuMeas& kin_imu) { return impl_.predictUpdateKinImu(to_lk(kin_imu)); }    // KILO.cc:260-314

    // The bucket loop of KILO::process (KILO.cc:372-395) as ONE call: time-sorted cloud in, every message stamped before a bucket's time
    // applied before it, messages the scan did not reach left in the queues - what the loop leaves behind.
    template <class ImuQueue, class KinQueue>
    void processSorted(const PointCloudType& sorted_body, double begin_time, bool imu_mode_only, ImuQueue& imus, KinQueue& kin_imus,
                       PointCloudType& cloud_down_world, size_t& success_pts_size_out) {
        const size_t n = sorted_body.points.size();
        if (n == 0) return;
        const double last_bucket_time = begin_time + sorted_body.points[n - 1].curvature;
        std::vector<lk_imu> li;
        std::vector<lk_kin_imu> lk;
        if (imu_mode_only)
            while (!imus.empty() && imus.front()->header.stamp.toSec() < last_bucket_time) li.push_back(to_lk(*imus.front())), imus.pop_front();
        else
            while (!kin_imus.empty() && kin_imus.front().time_stamp_ < last_bucket_time) lk.push_back(to_lk(kin_imus.front())), kin_imus.pop_front();
        std::vector<lk_point> pts(n);
        for (size_t i = 0; i < n; ++i) {
            const PointType& p = sorted_body.points[i];
            pts[i] = lk_point{p.x, p.y, p.z, p.curvature};
        }
        std::vector<float> w(3 * n);
        lk_pose pose;
        impl_.device().check(lk_process_scan(impl_.device().h(), pts.data(), n, begin_time, li.data(), li.size(), lk.data(), lk.size(), w.data(), &pose));
        for (size_t i = 0; i < n; ++i) cloud_down_world.points[i].x = w[3 * i], cloud_down_world.points[i].y = w[3 * i + 1], cloud_down_world.points[i].z = w[3 * i + 2];
        success_pts_size_out += (size_t)pose.n_effect;
    }

   private:
    legkilo_hip::VoxelMapConfig hip_vc_;
    legkilo_hip::KiloPath impl_;
    ESKF eskf_;
    VoxelMapManager map_manager_;
};

}  // namespace legkilo
