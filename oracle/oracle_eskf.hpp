// ORACLE — TEST INFRASTRUCTURE ONLY (see smallmat.hpp header).  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
//
// CPU restatement of legkilo/src/core/slam/eskf.{h,cc}: the 30-dof error-state
// Kalman filter.  Each function cites the reference lines it follows.
#pragma once
#include "smallmat.hpp"

namespace lko {

constexpr int DIM_STATE = 30;
using StateVec = Mat<DIM_STATE, 1>;
using StateCov = Mat<DIM_STATE, DIM_STATE>;

// eskf.h:15-32
struct State {
    Mat3 rot_;
    Vec3 pos_, vel_, ba_, bw_, grav_, imu_a_, imu_w_, bv_, contact_;
    State();                                 // eskf.cc:5-16
    void operator+=(const StateVec& delta);  // eskf.cc:18-29
    StateVec operator-(const State& other);  // eskf.cc:31-45
};

// eskf.h:34-44 (dynamic Eigen columns -> std::vector, row-major h)
struct ObsShared {
    std::vector<double> pt_z;
    std::vector<double> pt_h;  // N x 6 row-major
    std::vector<double> pt_R;
    std::vector<double> ki_z;
    std::vector<double> ki_h;  // M x 30 row-major
    std::vector<double> ki_R;
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
    explicit ESKF(const Config& c) : config_(c) {
        cov_ = StateCov::Zero();
        Q_ = StateCov::Zero();
    }
    State& state() { return state_; }
    void setState(const State& s) { state_ = s; }
    Mat3 getRot() const { return state_.rot_; }
    Vec3 getPos() const { return state_.pos_; }
    Vec3 getVel() const { return state_.vel_; }
    Mat3 getRotCov() const { return block3(0); }
    Mat3 getPosCov() const { return block3(3); }
    Mat3 getVelCov() const { return block3(6); }
    StateCov& Q() { return Q_; }
    StateCov& cov() { return cov_; }
    Config& config() { return config_; }

    void initProcessCovQ();               // eskf.cc:47-62
    StateVec getFunctionf(double dt);     // eskf.cc:64-70
    StateCov getFx(double dt);            // eskf.cc:72-81
    void predict(double dt, bool prop_state, bool prop_cov);  // eskf.cc:83-89
    void updateByPoints(ObsShared& obs);  // eskf.cc:91-113
    void updateByImu(ObsShared& obs);     // eskf.cc:125-135
    void updateByKinImu(ObsShared& obs);  // eskf.cc:137-145

    // N above which updateByPoints switches from the reference's literal N x N inverse
    // (eskf.cc:105-112, O(N^3)) to the algebraically identical 6 x 6 information form
    // (SURVEY.md section 3.3 / 8a9).  N == 1 always takes the literal branch (eskf.cc:98-104).
    int literal_max_n = 512;

   private:
    Mat3 block3(int o) const {
        Mat3 b;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) b(i, j) = cov_(o + i, o + j);
        return b;
    }
    void updateByPointsLiteral(const ObsShared& obs, int N);
    void updateByPointsInfo6(const ObsShared& obs, int N);
    Config config_;
    State state_;
    StateCov cov_;
    StateCov Q_;
};

}  // namespace lko
