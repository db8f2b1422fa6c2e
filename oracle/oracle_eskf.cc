// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
// Restatement of legkilo/src/core/slam/eskf.cc.
#include "oracle_eskf.hpp"

namespace lko {

// eskf.cc:5-16
State::State() {
    rot_ = Mat3::Identity();
    pos_ = vel_ = ba_ = bw_ = imu_a_ = imu_w_ = bv_ = contact_ = Vec3::Zero();
    grav_ = vec3(0.0, 0.0, -9.81);
}

static inline Vec3 seg3(const StateVec& d, int o) { return vec3(d[o], d[o + 1], d[o + 2]); }
static inline void put3(StateVec& d, int o, const Vec3& v) { d[o] = v[0], d[o + 1] = v[1], d[o + 2] = v[2]; }

// eskf.cc:18-29
void State::operator+=(const StateVec& delta) {
    rot_ = rot_ * Exp3(delta[0], delta[1], delta[2]);
    pos_ += seg3(delta, 3);
    vel_ += seg3(delta, 6);
    ba_ += seg3(delta, 9);
    bw_ += seg3(delta, 12);
    grav_ += seg3(delta, 15);
    imu_a_ += seg3(delta, 18);
    imu_w_ += seg3(delta, 21);
    bv_ += seg3(delta, 24);
    contact_ += seg3(delta, 27);
}

// eskf.cc:31-45
StateVec State::operator-(const State& other) {
    StateVec delta;
    Mat3 rot_delta = other.rot_.T() * rot_;
    put3(delta, 0, LogSO3(rot_delta));
    put3(delta, 3, pos_ - other.pos_);
    put3(delta, 6, vel_ - other.vel_);
    put3(delta, 9, ba_ - other.ba_);
    put3(delta, 12, bw_ - other.bw_);
    put3(delta, 15, grav_ - other.grav_);
    put3(delta, 18, imu_a_ - other.imu_a_);
    put3(delta, 21, imu_w_ - other.imu_w_);
    put3(delta, 24, bv_ - other.bv_);
    put3(delta, 27, contact_ - other.contact_);
    return delta;
}

// eskf.cc:47-62
void ESKF::initProcessCovQ() {
    Q_ = StateCov::Zero();
    auto diag3 = [&](int o, double v) { Q_(o, o) = Q_(o + 1, o + 1) = Q_(o + 2, o + 2) = v; };
    diag3(6, config_.vel_process_cov);
    diag3(9, config_.acc_bias_process_cov);
    diag3(12, config_.gyr_bias_process_cov);
    diag3(18, config_.imu_acc_process_cov);
    diag3(21, config_.imu_gyr_process_cov);
    diag3(24, config_.kin_bias_process_cov);
    diag3(27, config_.contact_process_cov);
}

// eskf.cc:64-70
StateVec ESKF::getFunctionf(double dt) {
    StateVec vec = StateVec::Zero();
    put3(vec, 0, dt * state_.imu_w_);
    put3(vec, 3, dt * state_.vel_);
    put3(vec, 6, dt * (state_.rot_ * state_.imu_a_ + state_.grav_));
    return vec;
}

static inline void setblock(StateCov& F, int r, int c, const Mat3& b) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) F(r + i, c + j) = b(i, j);
}

// eskf.cc:72-81
StateCov ESKF::getFx(double dt) {
    StateCov Fx = StateCov::Identity();
    setblock(Fx, 0, 0, ExpVec((-dt) * state_.imu_w_));
    setblock(Fx, 0, 21, dt * Mat3::Identity());
    setblock(Fx, 3, 6, dt * Mat3::Identity());
    setblock(Fx, 6, 0, ((-dt) * state_.rot_) * skew(state_.imu_a_));
    setblock(Fx, 6, 15, dt * Mat3::Identity());
    setblock(Fx, 6, 18, dt * state_.rot_);
    return Fx;
}

// eskf.cc:83-89
void ESKF::predict(double dt, bool prop_state, bool prop_cov) {
    if (prop_state) { state_ += getFunctionf(dt); }
    if (prop_cov) {
        StateCov Fx = getFx(dt);
        cov_ = Fx * cov_ * Fx.T() + (dt * dt) * Q_;
    }
}

// eskf.cc:91-113
void ESKF::updateByPoints(ObsShared& obs) {
    int N = (int)obs.pt_z.size();
    if (N == 1 || N <= literal_max_n)
        updateByPointsLiteral(obs, N);
    else
        updateByPointsInfo6(obs, N);
}

void ESKF::updateByPointsLiteral(const ObsShared& obs, int N) {
    const std::vector<double>& z = obs.pt_z;
    const std::vector<double>& h = obs.pt_h;
    const std::vector<double>& r = obs.pt_R;
    if (N == 1) {
        // eskf.cc:98-104
        double PHT[DIM_STATE];
        for (int i = 0; i < DIM_STATE; ++i) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += cov_(i, k) * h[k];
            PHT[i] = s;
        }
        double hPHT = 0.0;
        for (int k = 0; k < 6; ++k) hPHT += h[k] * PHT[k];
        double HPHT_R_inv = 1 / (0.0001 + hPHT + r[0]);
        StateVec K, delta_x;
        for (int i = 0; i < DIM_STATE; ++i) K[i] = HPHT_R_inv * PHT[i];
        for (int i = 0; i < DIM_STATE; ++i) delta_x[i] = K[i] * z[0];
        state_ += delta_x;
        // cov_ = cov_ - K * h * cov_.block<6,30>(0,0)   ((K*h) is 30x6, then times 6x30)
        Mat<DIM_STATE, 6> Kh;
        for (int i = 0; i < DIM_STATE; ++i)
            for (int k = 0; k < 6; ++k) Kh(i, k) = K[i] * h[k];
        StateCov nc;
        for (int i = 0; i < DIM_STATE; ++i)
            for (int j = 0; j < DIM_STATE; ++j) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += Kh(i, k) * cov_(k, j);
                nc(i, j) = cov_(i, j) - s;
            }
        cov_ = nc;
        return;
    }
    // eskf.cc:105-112
    MatX H(N, 6), PHT(DIM_STATE, N);
    for (int k = 0; k < N; ++k)
        for (int c = 0; c < 6; ++c) H(k, c) = h[(size_t)k * 6 + c];
    for (int i = 0; i < DIM_STATE; ++i)
        for (int k = 0; k < N; ++k) {
            double s = 0.0;
            for (int c = 0; c < 6; ++c) s += cov_(i, c) * H(k, c);
            PHT(i, k) = s;
        }
    MatX HPHT_R(N, N);
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b) {
            double s = 0.0;
            for (int c = 0; c < 6; ++c) s += H(a, c) * PHT(c, b);
            HPHT_R(a, b) = s;
        }
    for (int a = 0; a < N; ++a) HPHT_R(a, a) += r[a];
    MatX K = matmul(PHT, inverse(HPHT_R));  // 30 x N
    StateVec delta_x;
    for (int i = 0; i < DIM_STATE; ++i) {
        double s = 0.0;
        for (int k = 0; k < N; ++k) s += K(i, k) * z[k];
        delta_x[i] = s;
    }
    state_ += delta_x;
    MatX KH = matmul(K, H);  // 30 x 6
    StateCov nc;
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += KH(i, k) * cov_(k, j);
            nc(i, j) = cov_(i, j) - s;
        }
    cov_ = nc;
}

// Equivalent of eskf.cc:105-112 via the matrix-inversion lemma (H has 6 non-zero columns):
//   A = h^T R^-1 h (6x6), b = h^T R^-1 z, M = (I6 + A P66)^-1,
//   dx = P[:,0:6] M b,   P <- P - P[:,0:6] (M A) P[0:6,:]
void ESKF::updateByPointsInfo6(const ObsShared& obs, int N) {
    Mat6 A = Mat6::Zero();
    Mat<6, 1> b = Mat<6, 1>::Zero();
    for (int k = 0; k < N; ++k) {
        const double* hk = &obs.pt_h[(size_t)k * 6];
        double rinv = 1.0 / obs.pt_R[k];
        for (int i = 0; i < 6; ++i) {
            double hi = hk[i] * rinv;
            for (int j = 0; j < 6; ++j) A(i, j) += hi * hk[j];
            b[i] += hi * obs.pt_z[k];
        }
    }
    Mat6 P66;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) P66(i, j) = cov_(i, j);
    Mat6 M = inverse(Mat6::Identity() + A * P66);
    Mat<6, 1> Mb = M * b;
    Mat6 MA = M * A;
    StateVec delta_x;
    for (int i = 0; i < DIM_STATE; ++i) {
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += cov_(i, k) * Mb[k];
        delta_x[i] = s;
    }
    state_ += delta_x;
    Mat<6, DIM_STATE> T;  // (M A) P[0:6,:]
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < DIM_STATE; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += MA(i, k) * cov_(k, j);
            T(i, j) = s;
        }
    StateCov nc;
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += cov_(i, k) * T(k, j);
            nc(i, j) = cov_(i, j) - s;
        }
    cov_ = nc;
}

// eskf.cc:125-135
void ESKF::updateByImu(ObsShared& obs) {
    Mat<DIM_STATE, 6> PHT;
    Mat<6, DIM_STATE> HP;
    for (int i = 0; i < DIM_STATE; ++i)
        for (int k = 0; k < 6; ++k) {
            PHT(i, k) = cov_(i, 9 + k) + cov_(i, 18 + k);
            HP(k, i) = cov_(9 + k, i) + cov_(18 + k, i);
        }
    Mat6 HPHT;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) HPHT(i, j) = PHT(9 + i, j) + PHT(18 + i, j);
    for (int i = 0; i < 6; ++i) HPHT(i, i) += obs.ki_R[i];
    Mat<DIM_STATE, 6> K = PHT * inverse(HPHT);
    Mat<6, 1> z;
    for (int i = 0; i < 6; ++i) z[i] = obs.ki_z[i];
    StateVec delta_x = K * z;
    state_ += delta_x;
    cov_ = cov_ - K * HP;
}

// eskf.cc:137-145
void ESKF::updateByKinImu(ObsShared& obs) {
    int M = (int)obs.ki_z.size();
    MatX H(M, DIM_STATE), HT(DIM_STATE, M), P(DIM_STATE, DIM_STATE);
    for (int a = 0; a < M; ++a)
        for (int j = 0; j < DIM_STATE; ++j) H(a, j) = HT(j, a) = obs.ki_h[(size_t)a * DIM_STATE + j];
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) P(i, j) = cov_(i, j);
    MatX PHT = matmul(P, HT);
    MatX HPHT = matmul(H, PHT);
    for (int a = 0; a < M; ++a) HPHT(a, a) += obs.ki_R[a];
    MatX K = matmul(PHT, inverse(HPHT));
    StateVec delta_x;
    for (int i = 0; i < DIM_STATE; ++i) {
        double s = 0.0;
        for (int a = 0; a < M; ++a) s += K(i, a) * obs.ki_z[a];
        delta_x[i] = s;
    }
    state_ += delta_x;
    MatX KH = matmul(K, H);
    MatX KHP = matmul(KH, P);
    for (int i = 0; i < DIM_STATE; ++i)
        for (int j = 0; j < DIM_STATE; ++j) cov_(i, j) = P(i, j) - KHP(i, j);
}

}  // namespace lko
