// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
// Restatement of legkilo/src/core/slam/KILO.cc:86-399 and preprocess/state_initial.hpp:34-118.
#include "oracle_kilo.hpp"

namespace lko {

// KILO.cc:25-84 — the YAML values arrive already parsed in lk_config
KILO::KILO(const lk_config& c) {
    ESKF::Config e;
    e.vel_process_cov = c.vel_process_cov;
    e.imu_acc_process_cov = c.imu_acc_process_cov;
    e.imu_gyr_process_cov = c.imu_gyr_process_cov;
    e.contact_process_cov = c.contact_process_cov;
    e.acc_bias_process_cov = c.acc_bias_process_cov;
    e.gyr_bias_process_cov = c.gyr_bias_process_cov;
    e.kin_bias_process_cov = c.kin_bias_process_cov;
    e.imu_acc_meas_noise = c.imu_acc_meas_noise;
    e.imu_acc_z_meas_noise = c.imu_acc_z_meas_noise;
    e.imu_gyr_meas_noise = c.imu_gyr_meas_noise;
    e.kin_meas_noise = c.kin_meas_noise;
    e.chd_meas_noise = c.chd_meas_noise;
    e.contact_meas_noise = c.contact_meas_noise;
    e.lidar_point_meas_ratio = c.lidar_point_meas_ratio;
    eskf_ = std::make_unique<ESKF>(e);
    gravity_ = c.gravity;

    VoxelMapConfig v;
    v.max_layer_ = c.max_layer;
    v.max_iterations_ = c.max_iterations;
    v.max_voxel_size_ = c.max_voxel_size;
    v.planner_threshold_ = c.planner_threshold;
    v.sigma_num_ = c.sigma_num;
    v.beam_err_ = c.beam_err;
    v.dept_err_ = c.dept_err;
    v.layer_init_num_.assign(c.layer_init_num, c.layer_init_num + 5);
    v.max_points_num_ = c.max_points_num;
    map_manager_ = std::make_unique<VoxelMapManager>(v);

    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) ext_rot_(i, j) = c.ext_R[3 * i + j];
        ext_t_[i] = c.ext_T[i];
    }
    map_manager_->extT_ = ext_t_;
    map_manager_->extR_ = ext_rot_;
}

// KILO.cc:122-183, one point
bool KILO::matchPoint(const lk_point& cur_pt, pointWithVar& cur_pt_var, PointToPlane& single_ptpl, float* w) {
    // 2.1 point var (body and world)
    cur_pt_var.point_b = vec3(cur_pt.x, cur_pt.y, cur_pt.z);
    cur_pt_var.point_i = ext_rot_ * cur_pt_var.point_b + ext_t_;
    cur_pt_var.point_w = eskf_->getRot() * cur_pt_var.point_i + eskf_->getPos();
    if (w) {
        w[0] = (float)cur_pt_var.point_w[0];
        w[1] = (float)cur_pt_var.point_w[1];
        w[2] = (float)cur_pt_var.point_w[2];
        w[3] = 0;
    }
    calcBodyCov(cur_pt_var.point_b, map_manager_->config_setting_.dept_err_, map_manager_->config_setting_.beam_err_,
                cur_pt_var.body_var);
    cur_pt_var.point_crossmat = skew(cur_pt_var.point_i);
    Mat3 rot_extR = eskf_->getRot() * ext_rot_;
    Mat3 rot_crossmat = eskf_->getRot() * cur_pt_var.point_crossmat;
    cur_pt_var.var = (rot_extR * cur_pt_var.body_var) * rot_extR.T() +
                     (rot_crossmat * eskf_->getRotCov()) * rot_crossmat.T() + eskf_->getPosCov();

    // 2.2 residual
    float loc_xyz[3];
    for (int j = 0; j < 3; j++) {
        loc_xyz[j] = cur_pt_var.point_w[j] / map_manager_->config_setting_.max_voxel_size_;
        if (loc_xyz[j] < 0) { loc_xyz[j] -= 1.0; }
    }
    Vec3i position{(int)loc_xyz[0], (int)loc_xyz[1], (int)loc_xyz[2]};
    auto iter = map_manager_->voxel_map_.find(position);
    if (iter == map_manager_->voxel_map_.end()) return false;
    VoxelOctoTree* current_octo = iter->second;
    bool is_success = false;
    double prob = 0;
    map_manager_->build_single_residual(cur_pt_var, current_octo, 0, is_success, prob, single_ptpl);
    if (!is_success) {
        Vec3i near_position = position;
        for (int j = 0; j < 3; ++j) {  // KILO.cc:158-172 (voxel-units vs metres comparison kept as is)
            if (loc_xyz[j] > (current_octo->voxel_center_[j] + current_octo->quater_length_)) {
                near_position[j] = near_position[j] + 1;
            } else if (loc_xyz[j] < (current_octo->voxel_center_[j] - current_octo->quater_length_)) {
                near_position[j] = near_position[j] - 1;
            }
        }
        auto iter_near = map_manager_->voxel_map_.find(near_position);
        if (iter_near != map_manager_->voxel_map_.end()) {
            map_manager_->build_single_residual(cur_pt_var, iter_near->second, 0, is_success, prob, single_ptpl);
        }
    }
    return is_success;
}

// KILO.cc:195-209
void KILO::rowFromPtpl(const PointToPlane& p, double* h6, double& z, double& R) {
    Vec3 crossmat_rotT_u = (p.point_crossmat_ * eskf_->getRot().T()) * p.normal_;
    for (int c = 0; c < 3; ++c) h6[c] = crossmat_rotT_u[c], h6[3 + c] = p.normal_[c];
    z = -p.dis_to_plane_;
    Mat<1, 6> J_nq;
    for (int c = 0; c < 3; ++c) J_nq(0, c) = p.point_w_[c] - p.center_[c], J_nq(0, 3 + c) = -p.normal_[c];
    Mat3 var = (((eskf_->getRot() * ext_rot_) * p.body_cov_) * ext_rot_.T()) * eskf_->getRot().T();
    double single_l = ((J_nq * p.plane_var_) * J_nq.T())(0, 0);
    R = eskf_->config().lidar_point_meas_ratio * (single_l + ((p.normal_.T() * var) * p.normal_)(0, 0));
}

// KILO.cc:108-233
bool KILO::predictUpdatePoint(double current_time, size_t idx_i, size_t idx_j, const lk_point* cloud_down_body,
                              float* cloud_down_world, size_t& success_pts_size_out) {
    // 1) Predict state
    double dt_cov = current_time - last_state_update_time_;
    eskf_->predict(dt_cov, false, true);
    double dt = current_time - last_state_predict_time_;
    eskf_->predict(dt, true, false);
    last_state_predict_time_ = current_time;

    // 2) Residuals
    size_t points_size = idx_j - idx_i;
    std::vector<PointToPlane> ptpl_list;
    std::vector<pointWithVar> pv_list(points_size);
    ptpl_list.reserve(points_size);
    for (size_t i = 0; i < points_size; ++i) {
        PointToPlane single_ptpl;
        float* w = cloud_down_world ? cloud_down_world + 4 * (idx_i + i) : nullptr;
        if (matchPoint(cloud_down_body[i + idx_i], pv_list[i], single_ptpl, w)) {
            ++success_pts_size_out;
            ptpl_list.push_back(single_ptpl);
        }
    }

    // 3) KF update with points
    size_t effect_num = ptpl_list.size();
    bool eskf_update = effect_num > 0;
    if (eskf_update) {
        ObsShared obs_shared;
        obs_shared.pt_h.resize(effect_num * 6);
        obs_shared.pt_R.resize(effect_num);
        obs_shared.pt_z.resize(effect_num);
        for (size_t k = 0; k < effect_num; ++k)
            rowFromPtpl(ptpl_list[k], &obs_shared.pt_h[6 * k], obs_shared.pt_z[k], obs_shared.pt_R[k]);
        eskf_->updateByPoints(obs_shared);
        last_state_update_time_ = current_time;
    }

    // 4) voxel map update
    if (eskf_update) {
        for (size_t i = 0; i < points_size; ++i) {
            pv_list[i].point_w = eskf_->getRot() * pv_list[i].point_i + eskf_->getPos();
            if (cloud_down_world) {
                float* w = cloud_down_world + 4 * (idx_i + i);
                w[0] = (float)pv_list[i].point_w[0];
                w[1] = (float)pv_list[i].point_w[1];
                w[2] = (float)pv_list[i].point_w[2];
                w[3] = 255;
            }
            Mat3 rot_extR = eskf_->getRot() * ext_rot_;
            Mat3 rot_crossmat = eskf_->getRot() * pv_list[i].point_crossmat;
            pv_list[i].var = (rot_extR * pv_list[i].body_var) * rot_extR.T() +
                             (rot_crossmat * eskf_->getRotCov()) * rot_crossmat.T() + eskf_->getPosCov();
        }
    }
    if (map_insert_enabled_) map_manager_->UpdateVoxelMap(pv_list);
    return effect_num > 0;
}

void KILO::residualsOnly(const float* xyz_body, size_t n, std::vector<ResidualRow>& rows) {
    rows.resize(n);
    for (size_t i = 0; i < n; ++i) {
        lk_point pt{xyz_body[3 * i], xyz_body[3 * i + 1], xyz_body[3 * i + 2], 0.f};
        pointWithVar pv;
        PointToPlane ptpl;
        ResidualRow& r = rows[i];
        r.valid = matchPoint(pt, pv, ptpl, nullptr);
        if (r.valid) {
            rowFromPtpl(ptpl, r.h, r.z, r.R);
            r.layer = ptpl.layer_;
        } else {
            for (int c = 0; c < 6; ++c) r.h[c] = 0;
            r.z = r.R = 0;
            r.layer = -1;
        }
    }
}

// KILO.cc:235-258
bool KILO::predictUpdateImu(const lk_imu& imu) {
    double current_time = imu.stamp;
    double dt_cov = current_time - last_state_update_time_;
    eskf_->predict(dt_cov, false, true);
    double dt = current_time - last_state_predict_time_;
    eskf_->predict(dt, true, false);
    last_state_predict_time_ = current_time;

    ObsShared obs_shared;
    obs_shared.ki_R.resize(6);
    obs_shared.ki_z.resize(6);
    Vec3 imu_acc = vec3(imu.acc[0], imu.acc[1], imu.acc[2]);
    Vec3 imu_gyr = vec3(imu.gyr[0], imu.gyr[1], imu.gyr[2]);
    Vec3 za = (gravity_ / acc_norm_) * imu_acc - eskf_->state().imu_a_ - eskf_->state().ba_;
    Vec3 zw = imu_gyr - eskf_->state().imu_w_ - eskf_->state().bw_;
    for (int i = 0; i < 3; ++i) obs_shared.ki_z[i] = za[i], obs_shared.ki_z[3 + i] = zw[i];
    const auto& c = eskf_->config();
    double R6[6] = {c.imu_acc_meas_noise, c.imu_acc_meas_noise, c.imu_acc_z_meas_noise,
                    c.imu_gyr_meas_noise, c.imu_gyr_meas_noise, c.imu_gyr_meas_noise};
    obs_shared.ki_R.assign(R6, R6 + 6);
    eskf_->updateByImu(obs_shared);
    last_state_update_time_ = current_time;
    return true;
}

// KILO.cc:260-314
bool KILO::predictUpdateKinImu(const lk_kin_imu& kin_imu) {
    double current_time = kin_imu.time_stamp;
    double dt_cov = current_time - last_state_update_time_;
    eskf_->predict(dt_cov, false, true);
    double dt = current_time - last_state_predict_time_;
    eskf_->predict(dt, true, false);
    last_state_predict_time_ = current_time;

    int contact_nums = 0;
    for (int i = 0; i < 4; ++i)
        if (kin_imu.contact[i]) contact_nums++;

    int M = 6 + 3 * contact_nums;
    ObsShared obs_shared;
    obs_shared.ki_R.assign(M, 0.0);
    obs_shared.ki_z.assign(M, 0.0);
    obs_shared.ki_h.assign((size_t)M * DIM_STATE, 0.0);
    auto H = [&](int r, int c) -> double& { return obs_shared.ki_h[(size_t)r * DIM_STATE + c]; };
    for (int i = 0; i < 6; ++i) H(i, 9 + i) = 1.0, H(i, 18 + i) = 1.0;

    Vec3 imu_acc = vec3(kin_imu.acc[0], kin_imu.acc[1], kin_imu.acc[2]);
    Vec3 imu_gyr = vec3(kin_imu.gyr[0], kin_imu.gyr[1], kin_imu.gyr[2]);
    Vec3 za = (gravity_ / acc_norm_) * imu_acc - eskf_->state().imu_a_ - eskf_->state().ba_;
    Vec3 zw = imu_gyr - eskf_->state().imu_w_ - eskf_->state().bw_;
    for (int i = 0; i < 3; ++i) obs_shared.ki_z[i] = za[i], obs_shared.ki_z[3 + i] = zw[i];
    const auto& c = eskf_->config();
    double R6[6] = {c.imu_acc_meas_noise, c.imu_acc_meas_noise, c.imu_acc_z_meas_noise,
                    c.imu_gyr_meas_noise, c.imu_gyr_meas_noise, c.imu_gyr_meas_noise};
    for (int i = 0; i < 6; ++i) obs_shared.ki_R[i] = R6[i];

    int idx = 0;
    Mat3 w_skew = skew(eskf_->state().imu_w_);
    for (int i = 0; i < 4; ++i) {
        if (kin_imu.contact[i]) {
            Vec3 foot_pos = vec3(kin_imu.foot_pos[i][0], kin_imu.foot_pos[i][1], kin_imu.foot_pos[i][2]);
            Vec3 foot_vel = vec3(kin_imu.foot_vel[i][0], kin_imu.foot_vel[i][1], kin_imu.foot_vel[i][2]);
            Vec3 w_skew_pos_vel = w_skew * foot_pos + foot_vel;
            Mat3 b0 = (-eskf_->getRot()) * skew(w_skew_pos_vel);
            Mat3 b21 = (-eskf_->getRot()) * skew(foot_pos);
            for (int r = 0; r < 3; ++r)
                for (int cc = 0; cc < 3; ++cc) {
                    H(6 + 3 * idx + r, 0 + cc) = b0(r, cc);
                    H(6 + 3 * idx + r, 6 + cc) = (r == cc) ? 1.0 : 0.0;
                    H(6 + 3 * idx + r, 21 + cc) = b21(r, cc);
                }
            Vec3 zk = -eskf_->getVel() - eskf_->getRot() * w_skew_pos_vel;
            for (int r = 0; r < 3; ++r) {
                obs_shared.ki_z[6 + 3 * idx + r] = zk[r];
                obs_shared.ki_R[6 + 3 * idx + r] = c.kin_meas_noise;
            }
            idx++;
        }
    }
    eskf_->updateByKinImu(obs_shared);
    last_state_update_time_ = current_time;
    return true;
}

// KILO.cc:332-353 + state_initial.hpp:34-72 / :79-117
void KILO::firstFrame(const lk_point* cloud_raw, size_t n, double end_time, const lk_imu* imus, size_t n_imu,
                      const lk_kin_imu* kins, size_t n_kin) {
    // StateInitial::processing (b_first_frame_ branch + running mean)
    int N = 1;
    Vec3 mean_acc, mean_gyr;
    size_t cnt = imu_mode_only_ ? n_imu : n_kin;
    auto acc_of = [&](size_t k) {
        return imu_mode_only_ ? vec3(imus[k].acc[0], imus[k].acc[1], imus[k].acc[2])
                              : vec3(kins[k].acc[0], kins[k].acc[1], kins[k].acc[2]);
    };
    auto gyr_of = [&](size_t k) {
        return imu_mode_only_ ? vec3(imus[k].gyr[0], imus[k].gyr[1], imus[k].gyr[2])
                              : vec3(kins[k].gyr[0], kins[k].gyr[1], kins[k].gyr[2]);
    };
    mean_acc = acc_of(0);
    mean_gyr = gyr_of(0);
    for (size_t k = 0; k < cnt; ++k) {
        Vec3 cur_acc = acc_of(k), cur_gyr = gyr_of(k);
        mean_acc += (cur_acc - mean_acc) / (double)N;
        mean_gyr += (cur_gyr - mean_gyr) / (double)N;
        N++;
    }
    acc_norm_ = norm(mean_acc);  // getAccNorm(), KILO.cc:349
    eskf_->state().grav_ = ((-mean_acc) / acc_norm_) * gravity_;
    eskf_->state().bw_ = mean_gyr;
    eskf_->state().rot_ = Mat3::Identity();
    eskf_->cov() = 0.000001 * StateCov::Identity();
    eskf_->initProcessCovQ();

    // cloudLidarToWorld (KILO.cc:89-106) on the RAW cloud, then BuildVoxelMap
    map_manager_->feats_down_body_.resize(3 * n);
    map_manager_->feats_down_world_.resize(3 * n);
    for (size_t i = 0; i < n; ++i) {
        Vec3 pt_lidar = vec3(cloud_raw[i].x, cloud_raw[i].y, cloud_raw[i].z);
        Vec3 pt_imu = ext_rot_ * pt_lidar + ext_t_;
        Vec3 pt_world = eskf_->getRot() * pt_imu + eskf_->getPos();
        for (int c = 0; c < 3; ++c) {
            map_manager_->feats_down_body_[3 * i + c] = (float)pt_lidar[c];
            map_manager_->feats_down_world_[3 * i + c] = (float)pt_world[c];
        }
    }
    map_manager_->BuildVoxelMap(eskf_->getRot(), eskf_->getRotCov(), eskf_->getPosCov());
    last_state_predict_time_ = end_time;
    last_state_update_time_ = end_time;
}

// KILO.cc:367-396 minus the std::sort at :370 (input is already time-sorted)
void KILO::processSorted(const lk_point* pts, size_t pts_size, double begin_time, std::deque<lk_imu>& imus,
                         std::deque<lk_kin_imu>& kin_imus, float* cloud_down_world, size_t& success_pts_size_out,
                         uint32_t* n_buckets, uint32_t* n_updates) {
    size_t idx_i = 0;
    uint32_t nb = 0, nu = 0;
    while (idx_i < pts_size) {
        double cur_point_time = begin_time + pts[idx_i].curvature;
        size_t idx_j = idx_i + 1;
        while (idx_j < pts_size && pts[idx_i].curvature == pts[idx_j].curvature) { idx_j++; }

        if (imu_mode_only_) {
            while (!imus.empty() && imus.front().stamp < cur_point_time) {
                predictUpdateImu(imus.front());
                imus.pop_front();
            }
        } else {
            while (!kin_imus.empty() && kin_imus.front().time_stamp < cur_point_time) {
                predictUpdateKinImu(kin_imus.front());
                kin_imus.pop_front();
            }
        }
        bool upd = predictUpdatePoint(cur_point_time, idx_i, idx_j, pts, cloud_down_world, success_pts_size_out);
        nb++;
        nu += upd ? 1 : 0;
        idx_i = idx_j;
    }
    if (n_buckets) *n_buckets = nb;
    if (n_updates) *n_updates = nu;
}

}  // namespace lko
