// oracle/ref_capi.cc — TEST INFRASTRUCTURE ONLY.
//
// A C API (lkr_*) over the REFERENCE's own classes legkilo::ESKF (eskf.h:46-109, eskf.cc) and
// legkilo::VoxelMapManager / VoxelOctoTree (voxel_map.h:129-244, voxel_map.cc), whose SOURCES are compiled unmodified
// from /root/reference by the `_ref` target of oracle/Makefile against the stand-in headers under oracle/shim
// (Eigen / PCL / ROS are absent from this image).  The signatures mirror the lko_* API of oracle_capi.cc one to one
// for the functions both have, so the same test code can drive the restatement (liblegkilo_oracle.so) and the
// reference (oracle/_ref/liblegkilo_ref.so) and compare them: tests/test_reference_pin.py.
//
// What this pins: every arithmetic routine of the hot path that lives in eskf.cc / voxel_map.cc.  What it does not:
// the glue of KILO::predictUpdatePoint (KILO.cc:108-233), which needs ROS/PCL/yaml-cpp/glog far beyond shells and
// stays pinned by the oracle's own tests only.
#include <cstring>
#include <memory>
#include <vector>

#include "core/slam/eskf.h"
#include "core/slam/voxel_map.h"
#include "export_blob.hpp"

using namespace legkilo;

struct lkr_handle {
    lk_config cfg;
    double acc_norm = 9.81, last_predict_t = 0, last_update_t = 0;  // KILO members; carried for API symmetry only
    std::unique_ptr<ESKF> eskf;
    std::unique_ptr<VoxelMapManager> map;
};

static void state_to_x36(const State& s, double* x) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) x[3 * i + j] = s.rot_(i, j);
    const Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
    for (int k = 0; k < 9; ++k)
        for (int c = 0; c < 3; ++c) x[9 + 3 * k + c] = (*v[k])[c];
}
static void x36_to_state(const double* x, State& s) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s.rot_(i, j) = x[3 * i + j];
    Vec3D* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
    for (int k = 0; k < 9; ++k)
        for (int c = 0; c < 3; ++c) (*v[k])[c] = x[9 + 3 * k + c];
}

extern "C" {

// the same mapping of lk_config onto ESKF::Config / VoxelMapConfig as KILO::initializeFromYaml (KILO.cc:47-79)
lkr_handle* lkr_create(const lk_config* c, int /*imu_mode_only*/) {
    lkr_handle* h = new lkr_handle;
    h->cfg = *c;
    ESKF::Config e;
    e.vel_process_cov = c->vel_process_cov;
    e.imu_acc_process_cov = c->imu_acc_process_cov;
    e.imu_gyr_process_cov = c->imu_gyr_process_cov;
    e.contact_process_cov = c->contact_process_cov;
    e.acc_bias_process_cov = c->acc_bias_process_cov;
    e.gyr_bias_process_cov = c->gyr_bias_process_cov;
    e.kin_bias_process_cov = c->kin_bias_process_cov;
    e.imu_acc_meas_noise = c->imu_acc_meas_noise;
    e.imu_acc_z_meas_noise = c->imu_acc_z_meas_noise;
    e.imu_gyr_meas_noise = c->imu_gyr_meas_noise;
    e.kin_meas_noise = c->kin_meas_noise;
    e.chd_meas_noise = c->chd_meas_noise;
    e.contact_meas_noise = c->contact_meas_noise;
    e.lidar_point_meas_ratio = c->lidar_point_meas_ratio;
    h->eskf = std::make_unique<ESKF>(e);
    h->eskf->cov().setZero();
    h->eskf->Q().setZero();
    VoxelMapConfig v;
    v.max_voxel_size_ = c->max_voxel_size;
    v.max_layer_ = c->max_layer;
    v.max_iterations_ = c->max_iterations;
    v.layer_init_num_.assign(c->layer_init_num, c->layer_init_num + 5);
    v.max_points_num_ = c->max_points_num;
    v.planner_threshold_ = c->planner_threshold;
    v.beam_err_ = c->beam_err;
    v.dept_err_ = c->dept_err;
    v.sigma_num_ = c->sigma_num;
    v.is_pub_plane_map_ = false;
    v.sliding_thresh = 8;
    v.map_sliding_en = false;
    v.half_map_size = 100;
    h->map = std::make_unique<VoxelMapManager>(v);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) h->map->extR_(i, j) = c->ext_R[3 * i + j];
        h->map->extT_[i] = c->ext_T[i];
    }
    return h;
}
void lkr_destroy(lkr_handle* h) {
    if (!h) return;
    for (auto& kv : h->map->voxel_map_) delete kv.second;  // the reference never frees its trees
    delete h;
}

int lkr_set_state(lkr_handle* h, const double* x36, const double* P900) {
    if (x36) x36_to_state(x36, h->eskf->state());
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) h->eskf->cov()(i, j) = P900[30 * i + j];
    return 0;
}
int lkr_get_state(lkr_handle* h, double* x36, double* P900) {
    if (x36) state_to_x36(h->eskf->state(), x36);
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) P900[30 * i + j] = h->eskf->cov()(i, j);
    return 0;
}
int lkr_set_Q(lkr_handle* h, const double* Q900) {
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) h->eskf->Q()(i, j) = Q900[30 * i + j];
    return 0;
}
int lkr_get_Q(lkr_handle* h, double* Q900) {
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) Q900[30 * i + j] = h->eskf->Q()(i, j);
    return 0;
}
int lkr_init_process_cov_q(lkr_handle* h) {
    h->eskf->initProcessCovQ();
    return 0;
}
int lkr_set_times(lkr_handle* h, double last_predict_t, double last_update_t) {
    h->last_predict_t = last_predict_t, h->last_update_t = last_update_t;
    return 0;
}
int lkr_get_times(lkr_handle* h, double* last_predict_t, double* last_update_t) {
    *last_predict_t = h->last_predict_t, *last_update_t = h->last_update_t;
    return 0;
}
int lkr_set_acc_norm(lkr_handle* h, double a) {
    h->acc_norm = a;
    return 0;
}
int lkr_get_fx(lkr_handle* h, double dt, double* Fx900) {
    StateF F = h->eskf->getFx(dt);
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) Fx900[30 * i + j] = F(i, j);
    return 0;
}
int lkr_get_function_f(lkr_handle* h, double dt, double* f30) {
    StateVec f = h->eskf->getFunctionf(dt);
    for (int i = 0; i < 30; ++i) f30[i] = f[i];
    return 0;
}
int lkr_predict(lkr_handle* h, double dt, int prop_state, int prop_cov) {
    h->eskf->predict(dt, prop_state != 0, prop_cov != 0);
    return 0;
}
// State::operator- (eskf.cc:31-45): delta30 = state(a) - state(b)
int lkr_state_minus(const double* xa36, const double* xb36, double* delta30) {
    State a, b;
    x36_to_state(xa36, a);
    x36_to_state(xb36, b);
    StateVec d = a - b;
    for (int i = 0; i < 30; ++i) delta30[i] = d[i];
    return 0;
}
int lkr_update_by_points(lkr_handle* h, const double* h6, const double* z, const double* R, size_t N) {
    ObsShared o;
    o.pt_h.resize((Eigen::Index)N, 6);
    o.pt_z.resize((Eigen::Index)N);
    o.pt_R.resize((Eigen::Index)N);
    for (size_t i = 0; i < N; ++i) {
        for (int c = 0; c < 6; ++c) o.pt_h((Eigen::Index)i, c) = h6[6 * i + c];
        o.pt_z((Eigen::Index)i) = z[i];
        o.pt_R((Eigen::Index)i) = R[i];
    }
    h->eskf->updateByPoints(o);
    return 0;
}
int lkr_update_by_imu(lkr_handle* h, const double* z6, const double* R6) {
    ObsShared o;
    o.ki_z.resize(6);
    o.ki_R.resize(6);
    for (int i = 0; i < 6; ++i) o.ki_z(i) = z6[i], o.ki_R(i) = R6[i];
    h->eskf->updateByImu(o);
    return 0;
}
int lkr_update_by_kin_imu(lkr_handle* h, const double* ki_h, const double* ki_z, const double* ki_R, size_t M) {
    ObsShared o;
    o.ki_h.resize((Eigen::Index)M, 30);
    o.ki_z.resize((Eigen::Index)M);
    o.ki_R.resize((Eigen::Index)M);
    for (size_t i = 0; i < M; ++i) {
        for (int c = 0; c < 30; ++c) o.ki_h((Eigen::Index)i, c) = ki_h[30 * i + c];
        o.ki_z((Eigen::Index)i) = ki_z[i];
        o.ki_R((Eigen::Index)i) = ki_R[i];
    }
    h->eskf->updateByKinImu(o);
    return 0;
}

// BuildVoxelMap(rot, rot_cov, pos_cov) as called at KILO.cc:339
int lkr_map_build(lkr_handle* h, const float* xyz_world, const float* xyz_body, size_t n) {
    auto& m = *h->map;
    m.feats_down_world_->clear();
    m.feats_down_body_->clear();
    for (size_t i = 0; i < n; ++i) {
        PointType pw, pb;
        pw.x = xyz_world[3 * i], pw.y = xyz_world[3 * i + 1], pw.z = xyz_world[3 * i + 2];
        pb.x = xyz_body[3 * i], pb.y = xyz_body[3 * i + 1], pb.z = xyz_body[3 * i + 2];
        m.feats_down_world_->push_back(pw);
        m.feats_down_body_->push_back(pb);
    }
    m.BuildVoxelMap(h->eskf->getRot(), h->eskf->getRotCov(), h->eskf->getPosCov());
    return 0;
}
int lkr_map_update(lkr_handle* h, const double* pw, const double* var9, size_t n) {
    std::vector<pointWithVar> pv(n);
    for (size_t i = 0; i < n; ++i) {
        pv[i].point_w = Eigen::Vector3d(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) pv[i].var(r, c) = var9[9 * i + 3 * r + c];
    }
    h->map->UpdateVoxelMap(pv);
    return 0;
}
int lkr_map_export(lkr_handle* h, void* blob, size_t* bytes) {
    auto& m = *h->map;
    return lkx::export_map<VoxelOctoTree, VoxelPlane, pointWithVar>(m.voxel_map_, m.config_setting_.max_voxel_size_,
                                                                   m.config_setting_.max_layer_,
                                                                   m.config_setting_.max_points_num_, blob, bytes);
}
int lkr_map_stats(lkr_handle* h, uint32_t* n_roots) {
    *n_roots = (uint32_t)h->map->voxel_map_.size();
    return 0;
}
int lkr_map_slide(lkr_handle* h, const double* position3, double sliding_thresh, int half_map_size, int* slid, uint32_t* n_removed) {
    auto& m = *h->map;
    m.config_setting_.sliding_thresh = sliding_thresh;
    m.config_setting_.half_map_size = half_map_size;
    m.position_last_ = Eigen::Vector3d(position3[0], position3[1], position3[2]);
    const size_t before = m.voxel_map_.size();
    const bool s = m.mapSliding();
    if (slid) *slid = s ? 1 : 0;
    if (n_removed) *n_removed = (uint32_t)(before - m.voxel_map_.size());
    return 0;
}
int lkr_map_clear_outside(lkr_handle* h, int x_max, int x_min, int y_max, int y_min, int z_max, int z_min, uint32_t* n_removed) {
    const size_t before = h->map->voxel_map_.size();
    h->map->clearMemOutOfMap(x_max, x_min, y_max, y_min, z_max, z_min);
    if (n_removed) *n_removed = (uint32_t)(before - h->map->voxel_map_.size());
    return 0;
}
int lkr_map_slide_position(lkr_handle* h, int set, double* last3) {
    for (int i = 0; i < 3; ++i) {
        if (set) h->map->last_slide_position[i] = last3[i];
        else last3[i] = h->map->last_slide_position[i];
    }
    return 0;
}

// build_single_residual (voxel_map.cc:363-427) on the root voxel `key3`, started like KILO.cc:150-155
// (is_success = false, prob = 0).  found = 0 when the voxel does not exist.
int lkr_match_voxel(lkr_handle* h, const int* key3, const double* pw3, const double* var9, int* found, int* success,
                    double* prob, double* normal3, double* center3, double* d, float* dis_to_plane, int* layer) {
    auto& m = *h->map;
    Eigen::Vector3i key(key3[0], key3[1], key3[2]);
    auto it = m.voxel_map_.find(key);
    *found = it != m.voxel_map_.end();
    *success = 0, *prob = 0;
    if (!*found) return 0;
    pointWithVar pv;
    pv.point_w = Eigen::Vector3d(pw3[0], pw3[1], pw3[2]);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) pv.var(r, c) = var9[3 * r + c];
    PointToPlane pl;
    bool ok = false;
    double pr = 0;
    m.build_single_residual(pv, it->second, 0, ok, pr, pl);
    *success = ok ? 1 : 0;
    *prob = pr;
    if (ok) {
        for (int c = 0; c < 3; ++c) normal3[c] = pl.normal_[c], center3[c] = pl.center_[c];
        *d = pl.d_;
        *dis_to_plane = pl.dis_to_plane_;
        *layer = pl.layer_;
    }
    return 0;
}

// ---- unit-level hooks ----
int lkr_calc_body_cov(const double* pb3, float range_inc, float degree_inc, double* cov9) {
    Eigen::Vector3d pb(pb3[0], pb3[1], pb3[2]);
    Eigen::Matrix3d cov;
    calcBodyCov(pb, range_inc, degree_inc, cov);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov9[3 * i + j] = cov(i, j);
    return 0;
}
int lkr_init_plane(const double* pw, const double* var9, size_t n, float planer_threshold, lk_plane_rec* out,
                   double* plane_var36) {
    VoxelOctoTree t(2, 0, 5, 50, planer_threshold);
    std::vector<pointWithVar> pts(n);
    for (size_t i = 0; i < n; ++i) {
        pts[i].point_w = Eigen::Vector3d(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) pts[i].var(r, c) = var9[9 * i + 3 * r + c];
    }
    t.init_plane(pts, t.plane_ptr_);
    const VoxelPlane& pl = *t.plane_ptr_;
    std::memset(out, 0, sizeof(*out));
    for (int c = 0; c < 3; ++c) out->center[c] = pl.center_[c], out->normal[c] = pl.normal_[c];
    out->d = pl.d_;
    out->radius = pl.radius_;
    out->flags = pl.is_plane_ ? LK_PLANE_IS_PLANE : 0u;
    out->points_size = pl.points_size_;
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) out->plane_var[k++] = pl.plane_var_(r, c);
    out->min_eigen_value = pl.min_eigen_value_;
    out->mid_eigen_value = pl.mid_eigen_value_;
    out->max_eigen_value = pl.max_eigen_value_;
    if (plane_var36)
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) plane_var36[6 * r + c] = pl.plane_var_(r, c);
    return 0;
}
int lkr_exp_log(const double* v3, double* R9_exp3, double* R9_expvec, double* log3) {
    Eigen::Matrix3d a = Exp(v3[0], v3[1], v3[2]);
    Eigen::Matrix3d b = Exp(Eigen::Matrix<double, 3, 1>(Eigen::Vector3d(v3[0], v3[1], v3[2])));
    Eigen::Vector3d l = Log(a);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R9_exp3[3 * i + j] = a(i, j), R9_expvec[3 * i + j] = b(i, j);
        log3[i] = l[i];
    }
    return 0;
}
size_t lkr_hash_vec3(int x, int y, int z) { return hash_vec<3>()(Eigen::Vector3i(x, y, z)); }
int lkr_key_floor(const double* p3, double voxel_size, int* key3) {
    Eigen::Vector3i k = voxelKeyFloor(Eigen::Vector3d(p3[0], p3[1], p3[2]), voxel_size);
    for (int i = 0; i < 3; ++i) key3[i] = k[i];
    return 0;
}

}  // extern "C"
