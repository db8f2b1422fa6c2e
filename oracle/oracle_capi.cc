// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY PINNED through oracle/_ref (see smallmat.hpp header).
// extern "C" surface of liblegkilo_oracle.so, loaded with ctypes by tests/, by
// __graft_entry__.smoke() and by bench.py's cpu_baseline leg — nowhere else.
// Mirrors the lk_* calls of include/legkilo_hip.h with an lko_ prefix so a parity
// test is the same call sequence on both sides.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "export_blob.hpp"
#include "oracle_kilo.hpp"

using namespace lko;

struct lko_handle {
    lk_config cfg;
    std::unique_ptr<KILO> kilo;
    std::string err;
};

static void state_to_x36(const State& s, double* x) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) x[3 * i + j] = s.rot_(i, j);
    const Vec3* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
    for (int k = 0; k < 9; ++k)
        for (int c = 0; c < 3; ++c) x[9 + 3 * k + c] = (*v[k])[c];
}
static void x36_to_state(const double* x, State& s) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s.rot_(i, j) = x[3 * i + j];
    Vec3* v[9] = {&s.pos_, &s.vel_, &s.ba_, &s.bw_, &s.grav_, &s.imu_a_, &s.imu_w_, &s.bv_, &s.contact_};
    for (int k = 0; k < 9; ++k)
        for (int c = 0; c < 3; ++c) (*v[k])[c] = x[9 + 3 * k + c];
}

extern "C" {

lko_handle* lko_create(const lk_config* cfg, int imu_mode_only) {
    lko_handle* h = new lko_handle;
    h->cfg = *cfg;
    h->kilo = std::make_unique<KILO>(*cfg);
    h->kilo->imu_mode_only_ = imu_mode_only != 0;
    return h;
}
void lko_destroy(lko_handle* h) { delete h; }

int lko_set_state(lko_handle* h, const double* x36, const double* P900) {
    if (x36) x36_to_state(x36, h->kilo->eskf_->state());
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) h->kilo->eskf_->cov()(i, j) = P900[30 * i + j];
    return 0;
}
int lko_get_state(lko_handle* h, double* x36, double* P900) {
    if (x36) state_to_x36(h->kilo->eskf_->state(), x36);
    if (P900)
        for (int i = 0; i < 30; ++i)
            for (int j = 0; j < 30; ++j) P900[30 * i + j] = h->kilo->eskf_->cov()(i, j);
    return 0;
}
int lko_set_Q(lko_handle* h, const double* Q900) {
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) h->kilo->eskf_->Q()(i, j) = Q900[30 * i + j];
    return 0;
}
int lko_get_Q(lko_handle* h, double* Q900) {
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) Q900[30 * i + j] = h->kilo->eskf_->Q()(i, j);
    return 0;
}
int lko_init_process_cov_q(lko_handle* h) {
    h->kilo->eskf_->initProcessCovQ();
    return 0;
}
int lko_set_times(lko_handle* h, double last_predict_t, double last_update_t) {
    h->kilo->last_state_predict_time_ = last_predict_t;
    h->kilo->last_state_update_time_ = last_update_t;
    return 0;
}
int lko_get_times(lko_handle* h, double* last_predict_t, double* last_update_t) {
    *last_predict_t = h->kilo->last_state_predict_time_;
    *last_update_t = h->kilo->last_state_update_time_;
    return 0;
}
int lko_set_acc_norm(lko_handle* h, double a) {
    h->kilo->acc_norm_ = a;
    return 0;
}
double lko_get_acc_norm(lko_handle* h) { return h->kilo->acc_norm_; }
int lko_set_literal_max_n(lko_handle* h, int n) {
    h->kilo->eskf_->literal_max_n = n;
    return 0;
}
int lko_set_map_insert(lko_handle* h, int on) {
    h->kilo->map_insert_enabled_ = on != 0;
    return 0;
}
int lko_get_fx(lko_handle* h, double dt, double* Fx900) {
    StateCov F = h->kilo->eskf_->getFx(dt);
    for (int i = 0; i < 30; ++i)
        for (int j = 0; j < 30; ++j) Fx900[30 * i + j] = F(i, j);
    return 0;
}
int lko_get_function_f(lko_handle* h, double dt, double* f30) {
    StateVec f = h->kilo->eskf_->getFunctionf(dt);
    for (int i = 0; i < 30; ++i) f30[i] = f[i];
    return 0;
}
int lko_predict(lko_handle* h, double dt, int prop_state, int prop_cov) {
    h->kilo->eskf_->predict(dt, prop_state != 0, prop_cov != 0);
    return 0;
}
int lko_update_by_points(lko_handle* h, const double* h6, const double* z, const double* R, size_t N) {
    ObsShared o;
    o.pt_h.assign(h6, h6 + 6 * N);
    o.pt_z.assign(z, z + N);
    o.pt_R.assign(R, R + N);
    h->kilo->eskf_->updateByPoints(o);
    return 0;
}
int lko_update_by_imu(lko_handle* h, const double* z6, const double* R6) {
    ObsShared o;
    o.ki_z.assign(z6, z6 + 6);
    o.ki_R.assign(R6, R6 + 6);
    h->kilo->eskf_->updateByImu(o);
    return 0;
}
int lko_update_by_kin_imu(lko_handle* h, const double* ki_h, const double* ki_z, const double* ki_R, size_t M) {
    ObsShared o;
    o.ki_h.assign(ki_h, ki_h + 30 * M);
    o.ki_z.assign(ki_z, ki_z + M);
    o.ki_R.assign(ki_R, ki_R + M);
    h->kilo->eskf_->updateByKinImu(o);
    return 0;
}

int lko_map_build(lko_handle* h, const float* xyz_world, const float* xyz_body, size_t n) {
    auto& m = *h->kilo->map_manager_;
    m.feats_down_world_.assign(xyz_world, xyz_world + 3 * n);
    m.feats_down_body_.assign(xyz_body, xyz_body + 3 * n);
    auto& e = *h->kilo->eskf_;
    m.BuildVoxelMap(e.getRot(), e.getRotCov(), e.getPosCov());
    return 0;
}
int lko_map_update(lko_handle* h, const double* pw, const double* var9, size_t n) {
    std::vector<pointWithVar> pv(n);
    for (size_t i = 0; i < n; ++i) {
        pv[i].point_w = vec3(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) pv[i].var(r, c) = var9[9 * i + 3 * r + c];
    }
    h->kilo->map_manager_->UpdateVoxelMap(pv);
    return 0;
}
int lko_residuals(lko_handle* h, const float* xyz_body, size_t n, double* h6, double* z, double* R, uint8_t* valid) {
    std::vector<ResidualRow> rows;
    h->kilo->residualsOnly(xyz_body, n, rows);
    for (size_t i = 0; i < n; ++i) {
        valid[i] = rows[i].valid ? 1 : 0;
        for (int c = 0; c < 6; ++c) h6[6 * i + c] = rows[i].h[c];
        z[i] = rows[i].z;
        R[i] = rows[i].R;
    }
    return 0;
}
// TEST DIAGNOSTIC: the smallest relative distance of any gate on the match path of ONE body point from its threshold, with the
// current state: out3 = {range gate (float arithmetic), sigma gate, voxel key: distance of p_w / voxel_size from an integer}.
int lko_residual_margins(lko_handle* h, const float* xyz_body3, double* out3) {
    out3[0] = out3[1] = out3[2] = 1e300;
    auto* m = h->kilo->map_manager_.get();
    m->gate_margin_probe_ = out3;
    std::vector<ResidualRow> rows;
    h->kilo->residualsOnly(xyz_body3, 1, rows);
    m->gate_margin_probe_ = nullptr;
    lk_point pt{xyz_body3[0], xyz_body3[1], xyz_body3[2], 0.f};
    pointWithVar pv;
    PointToPlane pl;
    float w[4];
    h->kilo->matchPoint(pt, pv, pl, w);
    for (int j = 0; j < 3; ++j) {
        const double loc = pv.point_w[j] / m->config_setting_.max_voxel_size_;
        const double d = std::fabs(loc - std::nearbyint(loc));
        if (d < out3[2]) out3[2] = d;
    }
    return rows[0].valid ? 1 : 0;
}
int lko_update_points(lko_handle* h, double t, const float* xyz_body, size_t n, float* xyz_world_out,
                      float* intensity_out, size_t* n_effect) {
    std::vector<lk_point> pts(n);
    for (size_t i = 0; i < n; ++i) pts[i] = lk_point{xyz_body[3 * i], xyz_body[3 * i + 1], xyz_body[3 * i + 2], 0.f};
    std::vector<float> w(4 * n, 0.f);
    size_t ne = n_effect ? *n_effect : 0;
    h->kilo->predictUpdatePoint(t, 0, n, pts.data(), w.data(), ne);
    if (n_effect) *n_effect = ne;
    for (size_t i = 0; i < n; ++i) {
        if (xyz_world_out)
            for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
        if (intensity_out) intensity_out[i] = w[4 * i + 3];
    }
    return 0;
}
int lko_update_imu(lko_handle* h, const lk_imu* imu) {
    h->kilo->predictUpdateImu(*imu);
    return 0;
}
int lko_update_kin_imu(lko_handle* h, const lk_kin_imu* kin) {
    h->kilo->predictUpdateKinImu(*kin);
    return 0;
}
int lko_first_frame(lko_handle* h, const lk_point* raw, size_t n, double end_time, const lk_imu* imus, size_t n_imu,
                    const lk_kin_imu* kins, size_t n_kin) {
    h->kilo->firstFrame(raw, n, end_time, imus, n_imu, kins, n_kin);
    return 0;
}
// with_sort != 0 additionally runs the reference's std::sort (KILO.cc:369-370) on a private copy
// first so that the timed region matches the reference's Timer lambda; results are unaffected
// because the copy is discarded and the caller's (already sorted) order is the one processed.
int lko_process_scan(lko_handle* h, const lk_point* sorted_pts, size_t n, double t_begin, const lk_imu* imus,
                     size_t n_imu, const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out,
                     int with_sort) {
    if (with_sort) {
        std::vector<lk_point> tmp(sorted_pts, sorted_pts + n);
        std::sort(tmp.begin(), tmp.end(), [](const lk_point& a, const lk_point& b) { return a.curvature < b.curvature; });
        volatile float sink = tmp.empty() ? 0.f : tmp[n / 2].x;
        (void)sink;
    }
    std::deque<lk_imu> qi(imus, imus + n_imu);
    std::deque<lk_kin_imu> qk(kins, kins + n_kin);
    std::vector<float> w(4 * n, 0.f);
    size_t ne = 0;
    uint32_t nb = 0, nu = 0;
    h->kilo->processSorted(sorted_pts, n, t_begin, qi, qk, w.data(), ne, &nb, &nu);
    if (xyz_world_out)
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
    if (out) {
        const State& s = h->kilo->eskf_->state();
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) out->rot[3 * i + j] = s.rot_(i, j);
            out->pos[i] = s.pos_[i];
            out->vel[i] = s.vel_[i];
        }
        out->n_effect = ne;
        out->n_buckets = nb;
        out->n_updates = nu;
    }
    return 0;
}

// ---- blob export in the format of include/legkilo_hip.h: export_blob.hpp (shared with the reference-side wrapper)
int lko_map_export(lko_handle* h, void* blob, size_t* bytes) {
    auto& m = *h->kilo->map_manager_;
    return lkx::export_map<VoxelOctoTree, VoxelPlane, pointWithVar>(m.voxel_map_, m.config_setting_.max_voxel_size_,
                                                                   m.config_setting_.max_layer_,
                                                                   m.config_setting_.max_points_num_, blob, bytes);
}

// ---- blob import (the inverse of export_blob.hpp): rebuild the octrees from a blob of include/legkilo_hip.h, so that a
// checker can replay scans against EXACTLY the map a device handle holds (bench.py's in-line parity check of the timed loop,
// tests).  What the blob does not carry is not restored: points of first-frame leaves with more than LK_BLOCK_PTS points
// (LK_NODE_PTS_DROPPED - such leaves are frozen, voxel_map.cc:199) and points of inner non-plane nodes (never read again).
static VoxelOctoTree* import_node(const lk_config& cfg, const lk_node_rec* nodes, const lk_plane_rec* planes,
                                  const lk_block_rec* blocks, uint32_t n_nodes, uint32_t n_blocks, int id, int depth) {
    if (id < 0 || (uint32_t)id >= n_nodes || depth > 8) return nullptr;
    const lk_node_rec& n = nodes[id];
    const lk_plane_rec& p = planes[id];
    const int layer = n.layer;
    const int lin = (layer >= 0 && layer < 5) ? cfg.layer_init_num[layer] : cfg.layer_init_num[4];
    VoxelOctoTree* t = new VoxelOctoTree(cfg.max_layer, layer, lin, cfg.max_points_num, (float)cfg.planner_threshold);
    for (int c = 0; c < 3; ++c) t->voxel_center_[c] = n.voxel_center[c];
    t->quater_length_ = n.quater_length;
    t->new_points_ = n.new_points;
    t->init_octo_ = (n.state & LK_NODE_INIT_OCTO) != 0;
    t->update_enable_ = (n.state & LK_NODE_UPDATE_ENABLE) != 0;
    t->octo_state_ = (n.state & LK_NODE_OCTO_STATE) ? 1 : 0;
    t->layer_init_num_.assign(cfg.layer_init_num, cfg.layer_init_num + 5);
    if (n.block >= 0 && (uint32_t)n.block < n_blocks && n.npts > 0 && n.npts <= LK_BLOCK_PTS) {
        const lk_block_rec& b = blocks[n.block];
        t->temp_points_.resize(n.npts);
        for (int i = 0; i < n.npts; ++i) {
            pointWithVar& pv = t->temp_points_[i];
            for (int c = 0; c < 3; ++c) pv.point_w[c] = b.pts[i].pw[c];
            const double* v = b.pts[i].var;
            pv.var(0, 0) = v[0], pv.var(0, 1) = v[1], pv.var(0, 2) = v[2], pv.var(1, 0) = v[1], pv.var(1, 1) = v[3];
            pv.var(1, 2) = v[4], pv.var(2, 0) = v[2], pv.var(2, 1) = v[4], pv.var(2, 2) = v[5];
        }
    }
    VoxelPlane& pl = *t->plane_ptr_;
    for (int c = 0; c < 3; ++c) pl.center_[c] = p.center[c], pl.normal_[c] = p.normal[c];
    pl.d_ = p.d;
    pl.radius_ = p.radius;
    pl.is_plane_ = (p.flags & LK_PLANE_IS_PLANE) != 0;
    pl.is_init_ = (p.flags & LK_PLANE_IS_INIT) != 0;
    pl.points_size_ = p.points_size;
    int k = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c) pl.plane_var_(r, c) = pl.plane_var_(c, r) = p.plane_var[k++];
    pl.min_eigen_value_ = p.min_eigen_value;
    pl.mid_eigen_value_ = p.mid_eigen_value;
    pl.max_eigen_value_ = p.max_eigen_value;
    for (int l = 0; l < 8; ++l)
        if (n.child[l] >= 0) t->leaves_[l] = import_node(cfg, nodes, planes, blocks, n_nodes, n_blocks, n.child[l], depth + 1);
    return t;
}
int lko_map_import(lko_handle* h, const void* blob, size_t bytes) {
    if (!blob || bytes < sizeof(lk_blob_header)) return -1;
    lk_blob_header hd;
    std::memcpy(&hd, blob, sizeof(hd));
    const size_t want = sizeof(hd) + (size_t)hd.n_roots * sizeof(lk_root_rec) + (size_t)hd.n_nodes * (sizeof(lk_node_rec) + sizeof(lk_plane_rec)) +
                        (size_t)hd.n_blocks * sizeof(lk_block_rec);
    if (hd.magic != LK_BLOB_MAGIC || hd.version != LK_ABI_VERSION || hd.block_pts != LK_BLOCK_PTS || hd.bytes != want || want > bytes) return -1;
    const char* p = (const char*)blob + sizeof(hd);
    const lk_root_rec* roots = (const lk_root_rec*)p;
    p += (size_t)hd.n_roots * sizeof(lk_root_rec);
    const lk_node_rec* nodes = (const lk_node_rec*)p;
    p += (size_t)hd.n_nodes * sizeof(lk_node_rec);
    const lk_plane_rec* planes = (const lk_plane_rec*)p;
    p += (size_t)hd.n_nodes * sizeof(lk_plane_rec);
    const lk_block_rec* blocks = (const lk_block_rec*)p;
    // every id is checked BEFORE the oracle's own map is touched: a mismatched blob must fail, never leave a half-built map that a
    // later "parity" comparison would silently run against
    for (uint32_t r = 0; r < hd.n_roots; ++r)
        if (roots[r].node < 0 || (uint32_t)roots[r].node >= hd.n_nodes) return -1;
    for (uint32_t i = 0; i < hd.n_nodes; ++i) {
        for (int c = 0; c < 8; ++c)
            if (nodes[i].child[c] < -1 || (nodes[i].child[c] >= 0 && (uint32_t)nodes[i].child[c] >= hd.n_nodes)) return -1;
        if (nodes[i].block < -1 || (nodes[i].block >= 0 && (uint32_t)nodes[i].block >= hd.n_blocks)) return -1;
    }
    std::vector<std::pair<Vec3i, VoxelOctoTree*>> built;
    for (uint32_t r = 0; r < hd.n_roots; ++r) {
        VoxelOctoTree* t = import_node(h->cfg, nodes, planes, blocks, hd.n_nodes, hd.n_blocks, roots[r].node, 0);
        if (!t) {
            for (auto& kv : built) delete kv.second;
            return -1;
        }
        built.emplace_back(Vec3i{roots[r].key[0], roots[r].key[1], roots[r].key[2]}, t);
    }
    auto& m = *h->kilo->map_manager_;
    for (auto& kv : m.voxel_map_) delete kv.second;
    m.voxel_map_.clear();
    for (auto& kv : built) m.voxel_map_[kv.first] = kv.second;
    return 0;
}

// VoxelMapManager::mapSliding with the caller-set position_last_ and the two parameters of voxel_map.h:54,56
int lko_map_slide(lko_handle* h, const double* position3, double sliding_thresh, int half_map_size, int* slid, uint32_t* n_removed) {
    auto* m = h->kilo->map_manager_.get();
    m->config_setting_.sliding_thresh = sliding_thresh;
    m->config_setting_.half_map_size = half_map_size;
    m->position_last_ = vec3(position3[0], position3[1], position3[2]);
    const size_t before = m->voxel_map_.size();
    const bool s = m->mapSliding();
    if (slid) *slid = s ? 1 : 0;
    if (n_removed) *n_removed = (uint32_t)(before - m->voxel_map_.size());
    return 0;
}
int lko_map_clear_outside(lko_handle* h, int x_max, int x_min, int y_max, int y_min, int z_max, int z_min, uint32_t* n_removed) {
    const int n = h->kilo->map_manager_->clearMemOutOfMap(x_max, x_min, y_max, y_min, z_max, z_min);
    if (n_removed) *n_removed = (uint32_t)n;
    return 0;
}
int lko_map_slide_position(lko_handle* h, int set, double* last3) {
    auto* m = h->kilo->map_manager_.get();
    for (int i = 0; i < 3; ++i) {
        if (set) m->last_slide_position[i] = last3[i];
        else last3[i] = m->last_slide_position[i];
    }
    return 0;
}
// build_single_residual (voxel_map.cc:363-427) on the root voxel `key3`, started like KILO.cc:150-155
int lko_match_voxel(lko_handle* h, const int* key3, const double* pw3, const double* var9, int* found, int* success,
                    double* prob, double* normal3, double* center3, double* d, float* dis_to_plane, int* layer) {
    auto& m = *h->kilo->map_manager_;
    Vec3i key{key3[0], key3[1], key3[2]};
    auto it = m.voxel_map_.find(key);
    *found = it != m.voxel_map_.end();
    *success = 0, *prob = 0;
    if (!*found) return 0;
    pointWithVar pv;
    pv.point_w = vec3(pw3[0], pw3[1], pw3[2]);
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
int lko_state_minus(const double* xa36, const double* xb36, double* delta30) {
    State a, b;
    x36_to_state(xa36, a);
    x36_to_state(xb36, b);
    StateVec d = a - b;
    for (int i = 0; i < 30; ++i) delta30[i] = d[i];
    return 0;
}
int lko_key_floor(const double* p3, double voxel_size, int* key3) {
    Vec3i k = voxelKeyFloor(vec3(p3[0], p3[1], p3[2]), voxel_size);
    for (int i = 0; i < 3; ++i) key3[i] = k[i];
    return 0;
}
int lko_map_stats(lko_handle* h, uint32_t* n_roots) {
    *n_roots = (uint32_t)h->kilo->map_manager_->voxel_map_.size();
    return 0;
}

// ---- unit-level hooks for tests ----
int lko_calc_body_cov(const double* pb3, float range_inc, float degree_inc, double* cov9) {
    Vec3 pb = vec3(pb3[0], pb3[1], pb3[2]);
    Mat3 cov;
    calcBodyCov(pb, range_inc, degree_inc, cov);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov9[3 * i + j] = cov(i, j);
    return 0;
}
int lko_eig_sym3(const double* A9, double* evals3, double* evecs9) {
    Mat3 A, V;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A(i, j) = A9[3 * i + j];
    eig_sym3(A, evals3, V);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) evecs9[3 * i + j] = V(i, j);
    return 0;
}
int lko_init_plane(const double* pw, const double* var9, size_t n, float planer_threshold, lk_plane_rec* out,
                   double* plane_var36) {
    VoxelOctoTree t(2, 0, 5, 50, planer_threshold);
    std::vector<pointWithVar> pts(n);
    for (size_t i = 0; i < n; ++i) {
        pts[i].point_w = vec3(pw[3 * i], pw[3 * i + 1], pw[3 * i + 2]);
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
int lko_exp_log(const double* v3, double* R9_exp3, double* R9_expvec, double* log3) {
    Mat3 a = Exp3(v3[0], v3[1], v3[2]);
    Mat3 b = ExpVec(vec3(v3[0], v3[1], v3[2]));
    Vec3 l = LogSO3(a);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R9_exp3[3 * i + j] = a(i, j), R9_expvec[3 * i + j] = b(i, j);
        log3[i] = l[i];
    }
    return 0;
}
size_t lko_hash_vec3(int x, int y, int z) { return hash_vec3()(Vec3i{x, y, z}); }
int lko_abi_sizes(size_t* out8) {
    out8[0] = sizeof(lk_config), out8[1] = sizeof(lk_point), out8[2] = sizeof(lk_imu), out8[3] = sizeof(lk_kin_imu);
    out8[4] = sizeof(lk_pose), out8[5] = sizeof(lk_plane_rec), out8[6] = sizeof(lk_node_rec), out8[7] = sizeof(lk_block_rec);
    return 0;
}

}  // extern "C"
