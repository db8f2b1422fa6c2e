// legkilo_hip.hip - the main translation unit of liblegkilo_hip.so (see lk_internal.h)
#define LK_TU_MAIN 1
#include "lk_internal.h"

thread_local std::string g_err;

static int create_pools(lk_handle* h, const lk_config* cfg);

extern "C" {

int lk_abi_version(void) { return LK_ABI_VERSION; }

const char* lk_last_error(const lk_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

int lk_create(const lk_config* cfg, lk_handle** out) {
    if (!cfg || !out) return fail(nullptr, LK_ERR_INVALID, "lk_create: null argument");
    *out = nullptr;
    if (cfg->max_layer < 0 || cfg->max_layer > LK_MAX_LAYER)
        return fail(nullptr, LK_ERR_INVALID, "max_layer must be in [0,4]");
    if (cfg->max_points_num + 2 > LK_BLOCK_PTS) return fail(nullptr, LK_ERR_INVALID, "max_points_num must be <= 50");
    for (int i = 0; i < 5; ++i)
        if (cfg->layer_init_num[i] + 1 > LK_BLOCK_PTS || cfg->layer_init_num[i] < 1)
            return fail(nullptr, LK_ERR_INVALID, "layer_init_num out of range");
    if (cfg->n_slots < 1 || cfg->max_roots < 16 || cfg->max_nodes < cfg->max_roots || cfg->max_point_blocks < 16 ||
        cfg->max_scan_points < 64)
        return fail(nullptr, LK_ERR_INVALID, "capacities too small");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, LK_ERR_NO_DEVICE, "no HIP device visible (liblegkilo_hip has no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, LK_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess)
        return fail(nullptr, LK_ERR_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, LK_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    lk_handle* h = new lk_handle;
    h->cfg = *cfg;
    const int rc = create_pools(h, cfg);
    if (rc != LK_OK) {  // release whatever was allocated before the failure (lk_destroy is null-safe per field)
        const std::string why = h->err;
        lk_destroy(h);
        return fail(nullptr, rc, why);
    }
    *out = h;
    return LK_OK;
}

static int create_pools(lk_handle* h, const lk_config* cfg) {
    HIPCHK(h, hipSetDevice(cfg->device_id));
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    if (const char* e = getenv("LEGKILO_UPDATE_CLASSIC")) h->wave_update = atoi(e) == 0;
    if (const char* e = getenv("LEGKILO_GRID")) h->grid_enable = atoi(e) != 0;
    if (const char* e = getenv("LEGKILO_BATCH_ORDER")) h->batch_order_mode = atoi(e) != 0;
    HIPCHK(h, hipMalloc(&h->d_grid_mm, 8 * sizeof(int)));
    if (const char* e = getenv("LEGKILO_REPLAY_GROUPS")) h->replay_groups = std::min(std::max(atoi(e), 1), (int)lk_handle::kMaxGroups);
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    // parameters
    LkParams& pr = h->pr;
    memcpy(pr.ext_R, cfg->ext_R, sizeof(pr.ext_R));
    memcpy(pr.ext_T, cfg->ext_T, sizeof(pr.ext_T));
    {
        static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        pr.ext_identity = memcmp(cfg->ext_R, I3, sizeof(I3)) == 0 ? 1 : 0;   // +0.0 only: -0.0 entries take the general path
    }
    pr.voxel_size_d = cfg->max_voxel_size;
    pr.voxel_size_f = (float)cfg->max_voxel_size;
    pr.sigma_num = cfg->sigma_num;
    pr.lidar_ratio = cfg->lidar_point_meas_ratio;
    {
        float degree_inc = (float)cfg->beam_err, range_inc = (float)cfg->dept_err;  // float parameters of calcBodyCov
        double sd = std::sin((degree_inc) * 0.017453293);                           // DEG2RAD (pcl_macros.h)
        pr.dir_var = sd * sd;
        pr.range_var = range_inc * range_inc;
    }
    {
        double inv = 1.0 / cfg->max_voxel_size;
        int e = 0;
        bool pow2 = std::frexp(cfg->max_voxel_size, &e) == 0.5;  // mantissa exactly 0.5 <=> power of two
        pr.inv_vs_exact = pow2 ? inv : 0.0;
    }
    pr.planer_threshold = (float)cfg->planner_threshold;
    pr.max_layer = cfg->max_layer;
    pr.max_points_num = cfg->max_points_num;
    for (int i = 0; i < 5; ++i) pr.layer_init_num[i] = cfg->layer_init_num[i];
    // pools
    // load factor <= 1/8: probe chains are what a wave waits for (505 -> 480 us per 20.5 M points against 1/3); 16 B per slot
    h->hash_cap = next_pow2(8u * cfg->max_roots);
    LkMap& m = h->map;
    memset(&m, 0, sizeof(m));
    m.hash_mask = h->hash_cap - 1;
    m.max_nodes = cfg->max_nodes;
    m.max_blocks = cfg->max_point_blocks;
    m.max_scan = cfg->max_scan_points;
    HIPCHK(h, hipMalloc(&m.hash, sizeof(int4) * (size_t)h->hash_cap));
    HIPCHK(h, hipMalloc(&m.planes, sizeof(lk_plane_rec) * (size_t)m.max_nodes));
    HIPCHK(h, hipMalloc(&m.match, sizeof(lk_match_rec) * (size_t)m.max_nodes));
    HIPCHK(h, hipMalloc(&m.nodes, sizeof(lk_node_rec) * (size_t)m.max_nodes));
    HIPCHK(h, hipMalloc(&m.blocks, sizeof(lk_block_rec) * (size_t)m.max_blocks));
    HIPCHK(h, hipMalloc(&m.counters, sizeof(unsigned int) * LK_CTR_COUNT));
    HIPCHK(h, hipMalloc(&m.touched, sizeof(int) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&m.heavy, sizeof(int) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&m.next, sizeof(int) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&m.slots, sizeof(int) * (size_t)LK_SLOTS * (size_t)m.max_nodes));
    HIPCHK(h, hipMalloc(&m.scratch, sizeof(int) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&m.groups, sizeof(LkGroup) * 2 * (size_t)m.max_scan));   // leaf-group descriptors, then the fallback items
    HIPCHK(h, hipMalloc(&m.gidx, sizeof(int) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&m.free_list, sizeof(int) * (size_t)m.max_blocks));
    HIPCHK(h, hipMalloc(&m.freed_next, sizeof(int) * (size_t)m.max_blocks));
    HIPCHK(h, hipMalloc(&m.dirty, sizeof(unsigned int) * (size_t)m.max_nodes));
    HIPCHK(h, hipMemsetAsync(m.dirty, 0, sizeof(unsigned int) * (size_t)m.max_nodes, h->stream));
    HIPCHK(h, hipMalloc(&m.newroot, sizeof(unsigned int) * (size_t)(LK_NEWROOT_MASK + 1)));
    HIPCHK(h, hipMemsetAsync(m.newroot, 0, sizeof(unsigned int) * (size_t)(LK_NEWROOT_MASK + 1), h->stream));
    HIPCHK(h, hipMalloc(&m.spec, sizeof(unsigned int) * LK_SPEC_WORDS));
    HIPCHK(h, hipMemsetAsync(m.spec, 0, sizeof(unsigned int) * LK_SPEC_WORDS, h->stream));
    m.epoch = 0;
    HIPCHK(h, hipStreamCreateWithFlags(&h->ins, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_U[i], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_D[i], hipEventDisableTiming));
    }
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_I, hipEventDisableTiming));
    if (const char* e = getenv("LEGKILO_SPEC")) h->spec_enable = atoi(e) != 0;
    if (const char* e = getenv("LEGKILO_RESIDENT")) h->resident_enable = atoi(e) != 0;
    if (const char* e = getenv("LEGKILO_GRIDSCAN")) h->gridscan_mode = std::min(std::max(atoi(e), 0), 2);
    HIPCHK(h, hipMalloc(&h->d_snap, sizeof(LkFilter) * 2));
    HIPCHK(h, hipMemsetAsync(h->d_snap, 0, sizeof(LkFilter) * 2, h->stream));
    HIPCHK(h, hipMalloc(&h->d_ids, sizeof(int2) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&h->d_filters, sizeof(LkFilter) * (size_t)cfg->n_slots));
    HIPCHK(h, hipMemsetAsync(h->d_filters, 0, sizeof(LkFilter) * (size_t)cfg->n_slots, h->stream));
    HIPCHK(h, hipMalloc(&h->d_Q, sizeof(double) * 900));
    HIPCHK(h, hipMemsetAsync(h->d_Q, 0, sizeof(double) * 900, h->stream));
    size_t nblk_max = ((size_t)m.max_scan + LK_RB - 1) / LK_RB;
    h->part_stride = nblk_max * (LK_RB / LK_WAVE) * LK_NPART;  // one partial record per wave
    HIPCHK(h, hipMalloc(&h->d_partials, sizeof(double) * h->part_stride * cfg->n_slots));
    HIPCHK(h, hipMalloc(&h->d_scan, sizeof(lk_point) * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&h->d_world, sizeof(float) * 4 * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&h->d_rows, sizeof(double) * 8 * (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&h->d_valid, (size_t)m.max_scan));
    HIPCHK(h, hipMalloc(&h->d_tmp, sizeof(double) * 4096));
    HIPCHK(h, hipMalloc(&h->d_poses, sizeof(lk_pose) * (size_t)cfg->n_slots));
    unsigned int ninit = std::max(h->hash_cap, m.max_nodes);
    hipLaunchKernelGGL(lk_pool_init_kernel, dim3((ninit + 255) / 256), dim3(256), 0, h->stream, m, h->hash_cap);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

void lk_destroy(lk_handle* h) {
    if (!h) return;
    hipSetDevice(h->cfg.device_id);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->ins) hipStreamSynchronize(h->ins);
    for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)   // an asynchronous batch may still be running on a side stream
        if (h->side[i]) hipStreamSynchronize(h->side[i]);
    void* ptrs[] = {h->map.hash, h->map.planes, h->map.match, h->map.nodes, h->map.blocks, h->map.counters, h->map.touched, h->map.heavy,
                    h->map.next, h->map.slots, h->map.scratch, h->map.groups, h->map.gidx, h->map.free_list, h->map.freed_next, h->d_filters, h->d_Q, h->d_partials, h->d_scan, h->d_world,
                    h->d_rows, h->d_valid, h->d_tmp, h->d_poses, h->d_rag, h->d_grid_mm, h->d_ragdev, h->d_ragtmp,
                    h->map.dirty, h->map.newroot, h->map.spec, h->d_snap, h->d_ids, h->d_fbackup, h->d_ov_priors};
    for (void* p : ptrs)
        if (p) hipFree(p);
    void* pre[] = {h->pre_raw, h->pre_cells, h->pre_out, h->pre_k0, h->pre_k1, h->pre_flags, h->pre_pos, h->pre_misc,
                   h->pre_v0, h->pre_v1, h->pre_starts, h->pre_tmp};
    for (void* p : pre)
        if (p) hipFree(p);
    if (h->h_rag) hipHostFree(h->h_rag);
    if (h->h_result) hipHostFree(h->h_result);
    ov_free(h);
    for (auto& e : h->ord) {
        if (e.copy) hipFree(e.copy);
        if (e.d_ref) hipFree(e.d_ref);
        if (e.h_seen) hipHostFree(e.h_seen);
    }
    if (h->d_ov_status) hipFree(h->d_ov_status);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    for (int i = 0; i < 2; ++i) {
        if (h->ev_U[i]) hipEventDestroy(h->ev_U[i]);
        if (h->ev_D[i]) hipEventDestroy(h->ev_D[i]);
    }
    if (h->ev_I) hipEventDestroy(h->ev_I);
    if (h->ins) hipStreamDestroy(h->ins);
    for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i) {
        if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]);
        if (h->side[i]) hipStreamDestroy(h->side[i]);
    }
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}


// ------------------------------------------------------------------ ESKF surface
int lk_set_state(lk_handle* h, uint32_t slot, const double* x36, const double* P900) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    LkFilter* f = h->d_filters + slot;
    if (x36) HIPCHK(h, hipMemcpyAsync(f->x, x36, sizeof(double) * 36, hipMemcpyHostToDevice, h->stream));
    if (P900) HIPCHK(h, hipMemcpyAsync(f->P, P900, sizeof(double) * 900, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_get_state(lk_handle* h, uint32_t slot, double* x36, double* P900) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    LkFilter* f = h->d_filters + slot;
    if (x36) HIPCHK(h, hipMemcpyAsync(x36, f->x, sizeof(double) * 36, hipMemcpyDeviceToHost, h->stream));
    if (P900) HIPCHK(h, hipMemcpyAsync(P900, f->P, sizeof(double) * 900, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_set_Q(lk_handle* h, const double* Q900) {
    CHECK_H(h);
    if (!Q900) return fail(h, LK_ERR_INVALID, "Q is null");
    h->q_diag = true;
    for (int i = 0; i < 30 && h->q_diag; ++i)
        for (int j = 0; j < 30; ++j)
            if (i != j && Q900[i * 30 + j] != 0.0) {
                h->q_diag = false;
                break;
            }
    HIPCHK(h, hipMemcpyAsync(h->d_Q, Q900, sizeof(double) * 900, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_get_Q(lk_handle* h, double* Q900) {
    CHECK_H(h);
    HIPCHK(h, hipMemcpyAsync(Q900, h->d_Q, sizeof(double) * 900, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
// initProcessCovQ, eskf.cc:47-62 (diagonal blocks only; built on the host, it is 7 scalars)
int lk_init_process_cov_q(lk_handle* h) {
    CHECK_H(h);
    std::vector<double> Q(900, 0.0);
    auto diag3 = [&](int o, double v) {
        for (int k = 0; k < 3; ++k) Q[(o + k) * 30 + (o + k)] = v;
    };
    diag3(6, h->cfg.vel_process_cov);
    diag3(9, h->cfg.acc_bias_process_cov);
    diag3(12, h->cfg.gyr_bias_process_cov);
    diag3(18, h->cfg.imu_acc_process_cov);
    diag3(21, h->cfg.imu_gyr_process_cov);
    diag3(24, h->cfg.kin_bias_process_cov);
    diag3(27, h->cfg.contact_process_cov);
    return lk_set_Q(h, Q.data());
}
int lk_set_times(lk_handle* h, uint32_t slot, double last_predict_t, double last_update_t) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    double t[2] = {last_predict_t, last_update_t};
    HIPCHK(h, hipMemcpyAsync(&h->d_filters[slot].last_predict_t, t, sizeof(t), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_get_times(lk_handle* h, uint32_t slot, double* last_predict_t, double* last_update_t) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    double t[2];
    HIPCHK(h, hipMemcpyAsync(t, &h->d_filters[slot].last_predict_t, sizeof(t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *last_predict_t = t[0];
    *last_update_t = t[1];
    return LK_OK;
}
int lk_set_acc_norm(lk_handle* h, double acc_norm) {
    CHECK_H(h);
    h->acc_norm = acc_norm;
    return LK_OK;
}
int lk_get_acc_norm(lk_handle* h, double* acc_norm) {
    CHECK_H(h);
    if (!acc_norm) return fail(h, LK_ERR_INVALID, "acc_norm is null");
    *acc_norm = h->acc_norm;
    return LK_OK;
}
int lk_get_fx(lk_handle* h, uint32_t slot, double dt, double* Fx900) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    LAUNCH(h, "fx", hipLaunchKernelGGL(lk_fx_kernel, dim3(1), dim3(64), 0, h->stream, h->d_filters, (int)slot, dt, h->d_tmp,
                                       h->d_tmp + 900));
    HIPCHK(h, hipMemcpyAsync(Fx900, h->d_tmp, sizeof(double) * 900, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_get_function_f(lk_handle* h, uint32_t slot, double dt, double* f30) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    LAUNCH(h, "fx", hipLaunchKernelGGL(lk_fx_kernel, dim3(1), dim3(64), 0, h->stream, h->d_filters, (int)slot, dt, h->d_tmp,
                                       h->d_tmp + 900));
    HIPCHK(h, hipMemcpyAsync(f30, h->d_tmp + 900, sizeof(double) * 30, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_predict(lk_handle* h, uint32_t slot, double dt, int prop_state, int prop_cov) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    LAUNCH(h, "predict_dt", hipLaunchKernelGGL(lk_predict_dt_kernel, dim3(1), dim3(LK_FB), 0, h->stream,
                                               h->d_filters + slot, h->d_Q, dt, prop_state, prop_cov));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_update_by_points(lk_handle* h, uint32_t slot, const double* h6, const double* z, const double* R, size_t N) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    if (N == 0) return LK_OK;
    if (N > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "N exceeds max_scan_points");
    double* d = h->d_rows;
    HIPCHK(h, hipMemcpyAsync(d, h6, sizeof(double) * 6 * N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + 6 * N, z, sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d + 7 * N, R, sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "obs_points", hipLaunchKernelGGL(lk_obs_points_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters,
                                               (int)slot, d, d + 6 * N, d + 7 * N, (int)N));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_update_by_imu(lk_handle* h, uint32_t slot, const double* ki_z6, const double* ki_R6) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    HIPCHK(h, hipMemcpyAsync(h->d_tmp, ki_z6, sizeof(double) * 6, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tmp + 6, ki_R6, sizeof(double) * 6, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "obs_imu", hipLaunchKernelGGL(lk_obs_imu_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters,
                                            (int)slot, h->d_tmp, h->d_tmp + 6));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_update_by_kin_imu(lk_handle* h, uint32_t slot, const double* ki_h, const double* ki_z, const double* ki_R, size_t M) {
    CHECK_H(h);
    CHECK_SLOT(h, slot);
    if (M < 1 || M > 18) return fail(h, LK_ERR_INVALID, "M must be in [1,18]");
    HIPCHK(h, hipMemcpyAsync(h->d_tmp, ki_h, sizeof(double) * 30 * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tmp + 540, ki_z, sizeof(double) * M, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tmp + 560, ki_R, sizeof(double) * M, hipMemcpyHostToDevice, h->stream));
    LAUNCH(h, "obs_kin", hipLaunchKernelGGL(lk_obs_kin_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters, (int)slot,
                                            h->d_tmp, h->d_tmp + 540, h->d_tmp + 560, (int)M));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

// lk_update_kernel of the stream path (slot 0), followed in the same single-workgroup launch by the posterior's snapshot for the
// insert (dev_snapshot_posterior), the insert's pool bookkeeping (do_predict >= 0; the pipelined path does it on its insert stream)
// and - do_predict == 1 - the predict to the next bucket's time.
__global__ void __launch_bounds__(LK_FB)
    lk_update_snap_kernel(LkMap map, LkFilter* filters, const double* __restrict__ partials, int nblk, double t, const double* __restrict__ Q,
                          double t_next, int do_predict, LkFilter* snap) {
    __shared__ FilterSmem sm;
    __shared__ double red[8][LK_NPART];
    __shared__ double tot[LK_NPART];
    // A launch of TWO workgroups when the pool bookkeeping belongs to it: workgroup 1 does that (three dependent trips to the counters and the
    // free lists - it has nothing to do with the filter), workgroup 0 the update.  Round 5: the state is requested BEFORE the partial records
    // (it does not depend on them: one trip instead of two), and the snapshot is written from LDS together with the posterior instead of
    // being read back from what was just stored - the launch is one link of a bucket's chain of five, 11 us of ~70.
    if (blockIdx.x == 1) {
        if (do_predict >= 0) dev_bucket_begin(map);
        return;
    }
    const int tid = threadIdx.x;
    LkFilter* f = &filters[0];
    static_assert(sizeof(WaveSmem) <= sizeof(double) * 1800, "WaveSmem must fit FilterSmem::A + B");
    WaveSmem& w = *reinterpret_cast<WaveSmem*>(&sm.A[0]);   // the staging area of dev_point_update_wave0
    double pr_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pr_[k] = tid + LK_FB * k < 900 ? f->P[tid + LK_FB * k] : 0.0;
    const double xr_ = tid < 36 ? f->x[tid] : 0.0;
    dev_reduce_partials(partials, nblk, red, tot);   // the sum of dev_update_reduce
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tid + LK_FB * k < 900) w.P[tid + LK_FB * k] = pr_[k];
    if (tid < 36) w.x[tid] = xr_;
    const int N = (int)(tot[28] + 0.5);
    if (tid == 0) {   // the bookkeeping of dev_update_from_totals (KILO.cc:193,211-212)
        f->n_buckets += 1;
        f->last_N = N;
        f->updated = N > 0;
        if (N > 0) {
            f->n_updates += 1;
            f->n_effect += (unsigned long long)N;
            f->last_update_t = t;  // KILO.cc:212
        }
    }
    __syncthreads();
    if (N > 0 && tid < LK_WAVE) wave_update_core<true>(w, tid < 32 ? tot[tid] : 0.0, N, tid);   // dev_point_update_wave0's core
    __syncthreads();
    if (N > 0) {
        for (int i = tid; i < 900; i += LK_FB) f->P[i] = w.P[i];
        if (tid < 36) f->x[tid] = w.x[tid];
    }
    if (tid < LK_STATE_DOUBLES) snap->x[tid] = w.x[tid];   // dev_snapshot_posterior's fields
    if (tid < 180) snap->P[tid] = w.P[tid];
    if (tid == 0) snap->updated = N > 0, snap->last_N = N;
    if (do_predict == 1) {
        __syncthreads();  // f->x, f->P, f->last_update_t written by the update are re-read by dev_predict
        dev_predict(f, Q, t_next, sm);
    }
}

// lk_bucket_begin_kernel + lk_predict_kernel in one launch (on a single dependent stream every kernel boundary costs
// ~8-10 us; the two pieces touch disjoint data)
static_assert(LK_FB == 256, "dev_bucket_begin strides by 256 threads");
__global__ void __launch_bounds__(LK_FB) lk_begin_predict_kernel(LkMap map, LkFilter* filters, const double* __restrict__ Q, double t) {
    __shared__ FilterSmem sm;
    dev_bucket_begin(map);
    dev_predict(&filters[0], Q, t, sm);
}

// Small buckets (the reference's 2 ms time bins hold tens to hundreds of points on a real scan) are pure per-bucket
// latency: for n <= LK_SMALL_MAX the bookkeeping, the predict, the residual pass and the update run as ONE
// single-workgroup kernel - block barriers instead of three dependent launches.  The tiles of the bucket are spread
// over the workgroup's four waves (same residual_tile code as lk_residual_kernel); wave partials are combined in a
// fixed order.
#define LK_SMALL_MAX 512
}  // extern "C" (the kernel below is a template)
template <bool XID>
__global__ void __launch_bounds__(LK_FB)
    lk_small_bucket_kernel(LkMap map, LkParams pr, LkFilter* filters, const double* __restrict__ Q, double t,
                           const lk_point* __restrict__ pts, int n, float* world, int reproject) {
    __shared__ FilterSmem sm;
    __shared__ double rows[LK_FB / LK_WAVE][64 * LK_ROW2];
    __shared__ double red[LK_FB / LK_WAVE][LK_NPART];
    __shared__ double tot[LK_NPART];
    LkFilter* f = &filters[0];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    dev_bucket_begin(map);
    dev_predict(f, Q, t, sm);  // ends with a workgroup barrier: the propagated state is visible to every thread
    BucketConst bc;
    load_bucket_const<false>(f, pr, bc);
    ResidualOut ro;
    ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = world;
    double acc = 0.0;
    for (int base = wv * LK_WAVE; base < n; base += LK_FB) {
        __builtin_amdgcn_wave_barrier();  // the previous tile's reads of this wave's rows are complete
        acc += residual_tile<false, 0, XID, true>(map, pr, bc, reinterpret_cast<const float4*>(pts), base + lane, n, &rows[wv][0], lane, ro, (size_t)0);
    }
    if (lane < LK_NPART) red[wv][lane] = (lane < 29) ? acc : 0.0;
    __syncthreads();
    if (tid < LK_NPART) {
        double s = 0.0;
        for (int w = 0; w < LK_FB / LK_WAVE; ++w) s += red[w][tid];
        tot[tid] = s;
    }
    __syncthreads();
    dev_update_from_totals(f, sm, tot, t);
    // reproject != 0 (tiny buckets): the re-projection with the posterior (KILO.cc:216-230) and the root hashing of the insert
    // (reproject == 2) follow in the same workgroup - a device-scope fence + barrier instead of a launch boundary (~4 us on a
    // dependent stream); the same dev_reproject_point per point as lk_reproject_kernel
    if (reproject) {
        __threadfence();
        __syncthreads();
        for (int i = tid; i < n; i += LK_FB) dev_reproject_point(map, pr, filters, pts, world, reproject == 2 ? 1 : 0, i);
    }
}
// Tiny buckets (n <= 64: ONE tile): the whole filter side of the bucket as ONE WAVE - pool bookkeeping, predict, the tile, update,
// re-projection - with the one-wave filter cores of the batch-replay kernels (wave_predict_core / wave_update_core: P and x stay
// in 7.7 KB of LDS, the 6 x 37 system one column per lane, broadcasts through v_readlane) instead of the 256-thread
// dev_predict / dev_update_from_totals, whose steps are separated by workgroup barriers.  Same sums in the same order (the
// one-wave cores agree with the 256-thread kernels bit for bit, test_batch_replay_frozen_map; one tile = no cross-wave sum).
template <bool XID>
__global__ void __launch_bounds__(LK_WAVE)
    lk_tiny_bucket_kernel(LkMap map, LkParams pr, LkFilter* filters, const double* __restrict__ Q, int q_diag, double t,
                          const lk_point* __restrict__ pts, int n, float* world, int reproject) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    LkFilter* f = &filters[0];
    const int lane = threadIdx.x;
    dev_bucket_begin(map);
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    const double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    __syncthreads();
    wave_predict_core(sm, Q, t - t_upd, t - t_pred, lane, q_diag != 0);   // KILO.cc:111-115
    BucketConst bc;
#pragma unroll
    for (int i = 0; i < 9; ++i) bc.R[i] = sm.x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bc.p[i] = sm.x[9 + i];
    {
        const double* P = sm.P;
        bc.Prr = S3{P[0], P[1], P[2], P[31], P[32], P[62]};
        bc.Ppp = S3{P[3 * 30 + 3], P[3 * 30 + 4], P[3 * 30 + 5], P[4 * 30 + 4], P[4 * 30 + 5], P[5 * 30 + 5]};
    }
    ResidualOut ro;
    ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = world;
    const double a = residual_tile<false, 0, XID, true>(map, pr, bc, reinterpret_cast<const float4*>(pts), lane, n, rows, lane, ro, (size_t)0);
    const double totv = (lane < 29) ? a : 0.0;   // tot[j] in lanes 0..31
    const int N = (int)(lane_bcast<28>(totv) + 0.5);
    if (lane == 0) {   // the bookkeeping of dev_predict / dev_update_from_totals (KILO.cc:193,211-212)
        f->last_predict_t = t;
        f->n_buckets += 1;
        f->last_N = N;
        f->updated = N > 0;
        if (N > 0) {
            f->n_updates += 1;
            f->n_effect += (unsigned long long)N;
            f->last_update_t = t;
        }
    }
    if (N > 0) wave_update_core(sm, totv, N, lane);
    __syncthreads();
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    if (lane < 36) f->x[lane] = sm.x[lane];
    if (reproject) {
        __threadfence();
        __syncthreads();
        for (int i = lane; i < n; i += LK_WAVE) dev_reproject_point(map, pr, filters, pts, world, reproject == 2 ? 1 : 0, i);
    }
}
extern "C" {

// Ragged batch of SMALL buckets (a real scan: 2 ms time bins of tens of points): the whole bucket chain of a scan - predict,
// residual tiles, update, predict, ... - as ONE WAVE in one launch.  State and covariance stay in LDS from the first
// predict to the last update (WaveSmem), the bucket totals never leave the registers, and there is no launch boundary
// or partial-record round trip per bucket: the batch costs one scan's dependent chain, whatever the number of scans
// (up to the GPU's resident waves).  Arithmetic = residual_tile + wave_update_core + wave_predict_core, i.e. what the
// per-bucket launches of lk_batch_replay_ragged_dev compute; a bucket's tile totals are added in tile order, which is
// the order lk_update_wave_kernel uses for up to 8 tiles (one per group) - the host takes this path only when every bucket
// has <= LK_SCAN_WAVE_MAX points, so both paths give the same bits.
#ifdef LK_DEBUG_PHASES
__device__ unsigned long long lk_sw_dbg[16];   // DEBUG BUILD ONLY: s_memtime deltas per phase of dev_scan_wave, [15] = buckets
#define SW_STAMP(k) do { const unsigned long long t1_ = wall_clock64(); ph_[k] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define SW_STAMP(k) do { } while (0)
#endif
__device__ __forceinline__ void dev_scan_wave(const LkMap& map, const LkParams& pr, LkFilter* filters, const lk_point* __restrict__ pts,
                                              const LkRagged& rg, const double* __restrict__ Q, WaveSmem& sm, double* rows, const int MSG) {
    const bool WITH_IMU = MSG != 0;   // 1: lk_imu messages (only_imu_use), 2: lk_kin_imu messages (leg fusion, KILO.cc:384-390)
    const int slot = blockIdx.x, lane = threadIdx.x;
    LkFilter* f = &filters[slot];
    const int nbk = rag_nb(rg, slot);
    if (nbk == 0) return;
    const double* T = rag_t(rg, slot);
    const unsigned long long* po = rag_pt_off(rg, slot);
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    unsigned long long n_effect = f->n_effect;
    unsigned int n_updates = f->n_updates, n_buckets = f->n_buckets;
    int last_N = f->last_N, updated = f->updated;
    __syncthreads();
    unsigned int qi = 0, qn = 0;   // this scan's IMU messages (KILO.cc:379-383: those stamped before the bucket come first)
    if (WITH_IMU) qi = rg.imu_off[slot], qn = rg.imu_off[slot + 1];
    ResidualOut ro;
    ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = nullptr;
#ifdef LK_DEBUG_PHASES
    unsigned long long ph_[6] = {0, 0, 0, 0, 0, 0}, t0_ = wall_clock64();
#endif
    for (int b = 0; b < nbk;) {
        // next event of the scan: an IMU message stamped before the bucket's time (KILO.cc:379-383), else the bucket
        const double tb_ = T[b];
        const size_t mstride = MSG == 2 ? 33 : 7;
        const bool is_imu = WITH_IMU && qi < qn && rg.imu[mstride * (size_t)qi] < tb_;
        const double t = is_imu ? rg.imu[mstride * (size_t)qi] : tb_;
        SW_STAMP(0);
        if (!LK_X_NOPRED) wave_predict_core(sm, Q, t - t_upd, t - t_pred, lane, rg.q_diag != 0);   // KILO.cc:111-115 / :240-244
        SW_STAMP(1);
        t_pred = t;
        if (is_imu) {   // predictUpdateImu, KILO.cc:235-258 / predictUpdateKinImu, KILO.cc:260-314
            const double* m = rg.imu + mstride * (size_t)qi;
            if (MSG == 2)
                wave_kin_update_core(sm, rows, m, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else
                wave_imu_update_core(sm, m + 1, m + 4, rg.acc_scale, rg.Rn, lane);
            t_upd = t;  // KILO.cc:256 / :312
            ++qi;
            continue;
        }
        const unsigned long long base = po[b];
        const int n = (int)(po[b + 1] - base);
        BucketConst bc;   // load_bucket_const<false> from the LDS-resident state
#pragma unroll
        for (int i = 0; i < 9; ++i) bc.R[i] = sm.x[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bc.p[i] = sm.x[9 + i];
        {
            const double* P = sm.P;
            bc.Prr = S3{P[0], P[1], P[2], P[31], P[32], P[62]};
            bc.Ppp = S3{P[3 * 30 + 3], P[3 * 30 + 4], P[3 * 30 + 5], P[4 * 30 + 4], P[4 * 30 + 5], P[5 * 30 + 5]};
        }
        double totv = 0.0;  // tot[j] in lanes 0..31
        SW_STAMP(2);
        for (int i0 = 0; i0 < n; i0 += LK_WAVE) {
            __builtin_amdgcn_wave_barrier();  // the previous tile's reads of the rows are complete
            const double a = LK_X_NORES ? ((lane == 28) ? 1.0 : 0.0)
                                        : residual_tile<false, 2, false, true>(map, pr, bc, reinterpret_cast<const float4*>(pts + base), i0 + lane, n, rows, lane, ro, (size_t)0);
            totv += (lane < 29) ? a : 0.0;
        }
        const int N = (int)(lane_bcast<28>(totv) + 0.5);
        SW_STAMP(3);
        n_buckets += 1, last_N = N, updated = N > 0;
        if (N > 0) {
            n_updates += 1, n_effect += (unsigned long long)N;
            t_upd = t;  // KILO.cc:212
            if (!LK_X_NOUPD) wave_update_core(sm, totv, N, lane);
        }
        SW_STAMP(4);
        __syncthreads();
        ++b;
        SW_STAMP(5);
    }
#ifdef LK_DEBUG_PHASES
    if (lane == 0) {
        for (int k = 0; k < 6; ++k) atomicAdd(&lk_sw_dbg[k], ph_[k]);
        atomicAdd(&lk_sw_dbg[15], (unsigned long long)nbk);
    }
#endif
    __syncthreads();
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    if (lane < 36) f->x[lane] = sm.x[lane];
    if (lane == 0) {
        f->last_update_t = t_upd, f->last_predict_t = t_pred;
        f->n_effect = n_effect, f->n_updates = n_updates, f->n_buckets = n_buckets, f->last_N = last_N, f->updated = updated;
    }
}
// Ragged batch WITH insert, the way to bucket b of every scan that has one: the scan's messages stamped before the bucket's time that no
// earlier bucket has consumed (KILO.cc:379-390: predictUpdateImu / predictUpdateKinImu one after the other), then the predict to the bucket's
// time (KILO.cc:111-115) - dev_scan_wave's event loop between two buckets, as a launch of its own (one wave per scan).
__global__ void __launch_bounds__(LK_WAVE, 2)
    lk_rag_advance_kernel(LkFilter* filters, const double* __restrict__ Q, LkRagged rg, int b, int msg_kind) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (b >= rag_nb(rg, slot)) return;
    LkFilter* f = &filters[slot];
    const double* T = rag_t(rg, slot);
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    __syncthreads();
    const double tb = T[b];
    if (msg_kind) {
        const size_t mstride = msg_kind == 2 ? 33 : 7;
        const unsigned int q0 = rg.imu_off[slot], q1 = rg.imu_off[slot + 1];
        for (unsigned int q = q0; q < q1; ++q) {
            const double* m = rg.imu + mstride * (size_t)q;
            const double tm = m[0];
            if (!(tm < tb)) break;                  // time-sorted: the rest belongs to later buckets
            if (b > 0 && tm < T[b - 1]) continue;   // consumed on the way to an earlier bucket
            wave_predict_core(sm, Q, tm - t_upd, tm - t_pred, lane, rg.q_diag != 0);
            t_pred = tm;
            if (msg_kind == 2) wave_kin_update_core(sm, rows, m, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else wave_imu_update_core(sm, m + 1, m + 4, rg.acc_scale, rg.Rn, lane);
            t_upd = tm;   // KILO.cc:256 / :312
        }
    }
    wave_predict_core(sm, Q, tb - t_upd, tb - t_pred, lane, rg.q_diag != 0);
    t_pred = tb;
    __syncthreads();
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    if (lane < 36) f->x[lane] = sm.x[lane];
    if (lane == 0) f->last_update_t = t_upd, f->last_predict_t = t_pred;
}


__global__ void __launch_bounds__(LK_WAVE, 2)
    lk_scan_wave_kernel(LkMap map, LkParams pr, LkFilter* filters, const lk_point* __restrict__ pts, LkRagged rg,
                        const double* __restrict__ Q) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    dev_scan_wave(map, pr, filters, pts, rg, Q, sm, rows, 0);
}
__global__ void __launch_bounds__(LK_WAVE, 2)
    lk_scan_wave_imu_kernel(LkMap map, LkParams pr, LkFilter* filters, const lk_point* __restrict__ pts, LkRagged rg,
                            const double* __restrict__ Q) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    dev_scan_wave(map, pr, filters, pts, rg, Q, sm, rows, 1);
}
__global__ void __launch_bounds__(LK_WAVE, 2)
    lk_scan_wave_kin_kernel(LkMap map, LkParams pr, LkFilter* filters, const lk_point* __restrict__ pts, LkRagged rg,
                            const double* __restrict__ Q) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    dev_scan_wave(map, pr, filters, pts, rg, Q, sm, rows, 2);
}

// ------------------------------------------------------------------ scan-resident stream kernel
// A live scan of SMALL buckets (the reference's own shape: 2 ms time bins of a dozen points, hundreds per scan) as ONE launch of ONE
// workgroup (a filter wave + an insert team of three waves) that stays resident for the whole bucket loop of KILO::process (KILO.cc:375-395):
//   wave 0 (filter)  the chain of dev_scan_wave: the messages stamped before a bucket (predictUpdateImu / predictUpdateKinImu),
//                    predict, the bucket's residual tiles, update - state and covariance stay in LDS for the whole scan (the one-wave
//                    cores, MW = true: their barriers involve this wave only) - then the posterior's snapshot for the insert;
//   waves 1-3        the map insert of every bucket from that snapshot (KILO.cc:216-233): pool bookkeeping, re-projection + root
//   (insert team)    hashing, then the root pass / emitted groups / fallback items of the touched roots, the roots spread over the
//                    team; its phases are separated by a barrier of the team alone (an LDS counter).
// The two run as a PIPELINE: while wave 1 inserts bucket k, wave 0 already predicts and evaluates bucket k + 1 - speculatively,
// remembering which two roots every point looked at (SPEC codes).  Wave 1 stamps what an insert may change BEFORE it changes it
// (new roots in the re-projection; dev_stamp_dirty_roots: every touched root that is not a plain append) and says so
// (f_decided); wave 0 then keeps its tile sums if no point looked at a stamped root - they were computed from data no insert
// touched - and otherwise waits for the insert to finish (f_done) and evaluates the bucket's tiles again.  update(k + 1) therefore
// sees exactly the sums of the sequential order: results are bit-identical to the per-bucket launches
// (test_scan_resident_kernel_equals_per_bucket_launches).  Flags live in LDS; both waves sit on one CU, so workgroup-scope
// fences order the global-memory traffic between them.  Measured per bucket before the pipeline (one workgroup doing both in
// turn): predict 3.6 + tiles 3.4 + update 4.5 + snapshot 1.2 + re-projection 2.0 + insert 7.6 us = 23 us, the same as the per-bucket
// launches (their floor was never the cost); with the two chains side by side the bucket costs the longer of them.
// Host side: run_scan_resident().
#define LK_RESIDENT_MAX 512   // largest bucket (points) the resident kernel takes (= LK_SMALL_MAX): its tiles run one after the other in wave 0
}  // extern "C" (a kernel template follows)
// LDS flags between the waves of the resident workgroup (macros on the __shared__ variables themselves: through a pointer parameter
// the accesses became system-scope FLAT loads).  FLAG_WAIT: wave-uniform spin until the other side has posted `need`; evaluates to
// false when the wait was given up - another wave has raised f_abort, or this one does after LK_RESIDENT_TIMEOUT ticks of the 100 MHz
// clock (a device fault in the other role must fail the call, never hang the GPU): the caller leaves its bucket loop.
#define LK_RESIDENT_TIMEOUT_MS 2000u   // default bound of every wait inside the resident kernel (LEGKILO_RESIDENT_TIMEOUT_MS overrides)
#define LK_SPIN_UNTIL(cond, watch_exit)                                                                                      \
    ([&]() -> bool {                                                                                                          \
        unsigned long long t0_ = 0;                                                                                           \
        unsigned int spins_ = 0;                                                                                              \
        while (!(cond)) {                                                                                                     \
            if (__hip_atomic_load(&f_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;               \
            if ((watch_exit) && __hip_atomic_load(&f_exit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false; \
            __builtin_amdgcn_s_sleep(1);                                                                                      \
            if ((++spins_ & 1023u) == 0u) {                                                                                   \
                const unsigned long long now_ = wall_clock64();                                                               \
                if (t0_ == 0) t0_ = now_;                                                                                     \
                else if (now_ - t0_ > resident_timeout_) {                                                                  \
                    __hip_atomic_store(&f_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                          \
                    if ((threadIdx.x & 63) == 0) atomicOr(&map.counters[LK_CTR_ERR], LK_E_SPEC_TIMEOUT);                      \
                    return false;                                                                                             \
                }                                                                                                             \
            }                                                                                                                 \
        }                                                                                                                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                                                \
        return true;                                                                                                          \
    }())
#define FLAG_WAIT(flag, need) LK_SPIN_UNTIL(__hip_atomic_load(&(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (need), false)
// the filter wave's waits for the insert team: also given up (false) when the team has LEFT the launch with fallback items pending (f_exit)
#define FLAG_WAIT_X(flag, need) LK_SPIN_UNTIL(__hip_atomic_load(&(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (need), true)
#define FLAG_POST(flag, value)                                                                                              \
    do {                                                                                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* this wave's stores (global and LDS) are complete */        \
        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&(flag), (value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   \
    } while (0)
// barrier among the LK_INS_WAVES insert waves only (a monotonic LDS counter; `phase` counts this wave's arrivals); false = given up
#ifndef LK_INS_WAVES
#define LK_INS_WAVES 7   // with the filter wave: 512 threads = two waves per SIMD of one CU, 256 registers each
#endif
#define TEAM_BARRIER(ctr, phase)                                                                                            \
    (++(phase), __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"),                                                       \
     (((threadIdx.x & 63) == 0) ? (void)__hip_atomic_fetch_add(&(ctr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (void)0), \
     LK_SPIN_UNTIL(__hip_atomic_load(&(ctr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= LK_INS_WAVES * (phase), false))
// Where a scan stands between two launches of the resident kernel.  The generic fallback items of the insert (a voxel that has to be cut,
// leftovers after a flip to a tree, roots with more than 64 queued points: dev_insert_fallback, the per-point state machine) are NOT part of
// the resident kernel: their code alone needs 250 more registers and 6.3 KB of scratch per lane, which held the workgroup at one wave per
// SIMD = a team of three.  A config-1 stream meets such an item in a fraction of a percent of its buckets, a steady-state map in none.
// When the team finds one after a bucket's apply phase it records the bucket here and leaves; the filter wave stops at its next wait for
// the team (always with the predict to its bucket applied and that bucket's update not: stage1), both write their position, the host
// runs lk_resident_fallback_kernel and launches the resident kernel again, which picks up exactly there (run_scan_resident / resident_rounds).
struct LkResume {
    int bf;            // filter wave: next bucket
    int stage1;        // 1: the predict to bucket bf's time is applied, its messages are consumed - resume with the tiles
    unsigned int qi;   // message cursor
    int bi;            // insert team: next bucket
    int fb_bucket;     // the bucket whose fallback items are pending (its snapshot: snap2[fb_bucket & 1]); -1: none
    int pad_[3];
};
#ifndef LK_X_DYNROOT
#define LK_X_DYNROOT 1   // A/B builds: 0 = the grid-resident kernel's root pass strides the touched list
#endif
#ifndef LK_X_SLEEP
#define LK_X_SLEEP 0   // sensitivity probes (never in the product build): ~1 us of sleep per bucket on 1 the filter wave, 2 the insert team before / 4 behind its stamps
#endif
#ifdef LK_DEBUG_RES
__device__ unsigned long long lk_res_dbg[32];   // DEBUG BUILD ONLY: 100 MHz ticks per phase of the resident kernel's two roles; [0..7] filter wave, [8..15] insert wave 1, [31] buckets
__device__ unsigned long long lk_res_ts[6][1024];   // per bucket: 0 filter posted, 1 insert saw the post, 2 insert posted decided, 3 filter began to wait for decided, 4 filter saw decided, 5 insert done
#define RS_TS(k, b) do { if ((threadIdx.x & 63) == 0 && (b) < 1024) lk_res_ts[k][b] = wall_clock64(); } while (0)
#define RS_DECL unsigned long long rs_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rt0_ = wall_clock64()
#define RS_STAMP(k) do { const unsigned long long t1_ = wall_clock64(); rs_[k] += t1_ - rt0_; rt0_ = t1_; } while (0)
#define RS_FLUSH(o) do { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&lk_res_dbg[(o) + k_], rs_[k_]); } while (0)
#else
#define RS_TS(k, b) do { } while (0)
#define RS_DECL do { } while (0)
#define RS_STAMP(k) do { } while (0)
#define RS_FLUSH(o) do { } while (0)
#endif
template <int MSG, bool XID>
__global__ void __launch_bounds__((1 + LK_INS_WAVES) * LK_WAVE)
    lk_scan_stream_kernel(LkMap map, LkParams pr, LkFilter* filters, const lk_point* __restrict__ pts, LkRagged rg, const double* __restrict__ Q,
                          LkFilter* snap2 /* two snapshots */, float* world, int2* ids, unsigned int epoch0, unsigned int timeout_ms, LkResume* rs) {
    // bit 31 of timeout_ms (lk_test_stall: fault injection for the error-path test, never set otherwise): the insert team stops answering at
    // bucket 3, so the filter wave's bounded wait is given up and the call fails with LK_ERR_TIMEOUT
    const bool inject_stall = (timeout_ms >> 31) != 0u;
    const unsigned long long resident_timeout_ = (unsigned long long)(timeout_ms & 0x7fffffffu) * 100000ull;   // ticks of the 100 MHz wall clock
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    __shared__ int f_post, f_decided, f_done;   // bucket index of: latest posterior snapshot / stamps final / insert complete
    __shared__ int team_ctr;                    // arrivals at the insert team's barrier
    __shared__ int f_abort;                     // a wait was given up: every role leaves its loop
    __shared__ int f_exit;                      // the insert team has left with fallback items pending (LkResume)
    __shared__ unsigned int root_ticket;        // the root pass's next untaken root beyond the waves' own first ones (dev_insert_root's dyn_next)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    LkFilter* f = &filters[0];
    const int nbk = rag_nb(rg, 0);
    if (nbk == 0) return;
    const double* T = rag_t(rg, 0);
    const unsigned long long* po = rag_pt_off(rg, 0);
    const int bf0 = rs->bf, bi0 = rs->bi, stage1_0 = rs->stage1;   // (0, 0, 0) in a scan's first launch
    if (bf0 >= nbk && bi0 >= nbk) return;
    if (tid == 0) f_post = bf0 - 1, f_decided = bi0 - 1, f_done = bi0 - 1, team_ctr = 0, f_abort = 0, f_exit = 0;
    __syncthreads();
    if (wv >= 1) {
        // ================================================================= insert team (waves 1 .. LK_INS_WAVES)
        const int rank = wv - 1;
        int phase = 0;
        RS_DECL;
        // the pools' bookkeeping for a bucket's insert (dev_bucket_begin) only needs the PREVIOUS insert to be complete: it runs behind that
        // one (and once before the first), not between the posterior's arrival and the stamps the filter wave waits for
        if (rank == 0) dev_bucket_begin_wave(map);
        if (!TEAM_BARRIER(team_ctr, phase)) return;
        int b = bi0;
        for (; b < nbk; ++b) {
            const unsigned long long base = po[b];
            const int n = (int)(po[b + 1] - base);
            LkMap m = map;
            m.epoch = epoch0 + (unsigned int)b;
            const LkFilter* sn = snap2 + (b & 1);
            if (!FLAG_WAIT(f_post, inject_stall && b >= 3 ? nbk + 1 : b)) break;   // (injected stall: a post that never comes)
            if (rank == 0) RS_TS(1, b);
            RS_STAMP(0);
#if LK_X_SLEEP & 2
            if (rank == 0) __builtin_amdgcn_s_sleep(38);   // sensitivity probe: ~1 us on the insert team's chain, before its stamps are final
            if (!TEAM_BARRIER(team_ctr, phase)) break;
#endif
            RS_STAMP(1);
            for (int i = rank * LK_WAVE + lane; i < n; i += LK_INS_WAVES * LK_WAVE) dev_reproject_point(m, pr, sn, pts + base, world ? world + 4 * base : nullptr, 1, i);
            if (!TEAM_BARRIER(team_ctr, phase)) break;
            RS_STAMP(2);
            const int n_touched = (int)__hip_atomic_load(&m.counters[LK_CTR_TOUCHED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (rank == 0) {
                if (n_touched > 0) dev_stamp_dirty_roots(m, pr, n_touched);
                if (lane == 0) root_ticket = 0u;
                FLAG_POST(f_decided, b);
                RS_TS(2, b);
            }
            if (n_touched > 0) {
                if (!TEAM_BARRIER(team_ctr, phase)) break;   // the stamping pass has read the roots' queues before the root pass resets them
                RS_STAMP(3);
                dev_insert_root<false>(m, pr, sn, pts + base, (const lk_pt_rec*)nullptr, n, rank, LK_INS_WAVES, nullptr, nullptr, 0, LK_X_DYNROOT ? &root_ticket : nullptr);
                if (!TEAM_BARRIER(team_ctr, phase)) break;
                RS_STAMP(4);
                dev_insert_apply<false>(m, pr, sn, pts + base, (const lk_pt_rec*)nullptr, n, rank, LK_INS_WAVES);
                if (!TEAM_BARRIER(team_ctr, phase)) break;
                RS_STAMP(5);
                // generic fallback items: not in this kernel (LkResume) - every team wave reads the same count behind the barrier and leaves
                if (__hip_atomic_load(&m.counters[LK_CTR_FALLBACK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
#ifdef LK_DEBUG_RES
                    rs_[7] += 1;
#endif
                    if (rank == 0) {
                        if (lane == 0) rs->bi = b + 1, rs->fb_bucket = b;
                        FLAG_POST(f_exit, 1);
                    }
                    b = -1;
                    break;
                }
            }
#if LK_X_SLEEP & 4
            if (rank == 0) __builtin_amdgcn_s_sleep(38);   // sensitivity probe: ~1 us on the insert team's chain, behind its stamps
#endif
            if (!TEAM_BARRIER(team_ctr, phase)) break;
            if (rank == 0) FLAG_POST(f_done, b);
            if (rank == 0) RS_TS(5, b);
            if (rank == 0 && b + 1 < nbk) dev_bucket_begin_wave(map);   // for the next bucket (its re-projection is behind a team barrier of that bucket... the one below)
            if (!TEAM_BARRIER(team_ctr, phase)) break;
            RS_STAMP(6);
        }
        if (rank == 0 && lane == 0 && b == nbk) rs->bi = nbk, rs->fb_bucket = -1;   // (a wait given up: the call fails, LkResume is not read)
        if (rank == 0) RS_FLUSH(8);
        return;
    }
    // ===================================================================== filter wave
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    unsigned long long n_effect = f->n_effect;
    unsigned int n_updates = f->n_updates, n_buckets = f->n_buckets;
    int last_N = f->last_N, updated = f->updated;
    core_sync<true>();
    unsigned int qi = 0, qn = 0;   // the scan's messages (KILO.cc:379-390: those stamped before the bucket come first)
    if (MSG) qi = rs->qi, qn = rg.imu_off[1];
    constexpr size_t mstride = MSG == 2 ? 33 : 7;
    RS_DECL;
    bool predicted = stage1_0 != 0;   // picked up behind a predict (LkResume::stage1)
    bool stopped = false;             // left the loop in a wait for the insert team
    int b = bf0;
    while (b < nbk) {
        const double tb_ = T[b];
#if LK_X_SLEEP & 1
        __builtin_amdgcn_s_sleep(38);   // sensitivity probe: ~1 us on the filter wave's chain
#endif
        const bool is_msg = !predicted && MSG && qi < qn && rg.imu[mstride * (size_t)qi] < tb_;
        const double t = is_msg ? rg.imu[mstride * (size_t)qi] : tb_;
        if (!predicted) wave_predict_core<true>(sm, Q, t - t_upd, t - t_pred, lane, rg.q_diag != 0);   // KILO.cc:111-115 / :240-244
        predicted = false;
        t_pred = t;
        RS_STAMP(0);
        if (is_msg) {   // predictUpdateImu, KILO.cc:235-258 / predictUpdateKinImu, KILO.cc:260-314
            const double* mm = rg.imu + mstride * (size_t)qi;
            if (MSG == 2)
                wave_kin_update_core<true>(sm, rows, mm, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else
                wave_imu_update_core<true>(sm, mm + 1, mm + 4, rg.acc_scale, rg.Rn, lane);
            t_upd = t;  // KILO.cc:256 / :312
            ++qi;
            RS_STAMP(6);
            continue;
        }
        const unsigned long long base = po[b];
        const int n = (int)(po[b + 1] - base);
        BucketConst bc;   // load_bucket_const<false> from the LDS-resident state
#pragma unroll
        for (int i = 0; i < 9; ++i) bc.R[i] = sm.x[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bc.p[i] = sm.x[9 + i];
        {
            const double* P = sm.P;
            bc.Prr = S3{P[0], P[1], P[2], P[31], P[32], P[62]};
            bc.Ppp = S3{P[3 * 30 + 3], P[3 * 30 + 4], P[3 * 30 + 5], P[4 * 30 + 4], P[4 * 30 + 5], P[5 * 30 + 5]};
        }
        ResidualOut ro;
        ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr;
        ro.world = world ? world + 4 * base : nullptr;
        ro.ids = ids + base;
        int2 my_ids = make_int2(LK_SPEC_NONE, LK_SPEC_NONE);   // a one-tile bucket's lookup codes stay in the lane that made them
        ro.ids_lane = n <= LK_WAVE ? &my_ids : nullptr;
        // speculative pass (the insert of bucket b - 1, possibly the tail of b - 2, may be running beside it)
        // A bucket's tile sums are combined in the order lk_small_bucket_kernel combines them (its four waves take the tiles round
        // robin, then the wave sums are added in wave order): the two paths give the same bits for any bucket size.
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int i0 = 0; i0 < n; i0 += LK_WAVE) {
            __builtin_amdgcn_wave_barrier();  // the previous tile's reads of the rows are complete
            const double a = residual_tile<false, 0, XID, true, true>(map, pr, bc, reinterpret_cast<const float4*>(pts + base), i0 + lane, n, rows, lane, ro, (size_t)0);
            const int w4 = (i0 >> 6) & 3;
            if (w4 == 0) a0 += a; else if (w4 == 1) a1 += a; else if (w4 == 2) a2 += a; else a3 += a;
        }
        double totv = (lane < 29) ? (((0.0 + a0) + a1) + a2) + a3 : 0.0;  // tot[j] in lanes 0..31
        RS_STAMP(1);
        if (b > 0) {
            RS_TS(3, b - 1);
            if (!FLAG_WAIT_X(f_decided, b - 1)) { stopped = true; break; }   // the stamps of insert b - 1 are final (and insert b - 2 is complete)
            RS_TS(4, b - 1);
            RS_STAMP(2);
            const unsigned int e_b = epoch0 + (unsigned int)b;
            const unsigned int from = b >= 2 ? e_b - 2u : epoch0;
            bool susp = false;
            if (n <= LK_WAVE) {
                susp = lane < n && (spec_suspect(map, my_ids.x, from) || spec_suspect(map, my_ids.y, from));
            } else {
                for (int i = lane; i < n; i += LK_WAVE) {
                    const int2 c = ro.ids[i];
                    susp = susp || spec_suspect(map, c.x, from) || spec_suspect(map, c.y, from);
                }
            }
            if (__ballot(susp) != 0ull) {
                if (!FLAG_WAIT_X(f_done, b - 1)) { stopped = true; break; }
                if (lane == 0) atomicAdd(&map.counters[LK_CTR_RES_REDO], 1u);
                a0 = a1 = a2 = a3 = 0.0;
                for (int i0 = 0; i0 < n; i0 += LK_WAVE) {
                    __builtin_amdgcn_wave_barrier();
                    const double a = residual_tile<false, 0, XID, true, false>(map, pr, bc, reinterpret_cast<const float4*>(pts + base), i0 + lane, n, rows, lane, ro, (size_t)0);
                    const int w4 = (i0 >> 6) & 3;
                    if (w4 == 0) a0 += a; else if (w4 == 1) a1 += a; else if (w4 == 2) a2 += a; else a3 += a;
                }
                totv = (lane < 29) ? (((0.0 + a0) + a1) + a2) + a3 : 0.0;
            }
        }
        const int N = (int)(lane_bcast<28>(totv) + 0.5);
        n_buckets += 1, last_N = N, updated = N > 0;
        RS_STAMP(3);
        if (N > 0) {
            n_updates += 1, n_effect += (unsigned long long)N;
            t_upd = t;  // KILO.cc:212
            wave_update_core<true>(sm, totv, N, lane);
        }
        core_sync<true>();
        RS_STAMP(4);
        // the posterior for the insert (dev_snapshot_posterior's fields): the buffer of bucket b - 2 is free once that insert is done
        if (b >= 2 && !FLAG_WAIT(f_done, b - 2)) break;   // (never behind a team that has left: its f_decided(b - 1) came after f_done(b - 2))
        RS_STAMP(7);
        {
            LkFilter* sn = snap2 + (b & 1);
            for (int e = lane; e < 180; e += LK_WAVE) sn->P[e] = sm.P[e];
            if (lane < LK_STATE_DOUBLES) sn->x[lane] = sm.x[lane];
            if (lane == 0) sn->updated = N > 0, sn->last_N = N;
        }
        FLAG_POST(f_post, b);
        RS_TS(0, b);
        ++b;
        RS_STAMP(5);
    }
    RS_FLUSH(0);
#ifdef LK_DEBUG_RES
    if (lane == 0) atomicAdd(&lk_res_dbg[31], (unsigned long long)nbk);
#endif
    // a wait was given up (a fault or a pre-empted GPU): the filter keeps its PRE-SCAN state - the call fails with LK_ERR_TIMEOUT, the
    // map holds a partial insert (restore it from a checkpoint / blob and replay the scan)
    if (__hip_atomic_load(&f_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return;
    // the end of the scan, or the team has left with fallback items pending (stopped): bucket b is predicted to, not updated
    if (lane == 0) rs->bf = b, rs->stage1 = stopped ? 1 : 0, rs->qi = qi;
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    if (lane < 36) f->x[lane] = sm.x[lane];
    if (lane == 0) {
        f->last_update_t = t_upd, f->last_predict_t = t_pred;
        f->n_effect = n_effect, f->n_updates = n_updates, f->n_buckets = n_buckets, f->last_N = last_N, f->updated = updated;
    }
}
// The fallback items a resident kernel (scan-resident or grid-resident) left behind (LkResume::fb_bucket): the generic pass of that bucket's
// insert from the bucket's own snapshot, as a launch of its own between two launches of the resident kernel.  two_snaps: the scan-resident
// kernel's snapshots alternate (snap[b & 1]) and its buckets carry their own epoch; the grid-resident kernel has one snapshot, epoch as it is.
__global__ void __launch_bounds__(LK_MB)
    lk_resident_fallback_kernel(LkMap map, LkParams pr, const LkFilter* snap, const lk_point* __restrict__ pts, LkRagged rg, unsigned int epoch0, int two_snaps,
                                const LkResume* rs, unsigned int* grid_sync) {
    if (grid_sync && blockIdx.x == 0 && threadIdx.x < 4) grid_sync[threadIdx.x] = 0u;   // the grid-resident kernel's barrier words for its next launch
    const int b = rs->fb_bucket;
    if (b < 0) return;
    const unsigned long long* po = rag_pt_off(rg, 0);
    const unsigned long long base = po[b];
    const int n = (int)(po[b + 1] - base);
    LkMap m = map;
    if (two_snaps) m.epoch = epoch0 + (unsigned int)b;
    dev_insert_fallback<false>(m, pr, snap + (two_snaps ? (b & 1) : 0), pts + base, (const lk_pt_rec*)nullptr, n, (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6),
                               (int)((gridDim.x * LK_MB) >> 6));
}
extern "C" {

// ------------------------------------------------------------------ grid-resident stream kernel (large buckets)
// The bucket loop of KILO::process (KILO.cc:375-395) for a scan of LARGE buckets as ONE launch of G co-resident workgroups: the phases
// that are separate launches on the stream path - residual tiles | update + snapshot + bookkeeping | re-projection (workgroup 0 runs
// the next bucket's predict beside it) | root pass | emitted groups | fallback items - separated by GRID BARRIERS instead of kernel
// boundaries, and the last two phases only entered when the device counters say there is work for them (the host cannot know that
// without a synchronisation, the launch version always pays both).  Same device functions in the same order as enqueue_bucket():
// identical bits.  A barrier is the placement-independent hand-off of the CDNA guide: every wave drains its stores, the workgroup
// meets, thread 0 issues ONE agent-scope release, arrives on a global counter, polls it (relaxed), issues ONE agent-scope acquire
// (+ scalar-cache invalidate), the workgroup meets again.  Every wait is bounded: a timeout raises the abort word, every workgroup
// leaves, the call fails with LK_ERR_TIMEOUT and the filter gets its pre-scan state back (backup_filter).  On by default for scans whose
// buckets all hold 513 .. LK_GRIDSCAN_AUTO_MAX points (lk_stream_grid / LEGKILO_GRIDSCAN: 0 never, 2 whenever it applies); DESIGN.md section 6.
#define LK_GRIDSCAN_WG_MAX 128
#define LK_CTR_GRID_XCC 15   // LkMap.counters[15]: XCC ids (one bit each) the working blocks of the last one-XCD launch ran on
}  // extern "C" (a kernel template follows)
template <bool XID>
__global__ void __launch_bounds__(LK_FB)   // (compiled for two waves per SIMD - 256 registers, 240 B of spills - it is 6-10 % slower: profiles/EXPERIMENTS.md)
    lk_scan_grid_kernel(LkMap map, LkParams pr, LkFilter* filters, const lk_point* __restrict__ pts, LkRagged rg, const double* __restrict__ Q,
                        LkFilter* snap, float* world, double* partials, unsigned int* sync /* [0] arrivals, [1] abort, [2] XCC ids seen */, unsigned int timeout_ms,
                        int stride, int b0 /* first bucket; > 0: the predict to its time has been applied */, LkResume* rs) {
    __shared__ WaveSmem w;   // workgroup 0: the filter's covariance and state, resident for the whole scan (below)
    __shared__ double red[8][LK_NPART];
    __shared__ double tot[LK_NPART];
    __shared__ double rows[LK_FB / LK_WAVE][64 * LK_ROW2];
    __shared__ int s_abort, s_one_xcd;
    // stride 8: only the blocks b % 8 == 0 work, the others leave at once.  Blocks are OBSERVED to run on XCD b % 8 (no contract), so
    // the working ones normally share one XCD and its L2; whether they really do is checked on the device (HW_REG_XCC_ID of every
    // working block, below) and only then the barriers drop their L2 write-back
    if ((int)blockIdx.x % stride) return;
    const int G = (int)gridDim.x / stride, wg = (int)blockIdx.x / stride, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nbk = rag_nb(rg, 0);
    if (nbk == 0) return;
    const double* T = rag_t(rg, 0);
    const unsigned long long* po = rag_pt_off(rg, 0);
    const bool inject_stall = (timeout_ms >> 31) != 0u;   // lk_test_stall (fault injection for the error-path test): workgroup 1 leaves at bucket 3
    const unsigned long long timeout_ticks = (unsigned long long)(timeout_ms & 0x7fffffffu) * 100000ull;
    unsigned int phase = 0;
    if (tid == 0) {
        s_abort = 0, s_one_xcd = 0;
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        __hip_atomic_fetch_or(&sync[2], 1u << (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    auto grid_barrier = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        phase += 1;
        if (tid == 0) {
            // workgroups of ONE XCD share its L2: their drained stores (write-through from the CU) are what the others' L2 requests
            // see, no write-back of the L2 is needed - the acquire below (invalidate of this CU's vector L1) always is
            // (gfx942 / gfx950 behaviour, which is all this library is built for; the workgroup-scope release keeps the ordering in the
            // compiler's memory model without an L2 write-back)
            if (!s_one_xcd) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int need = (unsigned int)G * phase;
            unsigned long long t0 = 0;
            unsigned int spins = 0;
            while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    s_abort = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 255u) == 0u) {
                    const unsigned long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > timeout_ticks) {
                        __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicOr(&map.counters[LK_CTR_ERR], LK_E_SPEC_TIMEOUT);
                        s_abort = 1;
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_s_dcache_inv();
            if (phase == 1u && stride > 1) {   // every working block has arrived, so has its XCC id
                const unsigned int seen = __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_one_xcd = (seen & (seen - 1u)) == 0u;
                if (wg == 0) map.counters[LK_CTR_GRID_XCC] = seen;
            }
        }
        __syncthreads();
        return s_abort == 0;
    };
    // The filter is REPLICATED: every workgroup keeps the covariance and the state in its own LDS from the first predict to the last update
    // and runs the same predict and the same update on it (the one-wave cores of the batch replay in its wave 0: the arithmetic of
    // lk_update_snap_kernel / lk_insert_root_predict_kernel of the per-bucket launches; a deterministic function of the same inputs, so the
    // copies never differ).  Nothing of the filter then has to cross a grid barrier: the residual tiles read the predicted state, the
    // re-projection the posterior, from the workgroup's own LDS; the update needs the tiles' partial records (one barrier) and that is all.
    // Workgroup 0 is the copy of record: it keeps the bookkeeping words, writes the posterior's snapshot for the root pass (which is behind
    // the next barrier anyway) and the whole state when the launch ends (the scan's end, or fallback items pending); a launch given up leaves
    // filters[0] to the host's backup.
    LkFilter* f = &filters[0];
    double t_upd = 0.0, t_pred = 0.0;
    auto write_back = [&]() {   // workgroup 0, behind a barrier of its own
        for (int e = tid; e < 900; e += LK_FB) f->P[e] = w.P[e];
        if (tid < 36) f->x[tid] = w.x[tid];
    };
    auto bucket_const_of_w = [&](BucketConst& bc) {   // load_bucket_const<false> from the LDS-resident state
#pragma unroll
        for (int i = 0; i < 9; ++i) bc.R[i] = w.x[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) bc.p[i] = w.x[9 + i];
        const double* P = w.P;
        bc.Prr = S3{P[0], P[1], P[2], P[31], P[32], P[62]};
        bc.Ppp = S3{P[3 * 30 + 3], P[3 * 30 + 4], P[3 * 30 + 5], P[4 * 30 + 4], P[4 * 30 + 5], P[5 * 30 + 5]};
    };
    __syncthreads();
    for (int e = tid; e < 900; e += LK_FB) w.P[e] = f->P[e];
    if (tid < 36) w.x[tid] = f->x[tid];
    t_upd = f->last_update_t, t_pred = f->last_predict_t;
    if (!grid_barrier()) return;   // (every copy is loaded before workgroup 0 writes a word of filters[0]; the working blocks' XCC ids are in)
    if (b0 == 0) {   // KILO.cc:111-115 for the first bucket
        if (wv == 0) wave_predict_core<true>(w, Q, T[0] - t_upd, T[0] - t_pred, lane, rg.q_diag != 0);
        t_pred = T[0];
        if (wg == 0 && tid == 0) f->last_predict_t = T[0];
        __syncthreads();
    }
#ifdef LK_DEBUG_RES
    unsigned long long gt0_ = wall_clock64();
#define GS_STAMP(k) do { const unsigned long long t1_ = wall_clock64(); if (wg == 0 && tid == 0) atomicAdd(&lk_res_dbg[16 + (k)], t1_ - gt0_); gt0_ = t1_; } while (0)
#else
#define GS_STAMP(k) do { } while (0)
#endif
    for (int b = b0; b < nbk; ++b) {
        const unsigned long long base = po[b];
        const int n = (int)(po[b + 1] - base);
        const int ntiles = (n + LK_WAVE - 1) / LK_WAVE;
        const lk_point* bp = pts + base;
        float* bw = world ? world + 4 * base : nullptr;
        if (inject_stall && b >= 3 && wg == 1) return;   // (injected stall: the others' barrier wait is given up after the bound)
        // the pools' bookkeeping for this bucket's insert (the previous insert is complete, the tiles do not read what it touches)
        if (wg == (G > 1 ? 1 : 0)) dev_bucket_begin(map);
        {   // residual pass: tile t by wave t of the grid (lk_residual_kernel's body, one partial record per tile)
            BucketConst bc;
            bucket_const_of_w(bc);
            ResidualOut ro;
            ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = bw, ro.ids = nullptr;
            for (int tile = wg * (LK_FB / LK_WAVE) + wv; tile < ntiles; tile += G * (LK_FB / LK_WAVE)) {
                __builtin_amdgcn_wave_barrier();
                const double acc = residual_tile<false, 0, XID, false>(map, pr, bc, reinterpret_cast<const float4*>(bp), tile * LK_WAVE + lane, n, &rows[wv][0], lane, ro, (size_t)0);
                if (lane < LK_NPART) partials[(size_t)tile * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
            }
        }
        GS_STAMP(0);
        if (!grid_barrier()) return;
        GS_STAMP(1);
        // lk_update_snap_kernel in every workgroup: fixed-order sum of the tiles' records, update; workgroup 0: bookkeeping + snapshot
        dev_reduce_partials(partials, ntiles, red, tot);   // the sum of dev_update_reduce
        const int N = (int)(tot[28] + 0.5);
        if (wg == 0 && tid == 0) {   // the bookkeeping of dev_update_from_totals (KILO.cc:193,211-212)
            f->n_buckets += 1;
            f->last_N = N;
            f->updated = N > 0;
            if (N > 0) {
                f->n_updates += 1;
                f->n_effect += (unsigned long long)N;
                f->last_update_t = T[b];
            }
        }
        if (N > 0) {
            t_upd = T[b];
            if (wv == 0) wave_update_core<true>(w, lane < 32 ? tot[lane] : 0.0, N, lane);
        }
        __syncthreads();
        if (wg == 0) {   // dev_snapshot_posterior's fields, from LDS: what the root pass reads
            if (tid < LK_STATE_DOUBLES) snap->x[tid] = w.x[tid];
            if (tid < 180) snap->P[tid] = w.P[tid];
            if (tid == 0) snap->updated = N > 0, snap->last_N = N;
        }
        GS_STAMP(2);
        GS_STAMP(3);
        // re-projection + root hashing with the posterior (waves 1..3 of every workgroup, the posterior in their registers) while wave 0
        // propagates the workgroup's copy to the next bucket
        {
            BucketConst bc;
            bucket_const_of_w(bc);
            __syncthreads();   // every thread has read the posterior
            if (wv == 0) {
                if (b + 1 < nbk) wave_predict_core<true>(w, Q, T[b + 1] - t_upd, T[b + 1] - t_pred, lane, rg.q_diag != 0);
            } else {
                const int per = LK_FB - LK_WAVE;
                for (int i = wg * per + (tid - LK_WAVE); i < n; i += G * per) dev_reproject_point_bc(map, pr, bc, N > 0, bp, bw, 1, i);
            }
            if (b + 1 < nbk) {
                t_pred = T[b + 1];
                if (wg == 0 && tid == 0) f->last_predict_t = T[b + 1];
            }
        }
        GS_STAMP(4);
        if (!grid_barrier()) return;
        GS_STAMP(5);
        const int n_touched = (int)__hip_atomic_load(&map.counters[LK_CTR_TOUCHED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (n_touched > 0) {
            dev_insert_root<false>(map, pr, snap, bp, (const lk_pt_rec*)nullptr, n, wg * (LK_FB / LK_WAVE) + wv, G * (LK_FB / LK_WAVE), nullptr, nullptr, 0,
                                   LK_X_DYNROOT ? &map.counters[LK_CTR_HEAVY] : nullptr);   // (LK_CTR_HEAVY: zeroed by dev_bucket_begin, otherwise unused on this path)
            GS_STAMP(6);
            if (!grid_barrier()) return;
            GS_STAMP(7);
            const unsigned int n_groups = __hip_atomic_load(&map.counters[LK_CTR_GROUPS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_groups) {
                dev_insert_apply<false>(map, pr, snap, bp, (const lk_pt_rec*)nullptr, n, wg * (LK_FB / LK_WAVE) + wv, G * (LK_FB / LK_WAVE));
                GS_STAMP(8);
                if (!grid_barrier()) return;
                GS_STAMP(9);
            }
            // generic fallback items (dev_insert_fallback: 250 more registers, 6.5 KB of scratch per lane) are not part of this kernel: every
            // workgroup reads the same count behind the barrier and leaves; lk_resident_fallback_kernel runs them from the snapshot, the next
            // launch picks up at bucket b + 1, to whose time every copy of the filter has been propagated (LkResume, run_scan_grid)
            const unsigned int n_fb = __hip_atomic_load(&map.counters[LK_CTR_FALLBACK], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n_fb) {
                if (wg == 0) {
                    write_back();
                    if (tid == 0) rs->bf = b + 1, rs->bi = b + 1, rs->fb_bucket = b;
                }
                return;
            }
        }
    }
    if (wg == 0) {
        write_back();
        if (tid == 0) rs->bf = nbk, rs->bi = nbk, rs->fb_bucket = -1;
    }
}
extern "C" {

// ------------------------------------------------------------------ pipelined stream path
// A bucket's insert (re-projection + root hashing, light / group / apply / fallback passes) only feeds the NEXT bucket's matching,
// and only through the planes of the root voxels it refits, cuts or creates - with time buckets = azimuth sectors of a spinning
// LiDAR, a handful of voxels at the sector border.  So the insert of bucket k runs on its own HIP stream (`ins`), reading the
// posterior from a snapshot, while the main stream goes on with predict(k+1) and a SPECULATIVE residual pass (k+1) that also
// records which two roots every point looked at.  The insert stamps what it may change (LkMap::dirty / newroot, final once its
// light pass has run: event D); lk_verify_kernel then keeps the partial record of every tile that looked at unstamped roots only -
// by construction computed from data no insert touched - and re-evaluates the other tiles once the insert has completed
// (spec[LK_SPEC_DONE]; the verify waves wait on the device, bounded).  update(k+1) therefore sees exactly the sums the sequential
// order gives; what is gone from the critical chain is the insert:
//   main:  predict(k+1) -> residual_spec(k+1) -> [D_k] verify(k+1) -> update(k+1) + snapshot -> [U_k+1]
//   ins :  begin(k+1) | [U_k] re-project(k) -> light(k) -> [D_k] group(k) -> apply(k) -> fallback(k) (+ DONE = epoch k)
// Stamps carry the bucket's epoch; verify(e) treats stamps >= e - 2 as suspect (the residual pass of e may have overlapped the tail
// of insert e - 2; inserts <= e - 3 had completed before it started: D_(e-2) follows them on `ins`).
int spec_join(lk_handle* h) {
    if (!h->spec_open) return LK_OK;
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_I, 0));   // everything enqueued on the main stream from here on follows the inserts
    h->spec_open = false;
    h->spec_base = h->epoch + 1;
    return LK_OK;
}
// every launch of the pipelined path is checked where it is issued (a bad launch configuration must name its kernel, not the last one)
#define SPEC_LAUNCH(...)                                \
    do {                                                \
        __VA_ARGS__;                                    \
        HIPCHK(h, hipGetLastError());                   \
    } while (0)
static int enqueue_bucket_spec(lk_handle* h, const lk_point* d_pts, int n, double t, float* d_world, bool xid) {
    const int nblk = (n + LK_PB - 1) / LK_PB;
    const int nblk_r = (n + LK_RB - 1) / LK_RB;
    if (h->epoch >= 0xfffffff0u) {   // stamps are compared as plain unsigned numbers: start over long before they could wrap
        int rc = spec_join(h);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        HIPCHK(h, hipMemsetAsync(h->map.dirty, 0, sizeof(unsigned int) * (size_t)h->map.max_nodes, h->stream));
        HIPCHK(h, hipMemsetAsync(h->map.newroot, 0, sizeof(unsigned int) * (size_t)(LK_NEWROOT_MASK + 1), h->stream));
        HIPCHK(h, hipMemsetAsync(h->map.spec, 0, sizeof(unsigned int) * LK_SPEC_WORDS, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->epoch = 16, h->spec_base = 17;
    }
    const unsigned int e = ++h->epoch;
    const bool first = !h->spec_open;            // nothing in flight: no verify needed for this bucket
    LkMap m = h->map;
    m.epoch = e;
    LkFilter* snap = h->d_snap + (e & 1u);
    if (first) {   // the insert stream follows whatever the main stream did to the map before
        HIPCHK(h, hipEventRecord(h->ev_I, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->ins, h->ev_I, 0));
    }
    // insert stream, ahead of the posterior: the bucket's pool bookkeeping
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_bucket_begin_kernel, dim3(1), dim3(256), 0, h->ins, m));
    // main stream
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_predict_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters, h->d_Q, t));
    ResidualOut ro;
    memset(&ro, 0, sizeof(ro));
    ro.world = d_world;
    ro.ids = h->d_ids;
    if (first) {
        const auto res_kernel = xid ? lk_residual_kernel<false, 0, true> : lk_residual_kernel<false, 0, false>;
        SPEC_LAUNCH(hipLaunchKernelGGL(res_kernel, dim3(nblk_r, 1), dim3(LK_RB), 0, h->stream, m, h->pr, h->d_filters, d_pts, (size_t)0, n, h->d_partials,
                           h->part_stride, ro, (size_t)0));
    } else {
        const auto res_kernel = xid ? lk_residual_kernel<false, 0, true, true> : lk_residual_kernel<false, 0, false, true>;
        SPEC_LAUNCH(hipLaunchKernelGGL(res_kernel, dim3(nblk_r, 1), dim3(LK_RB), 0, h->stream, m, h->pr, h->d_filters, d_pts, (size_t)0, n, h->d_partials,
                           h->part_stride, ro, (size_t)0));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_D[(e - 1) & 1u], 0));
        const unsigned int from = std::max(e - 2, h->spec_base);
        const auto ver_kernel = xid ? lk_verify_kernel<true> : lk_verify_kernel<false>;
        SPEC_LAUNCH(hipLaunchKernelGGL(ver_kernel, dim3(nblk_r), dim3(LK_RB), 0, h->stream, m, h->pr, h->d_filters, d_pts, n, h->d_partials, ro, from, e - 1));
    }
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_update_snap_kernel, dim3(1), dim3(LK_FB), 0, h->stream, m, h->d_filters, h->d_partials, nblk_r * (LK_RB / LK_WAVE), t, h->d_Q, 0.0, -1, snap));
    HIPCHK(h, hipEventRecord(h->ev_U[e & 1u], h->stream));
    // insert stream: the bucket's insert, from the snapshot of its posterior
    HIPCHK(h, hipStreamWaitEvent(h->ins, h->ev_U[e & 1u], 0));
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_reproject_kernel, dim3(nblk), dim3(LK_PB), 0, h->ins, m, h->pr, snap, d_pts, n, d_world, 1));
    const int grid = std::min(std::max((n + 3) / 4, 1), 512);
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_insert_root_kernel<false>, dim3(grid), dim3(LK_MB), 0, h->ins, m, h->pr, snap, d_pts, (const lk_pt_rec*)nullptr, n));
    HIPCHK(h, hipEventRecord(h->ev_D[e & 1u], h->ins));   // the stamps are final: new roots (re-projection), roots whose planes may change (root pass)
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_insert_apply_kernel<false>, dim3(grid), dim3(LK_MB), 0, h->ins, m, h->pr, snap, d_pts, (const lk_pt_rec*)nullptr, n));
    SPEC_LAUNCH(hipLaunchKernelGGL(lk_insert_fallback_kernel<false>, dim3(std::min(grid, 8)), dim3(LK_MB), 0, h->ins, m, h->pr, snap, d_pts, (const lk_pt_rec*)nullptr, n));
    HIPCHK(h, hipEventRecord(h->ev_I, h->ins));
    HIPCHK(h, hipGetLastError());
    h->spec_open = true;
    h->spec_buckets += 1;
    if (!first) h->spec_tiles += (uint64_t)nblk_r;
    return LK_OK;
}

// The root pass of a large bucket with the NEXT bucket's predict beside it: workgroup 0 is the predict (it reads and writes the live
// filter only; the insert passes read the posterior's snapshot), the others are lk_insert_root_kernel.  The predict's ~5 us
// disappear behind the 30-40 us root pass instead of lengthening the update launch every later kernel of the bucket waits for.
static_assert(LK_MB == LK_FB, "the predict workgroup runs in the root kernel's launch shape");
extern "C++" __global__ void __launch_bounds__(LK_MB)
    lk_insert_root_predict_kernel(LkMap map, LkParams pr, const LkFilter* snap, const lk_point* __restrict__ pts, int n, LkFilter* live,
                                  const double* __restrict__ Q, double t_next, int q_diag) {
    if (blockIdx.x == 0) {
        // the predict as ONE wave on the one-wave core (7.7 KB of LDS instead of the 256-thread predict's 38 KB in every workgroup
        // of this launch; the same bits: the scan-resident kernel runs this core against the 256-thread kernels in the tests)
        __shared__ WaveSmem sm;
        if (threadIdx.x >= LK_WAVE) return;
        LkFilter* f = &live[0];
        const int lane = threadIdx.x;
        for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
        if (lane < 36) sm.x[lane] = f->x[lane];
        const double t_upd = f->last_update_t, t_pred = f->last_predict_t;
        core_sync<true>();
        wave_predict_core<true>(sm, Q, t_next - t_upd, t_next - t_pred, lane, q_diag != 0);   // KILO.cc:111-115
        core_sync<true>();
        for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
        if (lane < 36) f->x[lane] = sm.x[lane];
        if (lane == 0) f->last_predict_t = t_next;
        return;
    }
    dev_insert_root<false>(map, pr, snap, pts, (const lk_pt_rec*)nullptr, n, (int)(((blockIdx.x - 1) * LK_MB + threadIdx.x) >> 6),
                           (int)(((gridDim.x - 1) * LK_MB) >> 6));
}

// lk_reproject_kernel's body in one-wave workgroups for the stream path: the pass is a chain of dependent round trips per point (scan
// point -> hash slot -> node walk -> the root's queue counter), so it wants every CU, not throughput per CU - 313 single-wave
// workgroups instead of 79 of four waves: 12.9 -> 11.1 us per 20 000-point bucket, 6.5 -> 6.0 us at 1 960 points (kernel trace, same box)
extern "C++" __global__ void __launch_bounds__(LK_WAVE)
    lk_reproject_wave_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts, int n,
                             float* __restrict__ world, int do_insert) {
    const int i = blockIdx.x * LK_WAVE + threadIdx.x;
    if (i >= n) return;
    dev_reproject_point(map, pr, filters, pts, world, do_insert, i);
}

// ------------------------------------------------------------------ one time bucket on the stream (no sync)
// predict -> residual (+A,b partials) -> 6x6 update -> re-project + hash -> per-root insert
// t_next: time of the NEXT bucket if the caller knows that it follows directly (no IMU / kinematic message in between) and is itself
// a large bucket - its predict then runs in this bucket's launch (`*pre_predicted` tells the next call) -, NaN otherwise.
// dynamic-LDS padding knobs of the stream launches (placement experiments): clamped to what a workgroup may ask for on top of its static LDS
static int lds_knob(const char* name) {
    const char* e = getenv(name);
    return e ? std::min(std::max(atoi(e), 0), 48 * 1024) : 0;
}
static int enqueue_bucket(lk_handle* h, const lk_point* d_pts, int n, double t, float* d_world, bool do_insert, double t_next = NAN,
                          bool* pre_predicted = nullptr) {
    const LkMap& m = h->map;
    const bool was_pre = pre_predicted && *pre_predicted;   // filters[0] already stands at this bucket's time
    if (pre_predicted) *pre_predicted = false;
    if (do_insert) h->grid_valid = false;   // the map changes: batch replay rebuilds its root grid
    const int nblk = (n + LK_PB - 1) / LK_PB;
    const int nblk_r = (n + LK_RB - 1) / LK_RB;
    const LkFilter* ins_filters = h->d_filters;   // what the insert reads the posterior from (large buckets: its snapshot)
    bool predict_in_root = false;                 // large buckets: the next bucket's predict rides in the root pass's launch
    if (do_insert && h->spec_enable && !h->profiling && n > LK_SMALL_MAX && !was_pre) {
        static const bool xid_en = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
        return enqueue_bucket_spec(h, d_pts, n, t, d_world, h->pr.ext_identity && xid_en);
    }
    if (h->spec_open) {   // a bucket on the sequential path follows the inserts in flight
        int rcj = spec_join(h);
        if (rcj) return rcj;
    }
    // the stream path's residual code specialised for ext_R == I like the batch kernel (LEGKILO_XID=0: generic)
    static const bool xid_enable = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const bool xid = h->pr.ext_identity && xid_enable;
    // tiny buckets (a real scan's 2 ms bins: a dozen points): fewer dependent launches - re-projection inside the bucket kernel, light +
    // group pass as one launch (LEGKILO_FUSE_MAX: largest such bucket, 0 = off)
    static const int fuse_max = getenv("LEGKILO_FUSE_MAX") ? atoi(getenv("LEGKILO_FUSE_MAX")) : 64;
    const bool fuse = n <= fuse_max && n <= LK_SMALL_MAX;
    static const bool tiny_enable = getenv("LEGKILO_TINY") == nullptr || atoi(getenv("LEGKILO_TINY")) != 0;
    if (fuse && tiny_enable && n <= LK_WAVE) {
        const auto tiny_kernel = xid ? lk_tiny_bucket_kernel<true> : lk_tiny_bucket_kernel<false>;
        LAUNCH(h, "small_bucket", hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(LK_WAVE), 0, h->stream, m, h->pr, h->d_filters, h->d_Q,
                                                     h->q_diag ? 1 : 0, t, d_pts, n, d_world, (d_world || do_insert) ? (do_insert ? 2 : 1) : 0));
    } else if (n <= LK_SMALL_MAX) {
        const auto small_kernel = xid ? lk_small_bucket_kernel<true> : lk_small_bucket_kernel<false>;
        LAUNCH(h, "small_bucket", hipLaunchKernelGGL(small_kernel, dim3(1), dim3(LK_FB), 0, h->stream, m, h->pr, h->d_filters,
                                                     h->d_Q, t, d_pts, n, d_world, fuse && (d_world || do_insert) ? (do_insert ? 2 : 1) : 0));
    } else {
        // residual pass, then ONE single-workgroup launch for everything else on the filter side: the fixed-order sum of the tiles'
        // partial records + the update, the posterior's snapshot for the insert, the insert's pool bookkeeping and - t_next known -
        // the next bucket's predict.  (The same work in the LAST wave of the residual launch - a ticket per wave, the one-wave
        // filter cores - measured slower: 38-57 us for the launch against 16 + 12, profiles/r03e_timeline_fused_last_wave_update_rejected.txt.)
        if (!was_pre) LAUNCH(h, "predict", hipLaunchKernelGGL(lk_predict_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters, h->d_Q, t));
        ResidualOut ro;
        memset(&ro, 0, sizeof(ro));
        ro.world = d_world;
        const bool fuse_next = pre_predicted != nullptr && t_next == t_next;
        static const bool predict_in_root_on = getenv("LEGKILO_PREDICT_IN_ROOT") == nullptr || atoi(getenv("LEGKILO_PREDICT_IN_ROOT")) != 0;
        predict_in_root = fuse_next && do_insert && predict_in_root_on;   // n > LK_SMALL_MAX here: the insert below is the three-launch form
        const auto res_kernel = xid ? lk_residual_kernel<false, 0, true> : lk_residual_kernel<false, 0, false>;
        static const int lds_res = lds_knob("LEGKILO_LDS_RES");
        LAUNCH(h, "residual", hipLaunchKernelGGL(res_kernel, dim3(nblk_r, 1), dim3(LK_RB), lds_res, h->stream, m, h->pr, h->d_filters, d_pts, (size_t)0, n,
                                                 h->d_partials, h->part_stride, ro, (size_t)0));
        LAUNCH(h, "update", hipLaunchKernelGGL(lk_update_snap_kernel, dim3(2), dim3(LK_FB), 0, h->stream, m, h->d_filters, h->d_partials,
                                               nblk_r * (LK_RB / LK_WAVE), t, h->d_Q, fuse_next ? t_next : 0.0, fuse_next && !predict_in_root ? 1 : 0, h->d_snap));
        if (fuse_next) *pre_predicted = true;
        ins_filters = h->d_snap;
    }
    static const int lds_rp = lds_knob("LEGKILO_LDS_REPROJ");
    static const int lds_root = lds_knob("LEGKILO_LDS_ROOT");
    static const int lds_apply = lds_knob("LEGKILO_LDS_APPLY");
    static const int lds_rootp = lds_knob("LEGKILO_LDS_ROOTP");
    if ((d_world || do_insert) && !fuse)
        LAUNCH(h, "reproject", hipLaunchKernelGGL(lk_reproject_wave_kernel, dim3((n + LK_WAVE - 1) / LK_WAVE), dim3(LK_WAVE), lds_rp, h->stream, m, h->pr,
                                                  ins_filters, d_pts, n, d_world, do_insert ? 1 : 0));
    if (do_insert) {
        // one wave per touched root (append / group / apply of single-group roots), then one wave per emitted leaf group (2 resident
        // waves per SIMD at ~200 VGPRs: 512 blocks x 4 waves is one resident round on 256 CUs), then the generic fallback for the few
        // groups that need it; all loops are grid-stride and read their work counts on the device
        static const int root_grid_cap = getenv("LEGKILO_ROOT_GRID") ? std::max(1, atoi(getenv("LEGKILO_ROOT_GRID"))) : 512;
        int grid = std::min(std::max((n + 3) / 4, 1), root_grid_cap);
        static const bool small_insert = getenv("LEGKILO_SMALL_INSERT") == nullptr || atoi(getenv("LEGKILO_SMALL_INSERT")) != 0;
        if (n <= LK_SMALL_MAX && small_insert) {   // small bucket: root pass + (in the last workgroup) apply + fallback as one launch
            LAUNCH(h, "insert_root", hipLaunchKernelGGL(lk_insert_small_kernel, dim3(std::min(grid, 128)), dim3(LK_MB), 0, h->stream, h->map, h->pr,
                                                        ins_filters, d_pts, n));
            return LK_OK;
        }
        if (predict_in_root)
            LAUNCH(h, "insert_root", hipLaunchKernelGGL(lk_insert_root_predict_kernel, dim3(grid + 1), dim3(LK_MB), lds_rootp, h->stream, h->map, h->pr,
                                                        ins_filters, d_pts, n, h->d_filters, h->d_Q, t_next, h->q_diag ? 1 : 0));
        else
            LAUNCH(h, "insert_root", hipLaunchKernelGGL(lk_insert_root_kernel<false>, dim3(grid), dim3(LK_MB), lds_root, h->stream, h->map, h->pr,
                                                        ins_filters, d_pts, (const lk_pt_rec*)nullptr, n));
        LAUNCH(h, "insert", hipLaunchKernelGGL(lk_insert_apply_kernel<false>, dim3(grid), dim3(LK_MB), lds_apply, h->stream, h->map, h->pr,
                                               ins_filters, d_pts, (const lk_pt_rec*)nullptr, n));
        LAUNCH(h, "insert_fallback", hipLaunchKernelGGL(lk_insert_fallback_kernel<false>, dim3(std::min(grid, 8)), dim3(LK_MB), 0, h->stream,
                                                        h->map, h->pr, ins_filters, d_pts, (const lk_pt_rec*)nullptr, n));
    }
    return LK_OK;
}

// The map as the batch-replay kernels see it: with the dense root grid when it can be built (LkMap::grid).  The grid is derived
// from the hash table + match records and rebuilt when the map has changed since (every mutating entry clears grid_valid);
// building it is two small kernels + a memset, synchronised once - it happens per map snapshot, not per batch.
static constexpr size_t kGridMaxCells = (size_t)1 << 24;   // 16 Mi cells x 144 B = 2.4 GB of grid at most (record INDICES are 32-bit, addressing is 64-bit); larger boxes stay on the hash
int frozen_map(lk_handle* h, LkMap* out) {
    *out = h->map;
    out->grid_on = 0;
    if (!h->grid_enable) return LK_OK;
    if (!h->grid_valid) {
        // the rebuild rewrites the cells (and may move the match pool): nothing enqueued earlier - on any of the handle's streams -
        // may still be reading them
        for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)
            if (h->side[i]) HIPCHK(h, hipStreamSynchronize(h->side[i]));
        const int init[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
        int mm[6];
        HIPCHK(h, hipMemcpyAsync(h->d_grid_mm, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(lk_grid_bounds_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, h->stream, h->map, h->hash_cap, h->d_grid_mm);
        HIPCHK(h, hipMemcpyAsync(mm, h->d_grid_mm, sizeof(mm), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        LkMap fm = h->map;
        fm.grid_on = 0;
        if (mm[0] <= mm[3]) {
            size_t dim[3], cells = 1;
            bool ok = true;
            for (int c = 0; c < 3; ++c) {
                dim[c] = (size_t)((long long)mm[3 + c] - (long long)mm[c] + 1);
                ok = ok && dim[c] <= kGridMaxCells;
                cells = ok ? cells * dim[c] : cells;
                ok = ok && cells <= kGridMaxCells;
            }
            if (ok && cells > h->grid_cap) {   // grow the match pool: [max_nodes node records | grid cells (+ slack) | flattened subtree lists]
                lk_match_rec* bigger = nullptr;
                const size_t want = cells + cells / 4;
                if (hipMalloc(&bigger, (2 * (size_t)h->map.max_nodes + want) * sizeof(lk_match_rec)) != hipSuccess) {
                    (void)hipGetLastError();   // no room for the grid: stay on the hash table
                    ok = false;
                } else {
                    HIPCHK(h, hipMemcpyAsync(bigger, h->map.match, (size_t)h->map.max_nodes * sizeof(lk_match_rec), hipMemcpyDeviceToDevice, h->stream));
                    HIPCHK(h, hipStreamSynchronize(h->stream));
                    hipFree(h->map.match);
                    h->map.match = bigger;
                    h->grid_cap = want;
                    fm.match = bigger;
                }
            }
            if (ok) {
                for (int c = 0; c < 3; ++c) fm.gmin[c] = mm[c], fm.gdim[c] = (int)dim[c];
                fm.grid_base = h->map.max_nodes;
                HIPCHK(h, hipMemsetAsync(fm.match + fm.grid_base, 0xff, cells * sizeof(lk_match_rec), h->stream));
                // the lists (every non-root node at most once) live behind the allocated cells
                const unsigned int cand0 = (unsigned int)((size_t)h->map.max_nodes + h->grid_cap);
                unsigned int* cursor = reinterpret_cast<unsigned int*>(h->d_grid_mm + 6);   // [0] next free list record, [1] overflow flag
                const unsigned int cur0[2] = {cand0, 0u};
                unsigned int cur1[2] = {0u, 0u};
                HIPCHK(h, hipMemcpyAsync(cursor, cur0, sizeof(cur0), hipMemcpyHostToDevice, h->stream));
                hipLaunchKernelGGL(lk_grid_fill_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, h->stream, fm, h->hash_cap, h->pr.max_layer,
                                   cursor, cand0 + h->map.max_nodes);
                HIPCHK(h, hipGetLastError());
                HIPCHK(h, hipMemcpyAsync(cur1, cursor, sizeof(cur1), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));   // the side streams of the replay entries may read it at once
                fm.grid_on = cur1[1] ? 0 : 1;   // list region overflowed (a blob whose subtrees share nodes): this snapshot stays on the hash table
            }
        }
        h->fmap = fm;
        h->grid_valid = true;
        ++h->map_gen;
    }
    *out = h->fmap;
    return LK_OK;
}

// The residual kernel of the uniform batch entries, specialised at compile time for what is launch-uniform: root lookup through
// the frozen-map grid, and ext_R == I (no 3 x 3 extrinsic products, and 18 fewer scalar registers in a kernel whose occupancy
// is set by registers).  LEGKILO_XID=0 keeps the generic instantiation (A/B).
using ResidualKernelFn = void (*)(LkMap, LkParams, const LkFilter*, const lk_point*, size_t, int, double*, size_t, ResidualOut, size_t);
static ResidualKernelFn batch_residual_kernel(const lk_handle* h, const LkMap& fmap) {
    static const bool xid_enable = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    if (!fmap.grid_on) return lk_residual_kernel<false, 0, false>;
    static const bool pair = getenv("LEGKILO_RES_PAIR") != nullptr && atoi(getenv("LEGKILO_RES_PAIR")) != 0;   // round-5 experiment: two tiles per wave (A/B; off)
    if (pair) return (h->pr.ext_identity && xid_enable) ? lk_residual_pair_kernel<true> : lk_residual_pair_kernel<false>;
    return (h->pr.ext_identity && xid_enable) ? lk_residual_kernel<false, 1, true> : lk_residual_kernel<false, 1, false>;
}

// Grid of a batch residual launch over `sn` slots x `nblk` tiles: plain 2-D grid (tile, slot); LEGKILO_XCDMAP=1: the XCD-aware 1-D grid of ResidualOut::xmap_slots (A/B)
static dim3 batch_residual_grid(int nblk, int sn, ResidualOut* ro) {
    static const bool xmap = getenv("LEGKILO_XCDMAP") != nullptr && atoi(getenv("LEGKILO_XCDMAP")) != 0;   // measured slower (EXPERIMENTS.md, round 6): off unless asked for
    const unsigned long long wg = 8ull * (unsigned long long)((nblk + 7) / 8) * (unsigned long long)sn;
    if (!xmap || sn < 2 || nblk < 16 || wg >= (1ull << 31)) {
        ro->xmap_slots = 0;
        return dim3(nblk, sn);
    }
    ro->xmap_slots = sn;
    return dim3((unsigned int)wg);
}

__global__ void lk_zero_scan_counters_kernel(LkFilter* filters, unsigned int n_slots) {
    const unsigned int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    LkFilter* f = &filters[s];
    f->n_effect = 0ull, f->n_updates = 0u, f->n_buckets = 0u, f->updated = 0, f->last_N = 0;
}
int zero_scan_counters(lk_handle* h, uint32_t first_slot, uint32_t n_slots) {
    // n_effect, n_updates, n_buckets, updated, last_N of every slot: ONE launch (a 24-byte-wide 2-D memset is two fill kernels)
    hipLaunchKernelGGL(lk_zero_scan_counters_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, h->stream, h->d_filters + first_slot, n_slots);
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}

__global__ void lk_pose_gather_kernel(const LkFilter* filters, lk_pose* out, int n) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const LkFilter* f = &filters[s];
    lk_pose p;
    for (int i = 0; i < 9; ++i) p.rot[i] = f->x[i];
    for (int i = 0; i < 3; ++i) p.pos[i] = f->x[9 + i], p.vel[i] = f->x[12 + i];
    p.n_effect = f->n_effect;
    p.n_buckets = f->n_buckets;
    p.n_updates = f->n_updates;
    out[s] = p;
}
int fetch_poses(lk_handle* h, lk_pose* out, int n) {
    hipLaunchKernelGGL(lk_pose_gather_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d_filters, h->d_poses, n);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_poses, sizeof(lk_pose) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

// End of a stream-path scan: the pose of filter slot 0 and the map's counter words (pool overflow / timeout bits) written by ONE kernel
// straight into host-mapped pinned memory, ONE stream synchronisation - instead of a gather kernel, two pageable device-to-host copies
// and two synchronisations (round 5: ~100 us of every scan's 430 were spent between its last kernel and the next scan's first).
__global__ void lk_scan_finish_kernel(const LkFilter* filters, const unsigned int* counters, lk_handle::ScanResult* out, unsigned int seq, const int* resume) {
    const int i = threadIdx.x;
    if (i == 1) {   // LkResume { bf, stage1, qi, bi, fb_bucket } of a scan-resident launch
        out->resume[0] = resume ? resume[0] : 0, out->resume[1] = resume ? resume[3] : 0, out->resume[2] = resume ? resume[4] : -1, out->resume[3] = 0;
    }
    if (i == 0) {
        const LkFilter* f = &filters[0];
        lk_pose p;
        for (int k = 0; k < 9; ++k) p.rot[k] = f->x[k];
        for (int k = 0; k < 3; ++k) p.pos[k] = f->x[9 + k], p.vel[k] = f->x[12 + k];
        p.n_effect = f->n_effect, p.n_buckets = f->n_buckets, p.n_updates = f->n_updates;
        out->pose = p;
    }
    if (i < LK_CTR_COUNT) out->ctr[i] = counters[i];
    __threadfence_system();
    __syncthreads();
    if (i == 0) {
        __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static int finish_scan(lk_handle* h, lk_pose* pose, const void* d_resume = nullptr) {
    if (!h->h_result) {
        HIPCHK(h, hipHostMalloc((void**)&h->h_result, sizeof(lk_handle::ScanResult), hipHostMallocMapped));
        memset(h->h_result, 0, sizeof(lk_handle::ScanResult));
        HIPCHK(h, hipHostGetDevicePointer((void**)&h->d_result, h->h_result, 0));
    }
    const unsigned int seq = ++h->result_seq ? h->result_seq : ++h->result_seq;   // never 0 (the buffer's initial value)
    hipLaunchKernelGGL(lk_scan_finish_kernel, dim3(1), dim3(64), 0, h->stream, h->d_filters, h->map.counters, h->d_result, seq, static_cast<const int*>(d_resume));
    HIPCHK(h, hipGetLastError());
    // a scan is a fraction of a millisecond to a few: the caller's thread POLLS the sequence word the kernel writes last (a blocking
    // synchronisation wakes through an interrupt, 10-20 us later) - for at most 20 ms, then it blocks (which also surfaces device errors).
    // LEGKILO_SPIN_WAIT=0: always block
    static const bool spin = getenv("LEGKILO_SPIN_WAIT") == nullptr || atoi(getenv("LEGKILO_SPIN_WAIT")) != 0;
    bool seen = false;
    if (spin) {
        volatile unsigned int* sq = &h->h_result->seq;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
        for (unsigned int it = 0;; ++it) {
            if (*sq == seq) {
                seen = true;
                break;
            }
            if ((it & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) break;
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen) HIPCHK(h, hipStreamSynchronize(h->stream));
    *pose = h->h_result->pose;
    return check_map_errors(h, h->h_result->ctr);
}

// ------------------------------------------------------------------ VoxelMapManager surface
int lk_map_build(lk_handle* h, const float* xyz_world, const float* xyz_body, size_t n) {
    CHECK_H(h);
    if (n == 0) return LK_OK;
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "first-frame cloud exceeds max_scan_points");
    unsigned int ctr[LK_CTR_COUNT];
    HIPCHK(h, hipMemcpy(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost));
    if (ctr[LK_CTR_NODES] != 0) return fail(h, LK_ERR_STATE, "lk_map_build needs an empty map (BuildVoxelMap runs once)");
    h->grid_valid = false;
    float *d_w = nullptr, *d_b = nullptr;
    lk_pt_rec* d_bpts = nullptr;
    unsigned int *d_k0 = nullptr, *d_k1 = nullptr;
    int *d_i0 = nullptr, *d_i1 = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    DevTemps tmp;
    HIPCHK(h, tmp.alloc(&d_w, sizeof(float) * 3 * n));
    HIPCHK(h, tmp.alloc(&d_b, sizeof(float) * 3 * n));
    HIPCHK(h, tmp.alloc(&d_bpts, sizeof(lk_pt_rec) * n));
    HIPCHK(h, tmp.alloc(&d_k0, sizeof(unsigned int) * n));
    HIPCHK(h, tmp.alloc(&d_k1, sizeof(unsigned int) * n));
    HIPCHK(h, tmp.alloc(&d_i0, sizeof(int) * n));
    HIPCHK(h, tmp.alloc(&d_i1, sizeof(int) * n));
    HIPCHK(h, hipMemcpyAsync(d_w, xyz_world, sizeof(float) * 3 * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_b, xyz_body, sizeof(float) * 3 * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lk_bucket_begin_kernel, dim3(1), dim3(256), 0, h->stream, h->map);
    const int nb = (int)((n + 255) / 256);
    LAUNCH(h, "build_points", hipLaunchKernelGGL(lk_build_points_kernel, dim3(nb), dim3(256), 0, h->stream, h->map, h->pr,
                                                 h->d_filters, d_w, d_b, (int)n, d_bpts, d_k0, d_i0));
    // stable sort by root id: groups each root's points, preserving input order (voxel_map.cc:313-332)
    HIPCHK(h, lk_prim_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_i0, d_i1, n, 0, 32, h->stream));
    HIPCHK(h, tmp.alloc(&d_tmp, tmp_bytes));
    HIPCHK(h, lk_prim_sort_pairs(d_tmp, tmp_bytes, d_k0, d_k1, d_i0, d_i1, n, 0, 32, h->stream));
    LAUNCH(h, "build_segments",
           hipLaunchKernelGGL(lk_build_segments_kernel, dim3(nb), dim3(256), 0, h->stream, h->map, d_k1, (int)n));
    int grid = std::min(std::max((int)((n + 3) / 4), 1), 256);
    LAUNCH(h, "build_tree", hipLaunchKernelGGL(lk_build_tree_kernel, dim3(grid), dim3(LK_MB), 0, h->stream, h->map, h->pr,
                                               d_bpts, d_i1, d_i0));
    return check_map_errors(h);  // synchronises the stream; `tmp` frees the temporaries
}

int lk_map_update(lk_handle* h, const double* pw, const double* var9, size_t n) {
    CHECK_H(h);
    if (n == 0) return LK_OK;
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "n exceeds max_scan_points");
    h->grid_valid = false;
    std::vector<lk_pt_rec> st(n);
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) st[i].pw[c] = pw[3 * i + c];
        const double* v = var9 + 9 * i;
        st[i].var[0] = v[0], st[i].var[1] = v[1], st[i].var[2] = v[2], st[i].var[3] = v[4], st[i].var[4] = v[5], st[i].var[5] = v[8];
    }
    lk_pt_rec* d_pv = nullptr;
    DevTemps tmp;
    HIPCHK(h, tmp.alloc(&d_pv, sizeof(lk_pt_rec) * n));
    HIPCHK(h, hipMemcpyAsync(d_pv, st.data(), sizeof(lk_pt_rec) * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lk_bucket_begin_kernel, dim3(1), dim3(256), 0, h->stream, h->map);
    const int nb = (int)((n + 255) / 256);
    LAUNCH(h, "queue_pv", hipLaunchKernelGGL(lk_queue_pv_kernel, dim3(nb), dim3(256), 0, h->stream, h->map, h->pr, d_pv, (int)n));
    int grid = std::min(std::max((int)((n + 3) / 4), 1), 256);
    LAUNCH(h, "insert_pv_root", hipLaunchKernelGGL(lk_insert_root_kernel<true>, dim3(grid), dim3(LK_MB), 0, h->stream, h->map, h->pr,
                                                   h->d_filters, (const lk_point*)nullptr, d_pv, (int)n));
    LAUNCH(h, "insert_pv", hipLaunchKernelGGL(lk_insert_apply_kernel<true>, dim3(grid), dim3(LK_MB), 0, h->stream, h->map, h->pr,
                                              h->d_filters, (const lk_point*)nullptr, d_pv, (int)n));
    LAUNCH(h, "insert_pv_fallback", hipLaunchKernelGGL(lk_insert_fallback_kernel<true>, dim3(std::min(grid, 8)), dim3(LK_MB), 0, h->stream,
                                                       h->map, h->pr, h->d_filters, (const lk_point*)nullptr, d_pv, (int)n));
    return check_map_errors(h);  // synchronises the stream; `tmp` frees d_pv
}

static int upload_xyz_as_points(lk_handle* h, const float* xyz, size_t n) {
    std::vector<lk_point> p(n);
    for (size_t i = 0; i < n; ++i) p[i] = lk_point{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.f};
    HIPCHK(h, hipMemcpyAsync(h->d_scan, p.data(), sizeof(lk_point) * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // p goes out of scope
    return LK_OK;
}

int lk_residuals(lk_handle* h, const float* xyz_body, size_t n, double* h6, double* z, double* R, uint8_t* valid) {
    CHECK_H(h);
    if (n == 0) return LK_OK;
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "n exceeds max_scan_points");
    int rc = upload_xyz_as_points(h, xyz_body, n);
    if (rc) return rc;
    ResidualOut ro;
    ro.h6 = h->d_rows;
    ro.z = h->d_rows + 6 * n;
    ro.R = h->d_rows + 7 * n;
    ro.valid = h->d_valid;
    ro.world = nullptr;
    const int nblk = (int)((n + LK_RB - 1) / LK_RB);
    LAUNCH(h, "residual_rows",
           hipLaunchKernelGGL(lk_residual_kernel<true>, dim3(nblk, 1), dim3(LK_RB), 0, h->stream, h->map, h->pr, h->d_filters,
                              h->d_scan, (size_t)0, (int)n, h->d_partials, h->part_stride, ro, (size_t)0));
    HIPCHK(h, hipMemcpyAsync(h6, ro.h6, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(z, ro.z, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(R, ro.R, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(valid, ro.valid, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

// clearMemOutOfMap (voxel_map.cc:571-594) as a pool compaction; see lk_map_kernels.h
static int clear_outside(lk_handle* h, const LkSlideBox& box, uint32_t* n_removed) {
    if (n_removed) *n_removed = 0;
    h->grid_valid = false;
    unsigned int ctr[LK_CTR_COUNT];
    HIPCHK(h, hipMemcpyAsync(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const unsigned int n_nodes = std::min(ctr[LK_CTR_NODES], h->map.max_nodes), n_blocks = std::min(ctr[LK_CTR_BLOCKS], h->map.max_blocks),
                       n_roots = ctr[LK_CTR_ROOTS];
    if (n_roots == 0) return LK_OK;
    struct Scratch {  // freed on every exit path
        std::vector<void*> p;
        ~Scratch() {
            for (void* q : p) hipFree(q);
        }
        void* get(size_t bytes) {
            void* q = nullptr;
            if (hipMalloc(&q, bytes ? bytes : 16) != hipSuccess) return nullptr;
            p.push_back(q);
            return q;
        }
    } sc;
    unsigned int* alive_node = (unsigned int*)sc.get(sizeof(unsigned int) * n_nodes);
    unsigned int* alive_block = (unsigned int*)sc.get(sizeof(unsigned int) * std::max(n_blocks, 1u));
    unsigned int* new_node = (unsigned int*)sc.get(sizeof(unsigned int) * n_nodes);
    unsigned int* new_block = (unsigned int*)sc.get(sizeof(unsigned int) * std::max(n_blocks, 1u));
    lk_root_rec* kept = (lk_root_rec*)sc.get(sizeof(lk_root_rec) * n_roots);
    unsigned int* cnt = (unsigned int*)sc.get(sizeof(unsigned int) * 2);
    if (!alive_node || !alive_block || !new_node || !new_block || !kept || !cnt) return fail(h, LK_ERR_HIP, "map slide: out of device memory");
    HIPCHK(h, hipMemsetAsync(alive_node, 0, sizeof(unsigned int) * n_nodes, h->stream));
    HIPCHK(h, hipMemsetAsync(alive_block, 0, sizeof(unsigned int) * std::max(n_blocks, 1u), h->stream));
    HIPCHK(h, hipMemsetAsync(cnt, 0, sizeof(unsigned int) * 2, h->stream));
    hipLaunchKernelGGL(lk_slide_mark_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, h->stream, h->map, box, h->hash_cap,
                       alive_node, alive_block, kept, cnt);
    unsigned int hc[2];
    HIPCHK(h, hipMemcpyAsync(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (hc[0] + hc[1] != n_roots) return fail(h, LK_ERR_STATE, "hash table and root counter disagree");
    if (n_removed) *n_removed = hc[1];
    if (hc[1] == 0) return LK_OK;
    // dense new ids
    size_t t1 = 0, t2 = 0;
    HIPCHK(h, lk_prim_exclusive_scan(nullptr, t1, alive_node, new_node, n_nodes, h->stream));
    HIPCHK(h, lk_prim_exclusive_scan(nullptr, t2, alive_block, new_block, std::max(n_blocks, 1u), h->stream));
    void* tmp = sc.get(std::max(t1, t2));
    if (!tmp) return fail(h, LK_ERR_HIP, "map slide: out of device memory");
    HIPCHK(h, lk_prim_exclusive_scan(tmp, t1, alive_node, new_node, n_nodes, h->stream));
    if (n_blocks)
        HIPCHK(h, lk_prim_exclusive_scan(tmp, t2, alive_block, new_block, n_blocks, h->stream));
    unsigned int last[4] = {0, 0, 0, 0};  // new_node[n-1], alive_node[n-1], new_block[n-1], alive_block[n-1]
    HIPCHK(h, hipMemcpyAsync(&last[0], new_node + n_nodes - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&last[1], alive_node + n_nodes - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    if (n_blocks) {
        HIPCHK(h, hipMemcpyAsync(&last[2], new_block + n_blocks - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(&last[3], alive_block + n_blocks - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const unsigned int live_nodes = last[0] + last[1], live_blocks = last[2] + last[3];
    lk_node_rec* tn = (lk_node_rec*)sc.get(sizeof(lk_node_rec) * live_nodes);
    lk_plane_rec* tp = (lk_plane_rec*)sc.get(sizeof(lk_plane_rec) * live_nodes);
    lk_match_rec* tm = (lk_match_rec*)sc.get(sizeof(lk_match_rec) * live_nodes);
    lk_block_rec* tb = (lk_block_rec*)sc.get(sizeof(lk_block_rec) * std::max(live_blocks, 1u));
    if (!tn || !tp || !tm || !tb) return fail(h, LK_ERR_HIP, "map slide: out of device memory");
    hipLaunchKernelGGL(lk_slide_move_nodes_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, h->stream, h->map, n_nodes, alive_node,
                       new_node, new_block, tn, tp, tm);
    if (n_blocks)
        hipLaunchKernelGGL(lk_slide_move_blocks_kernel, dim3(n_blocks), dim3(64), 0, h->stream, h->map, alive_block, new_block, tb);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->map.nodes, tn, sizeof(lk_node_rec) * live_nodes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->map.planes, tp, sizeof(lk_plane_rec) * live_nodes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->map.match, tm, sizeof(lk_match_rec) * live_nodes, hipMemcpyDeviceToDevice, h->stream));
    if (live_blocks)
        HIPCHK(h, hipMemcpyAsync(h->map.blocks, tb, sizeof(lk_block_rec) * live_blocks, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(lk_slide_hash_clear_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, h->stream, h->map, h->hash_cap);
    if (hc[0]) {
        hipLaunchKernelGGL(lk_slide_remap_roots_kernel, dim3((hc[0] + 255) / 256), dim3(256), 0, h->stream, kept, hc[0], new_node);
        hipLaunchKernelGGL(lk_hash_insert_kernel, dim3((hc[0] + 255) / 256), dim3(256), 0, h->stream, h->map, h->pr, kept, (int)hc[0]);
    }
    hipLaunchKernelGGL(lk_slide_counters_kernel, dim3(1), dim3(1), 0, h->stream, h->map, live_nodes, live_blocks, hc[0]);
    HIPCHK(h, hipGetLastError());
    return check_map_errors(h);  // synchronises: the scratch buffers may be released after this
}

int lk_map_clear_outside(lk_handle* h, int32_t x_max, int32_t x_min, int32_t y_max, int32_t y_min, int32_t z_max, int32_t z_min,
                         uint32_t* n_removed) {
    CHECK_H(h);
    return clear_outside(h, LkSlideBox{x_max, x_min, y_max, y_min, z_max, z_min}, n_removed);
}

int lk_map_slide(lk_handle* h, const double* position, double sliding_thresh, int32_t half_map_size, int32_t* slid, uint32_t* n_removed) {
    CHECK_H(h);
    if (!position) return fail(h, LK_ERR_INVALID, "position is null");
    if (slid) *slid = 0;
    if (n_removed) *n_removed = 0;
    const double dx = position[0] - h->last_slide_position[0], dy = position[1] - h->last_slide_position[1],
                 dz = position[2] - h->last_slide_position[2];
    if (sqrt(dx * dx + dy * dy + dz * dz) < sliding_thresh) return LK_OK;  // voxel_map.cc:553
    for (int i = 0; i < 3; ++i) h->last_slide_position[i] = position[i];
    const double vs = h->cfg.max_voxel_size;  // the DOUBLE voxel size (voxel_map.cc:561)
    const int ix = (int)floor(position[0] / vs), iy = (int)floor(position[1] / vs), iz = (int)floor(position[2] / vs);
    if (slid) *slid = 1;
    return clear_outside(h, LkSlideBox{ix + half_map_size, ix - half_map_size, iy + half_map_size, iy - half_map_size, iz + half_map_size,
                                       iz - half_map_size}, n_removed);
}

int lk_map_slide_position(lk_handle* h, int32_t set, double* last3) {
    CHECK_H(h);
    if (!last3) return fail(h, LK_ERR_INVALID, "last3 is null");
    for (int i = 0; i < 3; ++i) {
        if (set) h->last_slide_position[i] = last3[i];
        else last3[i] = h->last_slide_position[i];
    }
    return LK_OK;
}

int lk_map_stats(lk_handle* h, uint32_t* n_roots, uint32_t* n_nodes, uint32_t* n_blocks) {
    CHECK_H(h);
    unsigned int ctr[LK_CTR_COUNT];
    HIPCHK(h, hipMemcpyAsync(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_roots) *n_roots = ctr[LK_CTR_ROOTS];
    if (n_nodes) *n_nodes = ctr[LK_CTR_NODES];
    if (n_blocks) *n_blocks = ctr[LK_CTR_BLOCKS];
    return LK_OK;
}

static void fill_header(lk_handle* h, lk_blob_header& hd, const unsigned int* ctr, uint32_t version) {
    memset(&hd, 0, sizeof(hd));
    hd.magic = LK_BLOB_MAGIC;
    hd.version = version;
    hd.n_roots = ctr[LK_CTR_ROOTS];
    hd.n_nodes = std::min(ctr[LK_CTR_NODES], h->map.max_nodes);
    hd.n_blocks = std::min(ctr[LK_CTR_BLOCKS], h->map.max_blocks);
    hd.block_pts = LK_BLOCK_PTS;
    hd.voxel_size = h->cfg.max_voxel_size;
    hd.max_layer = h->cfg.max_layer;
    hd.max_points_num = h->cfg.max_points_num;
}

// host blob of one LkMap (the handle's map, or one slot's overlay): header | roots (sorted by key) | nodes | planes | blocks
int export_map_blob(lk_handle* h, const LkMap& m, unsigned int hash_cap, void* blob, size_t* bytes) {
    if (!bytes) return fail(h, LK_ERR_INVALID, "bytes is null");
    unsigned int ctr[LK_CTR_COUNT];
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(ctr, m.counters, sizeof(ctr), hipMemcpyDeviceToHost));
    lk_blob_header hd;
    fill_header(h, hd, ctr, LK_ABI_VERSION);
    hd.n_nodes = std::min(ctr[LK_CTR_NODES], m.max_nodes);
    hd.n_blocks = std::min(ctr[LK_CTR_BLOCKS], m.max_blocks);
    size_t total = sizeof(hd) + (size_t)hd.n_roots * sizeof(lk_root_rec) + (size_t)hd.n_nodes * (sizeof(lk_node_rec) + sizeof(lk_plane_rec)) +
                   (size_t)hd.n_blocks * sizeof(lk_block_rec);
    hd.bytes = total;
    if (!blob) {
        *bytes = total;
        return LK_OK;
    }
    if (*bytes < total) return fail(h, LK_ERR_INVALID, "blob buffer too small");
    std::vector<int4> table(hash_cap);
    HIPCHK(h, hipMemcpy(table.data(), m.hash, sizeof(int4) * hash_cap, hipMemcpyDeviceToHost));
    std::vector<lk_root_rec> roots;
    roots.reserve(hd.n_roots);
    for (const int4& e : table)
        if (e.w >= 0) roots.push_back(lk_root_rec{{e.x, e.y, e.z}, e.w});
    if (roots.size() != hd.n_roots) return fail(h, LK_ERR_STATE, "hash table and root counter disagree");
    std::sort(roots.begin(), roots.end(), [](const lk_root_rec& a, const lk_root_rec& b) {
        return std::lexicographical_compare(a.key, a.key + 3, b.key, b.key + 3);
    });
    char* p = (char*)blob;
    memcpy(p, &hd, sizeof(hd));
    p += sizeof(hd);
    memcpy(p, roots.data(), roots.size() * sizeof(lk_root_rec));
    p += roots.size() * sizeof(lk_root_rec);
    HIPCHK(h, hipMemcpy(p, m.nodes, (size_t)hd.n_nodes * sizeof(lk_node_rec), hipMemcpyDeviceToHost));
    p += (size_t)hd.n_nodes * sizeof(lk_node_rec);
    HIPCHK(h, hipMemcpy(p, m.planes, (size_t)hd.n_nodes * sizeof(lk_plane_rec), hipMemcpyDeviceToHost));
    p += (size_t)hd.n_nodes * sizeof(lk_plane_rec);
    HIPCHK(h, hipMemcpy(p, m.blocks, (size_t)hd.n_blocks * sizeof(lk_block_rec), hipMemcpyDeviceToHost));
    *bytes = total;
    return LK_OK;
}
int lk_map_export(lk_handle* h, void* blob, size_t* bytes) {
    CHECK_H(h);
    return export_map_blob(h, h->map, h->hash_cap, blob, bytes);
}
// the voxels scan `slot` of the last overlay replay holds privately (the roots its inserts touched or created), as a map blob: the
// slot's key table is turned into the int4 form the exporter reads (entry index = root node id)


// A blob is only usable by a handle configured like the one that wrote it: the voxel size defines the keys, max_layer
// and max_points_num the insert state machine.
static int check_blob_config(lk_handle* h, const lk_blob_header& hd) {
    if (hd.voxel_size != h->cfg.max_voxel_size || hd.max_layer != h->cfg.max_layer || hd.max_points_num != h->cfg.max_points_num) {
        char buf[200];
        snprintf(buf, sizeof(buf), "blob was written with voxel_size %.9g / max_layer %d / max_points_num %d, the handle has %.9g / %d / %d",
                 hd.voxel_size, hd.max_layer, hd.max_points_num, h->cfg.max_voxel_size, h->cfg.max_layer, h->cfg.max_points_num);
        return fail(h, LK_ERR_INVALID, buf);
    }
    return LK_OK;
}

// child / block / root-node ids of an imported map must index the imported pools (device-resident blobs are checked
// where they are; a host blob is checked on the host before anything is copied)
__global__ void lk_validate_ids_kernel(LkMap map, unsigned int n_nodes, unsigned int n_blocks, unsigned int hash_cap) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (i < n_nodes) {
        const lk_node_rec* nd = &map.nodes[i];
#pragma unroll
        for (int c = 0; c < 8; ++c) bad |= nd->child[c] < -1 || nd->child[c] >= (int)n_nodes;
        bad |= nd->block < -1 || nd->block >= (int)n_blocks;
        bad |= nd->npts < 0 || (nd->block >= 0 && nd->npts > LK_BLOCK_PTS);
    }
    if (i < hash_cap) {
        const int4 e = map.hash[i];
        bad |= e.w >= (int)n_nodes || e.w < LK_LOCKED;
    }
    if (bad) atomicOr(&map.counters[LK_CTR_ERR], LK_E_BAD_BLOB);
}

static int reset_pools(lk_handle* h) {
    h->grid_valid = false;
    unsigned int ninit = std::max(h->hash_cap, h->map.max_nodes);
    hipLaunchKernelGGL(lk_pool_init_kernel, dim3((ninit + 255) / 256), dim3(256), 0, h->stream, h->map, h->hash_cap);
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}

int lk_map_import(lk_handle* h, const void* blob, size_t bytes) {
    CHECK_H(h);
    if (!blob || bytes < sizeof(lk_blob_header)) return fail(h, LK_ERR_INVALID, "blob too small");
    lk_blob_header hd;
    memcpy(&hd, blob, sizeof(hd));
    if (hd.magic != LK_BLOB_MAGIC || hd.version != LK_ABI_VERSION || hd.block_pts != LK_BLOCK_PTS || hd.bytes > bytes)
        return fail(h, LK_ERR_INVALID, "bad blob header");
    {   // the counts must account for every byte: a truncated or corrupt blob must not make the copies read past `blob`
        const size_t expect = sizeof(hd) + (size_t)hd.n_roots * sizeof(lk_root_rec) +
                              (size_t)hd.n_nodes * (sizeof(lk_node_rec) + sizeof(lk_plane_rec)) + (size_t)hd.n_blocks * sizeof(lk_block_rec);
        if (expect != hd.bytes) return fail(h, LK_ERR_INVALID, "blob size does not match its record counts (truncated or corrupt)");
    }
    int rc = check_blob_config(h, hd);
    if (rc) return rc;
    if (hd.n_nodes > h->map.max_nodes || hd.n_blocks > h->map.max_blocks || 2 * (size_t)hd.n_roots > h->hash_cap)
        return fail(h, LK_ERR_CAPACITY, "blob exceeds pool capacities");
    {   // ids are range-checked here, on the host copy, before the device sees them
        const char* q = (const char*)blob + sizeof(hd);
        const lk_root_rec* rr = (const lk_root_rec*)q;
        for (uint32_t i = 0; i < hd.n_roots; ++i)
            if (rr[i].node < 0 || (uint32_t)rr[i].node >= hd.n_nodes) return fail(h, LK_ERR_INVALID, "blob: root node id out of range");
        const lk_node_rec* nn = (const lk_node_rec*)(q + (size_t)hd.n_roots * sizeof(lk_root_rec));
        for (uint32_t i = 0; i < hd.n_nodes; ++i) {
            lk_node_rec nd;
            memcpy(&nd, &nn[i], sizeof(nd));
            for (int c = 0; c < 8; ++c)
                if (nd.child[c] < -1 || nd.child[c] >= (int)hd.n_nodes) return fail(h, LK_ERR_INVALID, "blob: child id out of range");
            if (nd.block < -1 || nd.block >= (int)hd.n_blocks || nd.npts < 0 || (nd.block >= 0 && nd.npts > LK_BLOCK_PTS))
                return fail(h, LK_ERR_INVALID, "blob: point block id / count out of range");
        }
    }
    rc = reset_pools(h);
    if (rc) return rc;
    const char* p = (const char*)blob + sizeof(hd);
    lk_root_rec* d_roots = nullptr;
    DevTemps tmp;
    if (hd.n_roots) {
        HIPCHK(h, tmp.alloc(&d_roots, sizeof(lk_root_rec) * hd.n_roots));
        HIPCHK(h, hipMemcpyAsync(d_roots, p, sizeof(lk_root_rec) * hd.n_roots, hipMemcpyHostToDevice, h->stream));
    }
    p += (size_t)hd.n_roots * sizeof(lk_root_rec);
    HIPCHK(h, hipMemcpyAsync(h->map.nodes, p, (size_t)hd.n_nodes * sizeof(lk_node_rec), hipMemcpyHostToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_node_rec);
    HIPCHK(h, hipMemcpyAsync(h->map.planes, p, (size_t)hd.n_nodes * sizeof(lk_plane_rec), hipMemcpyHostToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_plane_rec);
    HIPCHK(h, hipMemcpyAsync(h->map.blocks, p, (size_t)hd.n_blocks * sizeof(lk_block_rec), hipMemcpyHostToDevice, h->stream));
    if (hd.n_roots)
        hipLaunchKernelGGL(lk_hash_insert_kernel, dim3((hd.n_roots + 255) / 256), dim3(256), 0, h->stream, h->map, h->pr,
                           d_roots, (int)hd.n_roots);
    if (hd.n_nodes)
        hipLaunchKernelGGL(lk_derive_match_kernel, dim3((hd.n_nodes + 255) / 256), dim3(256), 0, h->stream, h->map, (int)hd.n_nodes);
    unsigned int ctr[LK_CTR_COUNT] = {0};
    ctr[LK_CTR_NODES] = hd.n_nodes, ctr[LK_CTR_BLOCKS] = hd.n_blocks, ctr[LK_CTR_ROOTS] = hd.n_roots;
    // only the first three counters: the hash-insert kernel may raise the error word concurrently
    HIPCHK(h, hipMemcpyAsync(h->map.counters, ctr, 3 * sizeof(unsigned int), hipMemcpyHostToDevice, h->stream));
    return check_map_errors(h);
}

// Device-resident blob for the RCCL broadcast: header | hash table (full) | nodes | planes | blocks (used prefixes).
// version carries bit 0x100 to distinguish it from the host blob.
int lk_map_export_dev(lk_handle* h, void* d_blob, size_t* bytes) {
    CHECK_H(h);
    if (!bytes) return fail(h, LK_ERR_INVALID, "bytes is null");
    unsigned int ctr[LK_CTR_COUNT];
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost));
    lk_blob_header hd;
    fill_header(h, hd, ctr, LK_ABI_VERSION | 0x100u);
    size_t total = sizeof(hd) + sizeof(int4) * (size_t)h->hash_cap + (size_t)hd.n_nodes * (sizeof(lk_node_rec) + sizeof(lk_plane_rec)) +
                   (size_t)hd.n_blocks * sizeof(lk_block_rec);
    total = (total + 255) & ~(size_t)255;
    hd.bytes = total;
    if (!d_blob) {
        *bytes = total;
        return LK_OK;
    }
    if (*bytes < total) return fail(h, LK_ERR_INVALID, "device blob buffer too small");
    char* p = (char*)d_blob;
    HIPCHK(h, hipMemcpyAsync(p, &hd, sizeof(hd), hipMemcpyHostToDevice, h->stream));
    p += sizeof(hd);
    HIPCHK(h, hipMemcpyAsync(p, h->map.hash, sizeof(int4) * (size_t)h->hash_cap, hipMemcpyDeviceToDevice, h->stream));
    p += sizeof(int4) * (size_t)h->hash_cap;
    HIPCHK(h, hipMemcpyAsync(p, h->map.nodes, (size_t)hd.n_nodes * sizeof(lk_node_rec), hipMemcpyDeviceToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_node_rec);
    HIPCHK(h, hipMemcpyAsync(p, h->map.planes, (size_t)hd.n_nodes * sizeof(lk_plane_rec), hipMemcpyDeviceToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_plane_rec);
    HIPCHK(h, hipMemcpyAsync(p, h->map.blocks, (size_t)hd.n_blocks * sizeof(lk_block_rec), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *bytes = total;
    return LK_OK;
}

int lk_map_import_dev(lk_handle* h, const void* d_blob, size_t bytes) {
    CHECK_H(h);
    if (!d_blob || bytes < sizeof(lk_blob_header)) return fail(h, LK_ERR_INVALID, "device blob too small");
    lk_blob_header hd;
    HIPCHK(h, hipMemcpy(&hd, d_blob, sizeof(hd), hipMemcpyDeviceToHost));
    if (hd.magic != LK_BLOB_MAGIC || hd.version != (LK_ABI_VERSION | 0x100u) || hd.bytes > bytes)
        return fail(h, LK_ERR_INVALID, "bad device blob header");
    int rc = check_blob_config(h, hd);
    if (rc) return rc;
    if (hd.n_nodes > h->map.max_nodes || hd.n_blocks > h->map.max_blocks)
        return fail(h, LK_ERR_CAPACITY, "device blob exceeds pool capacities");
    size_t expect = sizeof(hd) + sizeof(int4) * (size_t)h->hash_cap + (size_t)hd.n_nodes * (sizeof(lk_node_rec) + sizeof(lk_plane_rec)) +
                    (size_t)hd.n_blocks * sizeof(lk_block_rec);
    if (((expect + 255) & ~(size_t)255) != hd.bytes)
        return fail(h, LK_ERR_INVALID, "device blob was exported with different capacities (max_roots must match)");
    rc = reset_pools(h);
    if (rc) return rc;
    const char* p = (const char*)d_blob + sizeof(hd);
    HIPCHK(h, hipMemcpyAsync(h->map.hash, p, sizeof(int4) * (size_t)h->hash_cap, hipMemcpyDeviceToDevice, h->stream));
    p += sizeof(int4) * (size_t)h->hash_cap;
    HIPCHK(h, hipMemcpyAsync(h->map.nodes, p, (size_t)hd.n_nodes * sizeof(lk_node_rec), hipMemcpyDeviceToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_node_rec);
    HIPCHK(h, hipMemcpyAsync(h->map.planes, p, (size_t)hd.n_nodes * sizeof(lk_plane_rec), hipMemcpyDeviceToDevice, h->stream));
    p += (size_t)hd.n_nodes * sizeof(lk_plane_rec);
    HIPCHK(h, hipMemcpyAsync(h->map.blocks, p, (size_t)hd.n_blocks * sizeof(lk_block_rec), hipMemcpyDeviceToDevice, h->stream));
    if (hd.n_nodes)
        hipLaunchKernelGGL(lk_derive_match_kernel, dim3((hd.n_nodes + 255) / 256), dim3(256), 0, h->stream, h->map, (int)hd.n_nodes);
    unsigned int ctr[LK_CTR_COUNT] = {0};
    ctr[LK_CTR_NODES] = hd.n_nodes, ctr[LK_CTR_BLOCKS] = hd.n_blocks, ctr[LK_CTR_ROOTS] = hd.n_roots;
    HIPCHK(h, hipMemcpyAsync(h->map.counters, ctr, sizeof(ctr), hipMemcpyHostToDevice, h->stream));
    {
        const unsigned int nv = std::max(hd.n_nodes, h->hash_cap);
        hipLaunchKernelGGL(lk_validate_ids_kernel, dim3((nv + 255) / 256), dim3(256), 0, h->stream, h->map, hd.n_nodes, hd.n_blocks, h->hash_cap);
        HIPCHK(h, hipGetLastError());
    }
    rc = check_map_errors(h);
    if (rc) {  // a map with dangling ids must not stay loaded
        reset_pools(h);
        hipStreamSynchronize(h->stream);
        return fail(h, LK_ERR_INVALID, "device blob: node / block ids out of range");
    }
    return LK_OK;
}

// ------------------------------------------------------------------ KILO path
int lk_update_points(lk_handle* h, double t, const float* xyz_body, size_t n, float* xyz_world_out, float* intensity_out,
                     size_t* n_effect) {
    CHECK_H(h);
    if (n == 0) return fail(h, LK_ERR_INVALID, "empty bucket");
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "n exceeds max_scan_points");
    int rc = upload_xyz_as_points(h, xyz_body, n);
    if (rc) return rc;
    rc = enqueue_bucket(h, h->d_scan, (int)n, t, h->d_world, true);
    if (rc) return rc;
    if ((rc = spec_join(h))) return rc;
    std::vector<float> w(4 * n);
    int lastN = 0;
    HIPCHK(h, hipMemcpyAsync(w.data(), h->d_world, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&lastN, &h->d_filters[0].last_N, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    rc = check_map_errors(h);  // synchronises
    if (rc) return rc;
    for (size_t i = 0; i < n; ++i) {
        if (xyz_world_out)
            for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
        if (intensity_out) intensity_out[i] = w[4 * i + 3];
    }
    if (n_effect) *n_effect += (size_t)lastN;
    return LK_OK;
}

static void imu_noise(const lk_config& c, double* Rn) {
    Rn[0] = Rn[1] = c.imu_acc_meas_noise;
    Rn[2] = c.imu_acc_z_meas_noise;
    Rn[3] = Rn[4] = Rn[5] = c.imu_gyr_meas_noise;
}
static int enqueue_imu(lk_handle* h, const lk_imu* imu) {
    LkImuArgs a;
    a.t = imu->stamp;
    for (int i = 0; i < 3; ++i) a.acc[i] = imu->acc[i], a.gyr[i] = imu->gyr[i];
    a.acc_scale = h->cfg.gravity / h->acc_norm;
    imu_noise(h->cfg, a.Rn);
    LAUNCH(h, "imu", hipLaunchKernelGGL(lk_imu_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters, h->d_Q, a));
    return LK_OK;
}
static int enqueue_kin(lk_handle* h, const lk_kin_imu* kin) {
    LkKinArgs a;
    a.k = *kin;
    a.acc_scale = h->cfg.gravity / h->acc_norm;
    imu_noise(h->cfg, a.Rn);
    a.kin_noise = h->cfg.kin_meas_noise;
    LAUNCH(h, "kin", hipLaunchKernelGGL(lk_kin_kernel, dim3(1), dim3(LK_FB), 0, h->stream, h->d_filters, h->d_Q, a));
    return LK_OK;
}
int lk_update_imu(lk_handle* h, const lk_imu* imu) {
    CHECK_H(h);
    int rc = enqueue_imu(h, imu);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_update_kin_imu(lk_handle* h, const lk_kin_imu* kin) {
    CHECK_H(h);
    int rc = enqueue_kin(h, kin);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

// bucket loop of KILO::process (KILO.cc:375-395); pts = host copy of the sorted cloud (bucket bounds, IMU interleave),
// d_pts = the same cloud in HBM
static int run_scan(lk_handle* h, const lk_point* pts, const lk_point* d_pts, size_t n, double t_begin, const lk_imu* imus,
                    size_t n_imu, const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out);

int lk_process_scan(lk_handle* h, const lk_point* pts, size_t n, double t_begin, const lk_imu* imus, size_t n_imu,
                    const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out) {
    CHECK_H(h);
    if (n == 0) return fail(h, LK_ERR_INVALID, "empty scan");
    if (n_imu && n_kin) return fail(h, LK_ERR_INVALID, "pass either IMU or kin+IMU messages, not both");
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "scan exceeds max_scan_points");
    HIPCHK(h, hipMemcpyAsync(h->d_scan, pts, sizeof(lk_point) * n, hipMemcpyHostToDevice, h->stream));
    return run_scan(h, pts, h->d_scan, n, t_begin, imus, n_imu, kins, n_kin, xyz_world_out, out);
}

// staging buffer of the ragged / resident tables (device copy + pinned host copy, grow-only); synchronises the stream: a previous
// call's upload from the staging buffer must have completed before it is overwritten
static int rag_reserve(lk_handle* h, size_t bytes) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (bytes <= h->rag_cap) return LK_OK;
    if (h->d_rag) hipFree(h->d_rag), h->d_rag = nullptr;
    if (h->h_rag) hipHostFree(h->h_rag), h->h_rag = nullptr;
    h->rag_cap = 0;
    HIPCHK(h, hipMalloc(&h->d_rag, bytes + bytes / 2));
    HIPCHK(h, hipHostMalloc(&h->h_rag, bytes + bytes / 2, hipHostMallocDefault));
    h->rag_cap = bytes + bytes / 2;
    return LK_OK;
}

// filters[0] before a scan that works on it in place (grid-resident kernel, pipelined launches): check_map_errors restores it on LK_ERR_TIMEOUT
static int backup_filter(lk_handle* h) {
    if (!h->d_fbackup) HIPCHK(h, hipMalloc(&h->d_fbackup, sizeof(LkFilter)));
    HIPCHK(h, hipMemcpyAsync(h->d_fbackup, h->d_filters, sizeof(LkFilter), hipMemcpyDeviceToDevice, h->stream));
    h->fbackup_valid = true;
    return LK_OK;
}
// The bucket loop of KILO::process for a scan of small buckets as ONE launch (lk_scan_stream_kernel).  bstart[k] / btime[k]: first
// point and absolute time of bucket k (nb buckets, bstart[nb] = n); the messages are the scan's lk_imu or lk_kin_imu records.
// The scan's result comes back through finish_scan; a launch that stopped at fallback items is followed by lk_resident_fallback_kernel and
// another launch from where it stopped, until the scan is through (LkResume).
static int run_scan_resident(lk_handle* h, const lk_point* d_pts, const std::vector<unsigned long long>& bstart, const std::vector<double>& btime,
                             const void* msgs, size_t n_msg, int msg_kind, float* d_world, lk_pose* pose) {
    const size_t nb = btime.size();
    const size_t msg_bytes = msg_kind == 2 ? sizeof(lk_kin_imu) : sizeof(lk_imu);
    const size_t o_po = 0, o_t = o_po + 8 * (nb + 1), o_im = o_t + 8 * nb, o_nb = o_im + msg_bytes * n_msg, o_io = o_nb + 8, o_rs = o_io + 8,
                 bytes = o_rs + sizeof(LkResume);
    int rc = rag_reserve(h, bytes);
    if (rc) return rc;
    unsigned char* stage = static_cast<unsigned char*>(h->h_rag);
    memcpy(stage + o_po, bstart.data(), 8 * (nb + 1));
    memcpy(stage + o_t, btime.data(), 8 * nb);
    if (n_msg) memcpy(stage + o_im, msgs, msg_bytes * n_msg);
    const unsigned int nbu[2] = {(unsigned int)nb, 0u}, io[2] = {0u, (unsigned int)n_msg};
    memcpy(stage + o_nb, nbu, 8);
    memcpy(stage + o_io, io, 8);
    {
        LkResume r0;
        memset(&r0, 0, sizeof(r0));
        r0.fb_bucket = -1;
        memcpy(stage + o_rs, &r0, sizeof(r0));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_rag, stage, bytes, hipMemcpyHostToDevice, h->stream));
    unsigned char* dr = static_cast<unsigned char*>(h->d_rag);
    LkResume* d_rs = reinterpret_cast<LkResume*>(dr + o_rs);
    LkRagged rg;
    rg.pt_off = reinterpret_cast<const unsigned long long*>(dr + o_po);
    rg.t = reinterpret_cast<const double*>(dr + o_t);
    rg.nb = reinterpret_cast<const unsigned int*>(dr + o_nb);
    rg.ldb = (int)nb;
    rg.bstart = nullptr;
    rg.imu_off = reinterpret_cast<const unsigned int*>(dr + o_io);
    rg.imu = reinterpret_cast<const double*>(dr + o_im);
    rg.msg_stride = (int)(msg_bytes / sizeof(double));
    rg.kin_noise = h->cfg.kin_meas_noise;
    rg.q_diag = h->q_diag ? 1 : 0;
    rg.acc_scale = h->cfg.gravity / h->acc_norm;
    imu_noise(h->cfg, rg.Rn);
    h->grid_valid = false;   // the map changes
    static const bool xid_en = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const bool xid = h->pr.ext_identity && xid_en;
    if (h->epoch + (unsigned int)nb + 16u < h->epoch || h->epoch >= 0xf0000000u) {   // stamps are plain unsigned numbers: start over long before they wrap
        HIPCHK(h, hipMemsetAsync(h->map.dirty, 0, sizeof(unsigned int) * (size_t)h->map.max_nodes, h->stream));
        HIPCHK(h, hipMemsetAsync(h->map.newroot, 0, sizeof(unsigned int) * (size_t)(LK_NEWROOT_MASK + 1), h->stream));
        h->epoch = 16, h->spec_base = 17;
    }
    const unsigned int epoch0 = h->epoch + 1u;
    h->epoch += (unsigned int)nb;
    h->spec_base = h->epoch + 1u;
    static const unsigned int timeout_ms = getenv("LEGKILO_RESIDENT_TIMEOUT_MS") ? (unsigned int)std::max(1, atoi(getenv("LEGKILO_RESIDENT_TIMEOUT_MS"))) : LK_RESIDENT_TIMEOUT_MS;
    void (*k)(LkMap, LkParams, LkFilter*, const lk_point*, LkRagged, const double*, LkFilter*, float*, int2*, unsigned int, unsigned int, LkResume*) =
        msg_kind == 2 ? (xid ? lk_scan_stream_kernel<2, true> : lk_scan_stream_kernel<2, false>)
      : msg_kind == 1 ? (xid ? lk_scan_stream_kernel<1, true> : lk_scan_stream_kernel<1, false>)
                      : (xid ? lk_scan_stream_kernel<0, true> : lk_scan_stream_kernel<0, false>);
    // a launch given up keeps the filter in LDS and returns before its write-back - but a scan that is picked up again after fallback items
    // has written it once: every scan starts with a copy, which a LK_ERR_TIMEOUT in any of its launches puts back (check_map_errors)
    if ((rc = backup_filter(h))) return rc;
    h->resident_scans += 1;
    for (size_t round = 0;; ++round) {
        h->fbackup_valid = true;   // (finish_scan's error check ends the previous launch's claim on the copy; it is still the pre-scan state)
    LAUNCH(h, "scan_stream", hipLaunchKernelGGL(k, dim3(1), dim3((1 + LK_INS_WAVES) * LK_WAVE), 0, h->stream, h->map, h->pr, h->d_filters, d_pts, rg, h->d_Q, h->d_snap, d_world,
                                                h->d_ids, epoch0, h->test_stall_ms ? (h->test_stall_ms | 0x80000000u) : timeout_ms, d_rs));
#ifdef LK_DEBUG_RES
    {
        unsigned long long hb[32];
        hipStreamSynchronize(h->stream);
        hipMemcpyFromSymbol(hb, HIP_SYMBOL(lk_res_dbg), sizeof(hb));
        const double nbk = (double)hb[31];
        const char* fn[8] = {"predict", "tiles", "wait-decided", "suspects(+redo)", "update", "snapshot+post", "messages", "wait-done(b-2)"};
        const char* in[8] = {"wait-post", "begin", "reproject", "stamp", "root", "apply", "fallback+done", "-"};
        fprintf(stderr, "[resident] %.0f buckets; filter wave (us per bucket):", nbk);
        for (int q = 0; q < 8; ++q) fprintf(stderr, " %s %.2f;", fn[q], (double)hb[q] / nbk * 0.01);
        fprintf(stderr, "\n[resident] insert wave 1:");
        for (int q = 0; q < 7; ++q) fprintf(stderr, " %s %.2f;", in[q], (double)hb[8 + q] / nbk * 0.01);
        fprintf(stderr, " buckets with fallback items %llu\n", hb[15]);
        {
            unsigned long long cd[16];
            hipMemcpyFromSymbol(cd, HIP_SYMBOL(lk_core_dbg), sizeof(cd));
            const double np_ = (double)(cd[7] ? cd[7] : 1), nu_ = (double)(cd[15] ? cd[15] : 1);
            fprintf(stderr, "[resident] predict core (%llu calls), us: rotations %.2f; rows of Fx P %.2f; columns %.2f; Q + state %.2f\n", cd[7],
                    cd[0] / np_ * 0.01, cd[1] / np_ * 0.01, cd[2] / np_ * 0.01, cd[3] / np_ * 0.01);
            fprintf(stderr, "[resident] update core (%llu calls), us: columns %.2f; Gauss-Jordan %.2f; X + dx %.2f; P update %.2f; rotation(s) + state %.2f\n", cd[15],
                    cd[8] / nu_ * 0.01, cd[9] / nu_ * 0.01, cd[10] / nu_ * 0.01, cd[11] / nu_ * 0.01, cd[12] / nu_ * 0.01);
            memset(cd, 0, sizeof(cd));
            hipMemcpyToSymbol(HIP_SYMBOL(lk_core_dbg), cd, sizeof(cd));
        }
        {
            static unsigned long long ts[6][1024];
            hipMemcpyFromSymbol(ts, HIP_SYMBOL(lk_res_ts), sizeof(ts));
            const int nbq = (int)std::min<size_t>(nb, 1024);
            double a01 = 0, a12 = 0, a24 = 0, a34 = 0, a25 = 0, a00 = 0;
            int c = 0;
            for (int b = 2; b + 2 < nbq; ++b, ++c) {
                a01 += (double)(long long)(ts[1][b] - ts[0][b]), a12 += (double)(long long)(ts[2][b] - ts[1][b]), a24 += (double)(long long)(ts[4][b] - ts[2][b]);
                a34 += (double)(long long)(ts[4][b] - ts[3][b]), a25 += (double)(long long)(ts[5][b] - ts[2][b]), a00 += (double)(long long)(ts[0][b + 1] - ts[0][b]);
            }
            if (c) fprintf(stderr, "[resident] hand-offs (us, mean over %d buckets): post -> insert sees it %.2f; -> decided posted %.2f; -> filter sees it %.2f (filter had waited %.2f); decided -> done %.2f; post to post %.2f\n",
                           c, a01 / c * 0.01, a12 / c * 0.01, a24 / c * 0.01, a34 / c * 0.01, a25 / c * 0.01, a00 / c * 0.01);
            for (int b = 100; b < 104 && b + 1 < nbq; ++b)
                fprintf(stderr, "[resident]   bucket %d: post 0, seen %+.2f, decided %+.2f, filter waits from %+.2f, sees %+.2f, done %+.2f, next post %+.2f\n", b,
                        (double)(long long)(ts[1][b] - ts[0][b]) * 0.01, (double)(long long)(ts[2][b] - ts[0][b]) * 0.01, (double)(long long)(ts[3][b] - ts[0][b]) * 0.01,
                        (double)(long long)(ts[4][b] - ts[0][b]) * 0.01, (double)(long long)(ts[5][b] - ts[0][b]) * 0.01, (double)(long long)(ts[0][b + 1] - ts[0][b]) * 0.01);
        }
        memset(hb, 0, sizeof(hb));
        hipMemcpyToSymbol(HIP_SYMBOL(lk_res_dbg), hb, sizeof(hb));
    }
#endif
        if ((rc = finish_scan(h, pose, d_rs))) return rc;
        const int* rsm = h->h_result->resume;   // { filter wave's next bucket, insert team's next bucket, bucket with fallback items pending }
        if (rsm[2] < 0) {
            if (rsm[0] >= (int)nb && rsm[1] >= (int)nb) break;   // both roles are through
            return fail(h, LK_ERR_STATE, "the scan-resident kernel stopped without a reason: filter wave at bucket " + std::to_string(rsm[0]) + ", insert team at " + std::to_string(rsm[1]) + " of " + std::to_string(nb));
        }
        if (round > nb + 4) return fail(h, LK_ERR_STATE, "the scan-resident kernel does not advance");
        h->resident_relaunches += 1;
        LAUNCH(h, "resident_fallback", hipLaunchKernelGGL(lk_resident_fallback_kernel, dim3(1), dim3(LK_MB), 0, h->stream, h->map, h->pr, h->d_snap, d_pts, rg, epoch0, 1, d_rs, (unsigned int*)nullptr));
        if (rsm[0] >= (int)nb && rsm[1] >= (int)nb) {   // they were the last bucket's: nothing to pick up
            if ((rc = finish_scan(h, pose, nullptr))) return rc;
            break;
        }
    }
    return LK_OK;
}
// The bucket loop of a scan of LARGE buckets as one grid-resident launch (lk_scan_grid_kernel); same table layout as run_scan_resident.
static int run_scan_grid(lk_handle* h, const lk_point* d_pts, const std::vector<unsigned long long>& bstart, const std::vector<double>& btime,
                         size_t biggest, float* d_world, lk_pose* pose) {
    const size_t nb = btime.size();
    const size_t o_po = 0, o_t = o_po + 8 * (nb + 1), o_nb = o_t + 8 * nb, o_io = o_nb + 8, o_sync = o_io + 8, o_rs = o_sync + 16, bytes = o_rs + sizeof(LkResume);
    int rc = rag_reserve(h, bytes);
    if (rc) return rc;
    unsigned char* stage = static_cast<unsigned char*>(h->h_rag);
    memcpy(stage + o_po, bstart.data(), 8 * (nb + 1));
    memcpy(stage + o_t, btime.data(), 8 * nb);
    const unsigned int nbu[2] = {(unsigned int)nb, 0u}, io[2] = {0u, 0u}, zero4[4] = {0u, 0u, 0u, 0u};
    memcpy(stage + o_nb, nbu, 8);
    memcpy(stage + o_io, io, 8);
    memcpy(stage + o_sync, zero4, 16);
    {
        LkResume r0;
        memset(&r0, 0, sizeof(r0));
        r0.fb_bucket = -1;
        memcpy(stage + o_rs, &r0, sizeof(r0));
    }
    HIPCHK(h, hipMemcpyAsync(h->d_rag, stage, bytes, hipMemcpyHostToDevice, h->stream));
    if ((rc = backup_filter(h))) return rc;
    unsigned char* dr = static_cast<unsigned char*>(h->d_rag);
    LkRagged rg;
    memset(&rg, 0, sizeof(rg));
    rg.pt_off = reinterpret_cast<const unsigned long long*>(dr + o_po);
    rg.t = reinterpret_cast<const double*>(dr + o_t);
    rg.nb = reinterpret_cast<const unsigned int*>(dr + o_nb);
    rg.ldb = (int)nb;
    rg.bstart = nullptr;
    rg.imu_off = reinterpret_cast<const unsigned int*>(dr + o_io);
    rg.q_diag = h->q_diag ? 1 : 0;
    h->grid_valid = false;   // the map changes
    static const bool xid_en = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const bool xid = h->pr.ext_identity && xid_en;
    static const unsigned int timeout_ms = getenv("LEGKILO_RESIDENT_TIMEOUT_MS") ? (unsigned int)std::max(1, atoi(getenv("LEGKILO_RESIDENT_TIMEOUT_MS"))) : LK_RESIDENT_TIMEOUT_MS;
    static const int wg_env = getenv("LEGKILO_GRIDSCAN_WG") ? atoi(getenv("LEGKILO_GRIDSCAN_WG")) : 0;
    const int tiles = (int)((biggest + LK_WAVE - 1) / LK_WAVE);
    int G = wg_env > 0 ? wg_env : std::max(16, (tiles + 3) / 4 + 12);  // a wave per tile of the largest bucket and some more for the per-root passes; every
                                                                      // further workgroup makes each barrier dearer (51 x 1 960 points, round 5: 8 workgroups 2.67 ms, 12: 2.37, 16: 2.30, 24: 2.29, 32: 2.31; round 4: 128: 2.81)
    G = std::min(G, LK_GRIDSCAN_WG_MAX);                             // 128 workgroups of 4 waves are resident on 256 CUs whatever else is true
    {   // a partitioned / smaller device (CPX: 32 CUs): never more spinning workgroups than can be resident at once
        static int resident_max = -1;
        if (resident_max < 0) {
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lk_scan_grid_kernel<true>, LK_FB, 0) == hipSuccess && per_cu > 0)
                resident_max = per_cu * prop.multiProcessorCount;
            else
                resident_max = LK_GRIDSCAN_WG_MAX;
            (void)hipGetLastError();
        }
        G = std::max(1, std::min(G, resident_max));
    }
    // up to one workgroup per CU of an XCD: launch 8 G blocks and let only every eighth work (LEGKILO_GRIDSCAN_XCD=0: all G blocks, any XCD)
    static const bool one_xcd_en = getenv("LEGKILO_GRIDSCAN_XCD") == nullptr || atoi(getenv("LEGKILO_GRIDSCAN_XCD")) != 0;
    const int stride = (one_xcd_en && G <= 32) ? 8 : 1;
    const auto k = xid ? lk_scan_grid_kernel<true> : lk_scan_grid_kernel<false>;
    LkResume* d_rs = reinterpret_cast<LkResume*>(dr + o_rs);
    h->grid_scans += 1;
    int b0 = 0;
    for (size_t round = 0;; ++round) {
        h->fbackup_valid = true;   // the copy taken above is the pre-scan state for every launch of this scan
        LAUNCH(h, "scan_grid", hipLaunchKernelGGL(k, dim3(G * stride), dim3(LK_FB), 0, h->stream, h->map, h->pr, h->d_filters, d_pts, rg, h->d_Q, h->d_snap, d_world,
                                                  h->d_partials, reinterpret_cast<unsigned int*>(dr + o_sync), h->test_stall_ms ? (h->test_stall_ms | 0x80000000u) : timeout_ms, stride, b0, d_rs));
        if ((rc = finish_scan(h, pose, d_rs))) return rc;
#ifdef LK_DEBUG_RES
        {
            unsigned long long hb[32];
            hipMemcpyFromSymbol(hb, HIP_SYMBOL(lk_res_dbg), sizeof(hb));
            const char* nm[10] = {"residual tiles", "barrier", "update+snapshot+begin (wg 0)", "barrier", "predict (wg 0; others re-project)", "barrier", "root pass (wg 0's share)", "barrier",
                                  "emitted groups (wg 0's share)", "barrier"};
            fprintf(stderr, "[grid] %zu buckets, G = %d; workgroup 0, us per bucket:", nb, G);
            for (int q = 0; q < 10; ++q) fprintf(stderr, " %s %.2f;", nm[q], (double)hb[16 + q] / (double)nb * 0.01);
            fprintf(stderr, "\n");
            memset(hb, 0, sizeof(hb));
            hipMemcpyToSymbol(HIP_SYMBOL(lk_res_dbg), hb, sizeof(hb));
        }
#endif
        const int* rsm = h->h_result->resume;   // { next bucket, next bucket, bucket with fallback items pending }
        if (rsm[2] < 0) {
            if (rsm[0] >= (int)nb) break;
            return fail(h, LK_ERR_STATE, "the grid-resident kernel stopped without a reason at bucket " + std::to_string(rsm[0]) + " of " + std::to_string(nb));
        }
        if (round > nb + 4 || rsm[0] <= b0) return fail(h, LK_ERR_STATE, "the grid-resident kernel does not advance");
        h->grid_relaunches += 1;
        LAUNCH(h, "resident_fallback", hipLaunchKernelGGL(lk_resident_fallback_kernel, dim3(8), dim3(LK_MB), 0, h->stream, h->map, h->pr, h->d_snap, d_pts, rg, 0u, 0, d_rs,
                                                          reinterpret_cast<unsigned int*>(dr + o_sync)));   // (also zeroes the barrier arrivals, abort word and XCC ids for the next launch)
        b0 = rsm[0];
        if (b0 >= (int)nb) {   // they were the last bucket's
            if ((rc = finish_scan(h, pose, nullptr))) return rc;
            break;
        }
    }
    return LK_OK;
}
// a scan is taken by the resident kernel when all its buckets are small (LEGKILO_RESIDENT=0: always per-bucket launches)
static bool resident_enabled(const lk_handle* h) { return h->resident_enable && !h->profiling && !h->spec_enable; }
#define LK_GRIDSCAN_AUTO_MAX 4096   // largest bucket of a scan the grid-resident kernel takes by default: 51 x 1 960 points 2.53 -> 2.34 ms per scan, 5 x 20 000 0.43 -> 0.54 (it runs on G <= 128 workgroups)
static bool grid_enabled(const lk_handle* h) { return h->gridscan_mode != 0 && !h->profiling && !h->spec_enable; }
static bool grid_takes(const lk_handle* h, size_t smallest, size_t biggest) {
    return grid_enabled(h) && smallest > (size_t)LK_SMALL_MAX && (h->gridscan_mode == 2 || biggest <= (size_t)LK_GRIDSCAN_AUTO_MAX);
}

static int run_scan(lk_handle* h, const lk_point* pts, const lk_point* d_pts, size_t n, double t_begin, const lk_imu* imus,
                    size_t n_imu, const lk_kin_imu* kins, size_t n_kin, float* xyz_world_out, lk_pose* out) {
    int rc = zero_scan_counters(h, 0, 1);
    if (rc) return rc;
    if (resident_enabled(h) || grid_enabled(h)) {
        std::vector<unsigned long long> bstart;
        std::vector<double> btime;
        size_t biggest = 0, smallest = n;
        for (size_t i = 0; i < n;) {   // runs of equal curvature = buckets (KILO.cc:375-378)
            size_t j = i + 1;
            while (j < n && pts[i].curvature == pts[j].curvature) j++;
            bstart.push_back(i);
            btime.push_back(t_begin + pts[i].curvature);
            biggest = std::max(biggest, j - i);
            smallest = std::min(smallest, j - i);
            i = j;
        }
        bstart.push_back(n);
        if (grid_takes(h, smallest, biggest) && n_imu == 0 && n_kin == 0) {   // every bucket takes the large-bucket kernels: one grid-resident launch
            lk_pose pose;
            rc = run_scan_grid(h, d_pts, bstart, btime, biggest, xyz_world_out ? h->d_world : nullptr, &pose);
            if (rc) return rc;
            std::vector<float> w;
            if (xyz_world_out) {
                w.resize(4 * n);
                HIPCHK(h, hipMemcpyAsync(w.data(), h->d_world, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
            if (xyz_world_out)
                for (size_t i = 0; i < n; ++i)
                    for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
            if (out) *out = pose;
            return LK_OK;
        }
        if (resident_enabled(h) && biggest <= LK_RESIDENT_MAX) {
            lk_pose pose;
            rc = run_scan_resident(h, d_pts, bstart, btime, n_kin ? (const void*)kins : (const void*)imus, n_kin ? n_kin : n_imu, n_kin ? 2 : (n_imu ? 1 : 0),
                                   xyz_world_out ? h->d_world : nullptr, &pose);
            if (rc) return rc;
            std::vector<float> w;
            if (xyz_world_out) {
                w.resize(4 * n);
                HIPCHK(h, hipMemcpyAsync(w.data(), h->d_world, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
            }
            if (xyz_world_out)
                for (size_t i = 0; i < n; ++i)
                    for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
            if (out) *out = pose;
            return LK_OK;
        }
    }
    if (h->spec_enable && (rc = backup_filter(h))) return rc;
    size_t qi = 0, qk = 0;
    size_t idx_i = 0;
    bool pre_predicted = false;
    while (idx_i < n) {  // KILO.cc:375-395
        double cur_point_time = t_begin + pts[idx_i].curvature;
        size_t idx_j = idx_i + 1;
        while (idx_j < n && pts[idx_i].curvature == pts[idx_j].curvature) idx_j++;
        while (qi < n_imu && imus[qi].stamp < cur_point_time) {
            if ((rc = enqueue_imu(h, &imus[qi]))) return rc;
            ++qi;
        }
        while (qk < n_kin && kins[qk].time_stamp < cur_point_time) {
            if ((rc = enqueue_kin(h, &kins[qk]))) return rc;
            ++qk;
        }
        // the next bucket's predict rides in this bucket's launch when nothing lies in between and both are large buckets
        double t_next = NAN;
        if (idx_j < n) {
            size_t idx_k = idx_j + 1;
            while (idx_k < n && pts[idx_j].curvature == pts[idx_k].curvature) idx_k++;
            const double tn = t_begin + pts[idx_j].curvature;
            const bool msg_between = (qi < n_imu && imus[qi].stamp < tn) || (qk < n_kin && kins[qk].time_stamp < tn);
            if (!msg_between && idx_k - idx_j > LK_SMALL_MAX && idx_j - idx_i > LK_SMALL_MAX) t_next = tn;
        }
        rc = enqueue_bucket(h, d_pts + idx_i, (int)(idx_j - idx_i), cur_point_time,
                            xyz_world_out ? h->d_world + 4 * idx_i : nullptr, true, t_next, &pre_predicted);
        if (rc) return rc;
        idx_i = idx_j;
    }
    if ((rc = spec_join(h))) return rc;   // the last buckets' inserts (pipelined stream path) precede the read-backs below
    std::vector<float> w;
    if (xyz_world_out) {
        w.resize(4 * n);
        HIPCHK(h, hipMemcpyAsync(w.data(), h->d_world, sizeof(float) * 4 * n, hipMemcpyDeviceToHost, h->stream));
    }
    lk_pose pose;
    rc = finish_scan(h, &pose);
    if (rc) return rc;
    if (xyz_world_out)
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < 3; ++c) xyz_world_out[3 * i + c] = w[4 * i + c];
    if (out) *out = pose;
    return LK_OK;
}

int lk_process_scan_dev(lk_handle* h, const lk_point* d_pts, size_t n, double t_begin, const uint32_t* bucket_off,
                        const double* bucket_dt, size_t n_buckets, lk_pose* out) {
    CHECK_H(h);
    if (n == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty scan");
    if (n > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "scan exceeds max_scan_points");
    if (!d_pts || !bucket_off || !bucket_dt) return fail(h, LK_ERR_INVALID, "null argument");
    // the same table rules as the ragged batch entry, checked before any path is chosen: offsets non-decreasing and inside the scan,
    // times finite and non-decreasing (KILO.cc:367-370 sorts the scan by time)
    if (bucket_off[n_buckets] > n) return fail(h, LK_ERR_INVALID, "bucket offsets run past the scan");
    for (size_t b = 0; b < n_buckets; ++b) {
        if (bucket_off[b + 1] < bucket_off[b]) return fail(h, LK_ERR_INVALID, "bucket offsets must be non-decreasing");
        if (!std::isfinite(bucket_dt[b]) || (b > 0 && bucket_dt[b] < bucket_dt[b - 1])) return fail(h, LK_ERR_INVALID, "bucket times must be finite and non-decreasing");
    }
    int rc = zero_scan_counters(h, 0, 1);
    if (rc) return rc;
    if (resident_enabled(h) || grid_enabled(h)) {
        std::vector<unsigned long long> bstart;
        std::vector<double> btime;
        uint32_t biggest = 0, smallest = 0xffffffffu;
        for (size_t b = 0; b < n_buckets; ++b) {
            if (bucket_off[b + 1] <= bucket_off[b]) continue;
            bstart.push_back(bucket_off[b]);
            btime.push_back(t_begin + bucket_dt[b]);
            biggest = std::max(biggest, bucket_off[b + 1] - bucket_off[b]);
            smallest = std::min(smallest, bucket_off[b + 1] - bucket_off[b]);
        }
        bstart.push_back(bucket_off[n_buckets]);
        if (!btime.empty() && grid_takes(h, smallest, biggest)) {
            lk_pose pose;
            if ((rc = run_scan_grid(h, d_pts, bstart, btime, biggest, nullptr, &pose))) return rc;
            if (out) *out = pose;
            return LK_OK;
        }
        if (!btime.empty() && resident_enabled(h) && biggest <= LK_RESIDENT_MAX) {
            lk_pose pose;
            if ((rc = run_scan_resident(h, d_pts, bstart, btime, nullptr, 0, 0, nullptr, &pose))) return rc;
            if (out) *out = pose;
            return LK_OK;
        }
    }
    if (h->spec_enable && (rc = backup_filter(h))) return rc;
    bool pre_predicted = false;
    for (size_t b = 0; b < n_buckets; ++b) {
        int nb = (int)(bucket_off[b + 1] - bucket_off[b]);
        if (nb <= 0) continue;
        double t_next = NAN;
        if (b + 1 < n_buckets && nb > LK_SMALL_MAX && (int)(bucket_off[b + 2] - bucket_off[b + 1]) > LK_SMALL_MAX) t_next = t_begin + bucket_dt[b + 1];
        rc = enqueue_bucket(h, d_pts + bucket_off[b], nb, t_begin + bucket_dt[b], nullptr, true, t_next, &pre_predicted);
        if (rc) return rc;
    }
    if ((rc = spec_join(h))) return rc;
    lk_pose pose;
    rc = finish_scan(h, &pose);
    if (rc) return rc;
    if (out) *out = pose;
    return LK_OK;
}

// ------------------------------------------------------------------ voxel-grid centroid filter + time sort
static int pre_reserve(lk_handle* h, size_t n) {
    if (n <= h->pre_cap) return LK_OK;
    void** ptrs[] = {(void**)&h->pre_raw, (void**)&h->pre_cells, (void**)&h->pre_out, (void**)&h->pre_k0, (void**)&h->pre_k1,
                     (void**)&h->pre_flags, (void**)&h->pre_pos, (void**)&h->pre_misc, (void**)&h->pre_v0, (void**)&h->pre_v1,
                     (void**)&h->pre_starts, (void**)&h->pre_tmp};
    for (void** p : ptrs)
        if (*p) hipFree(*p), *p = nullptr;
    size_t cap = n + n / 4 + 1024;
    HIPCHK(h, hipMalloc(&h->pre_raw, sizeof(lk_point) * cap));
    HIPCHK(h, hipMalloc(&h->pre_cells, sizeof(lk_point) * cap));
    HIPCHK(h, hipMalloc(&h->pre_out, sizeof(lk_point) * cap));
    HIPCHK(h, hipMalloc(&h->pre_k0, sizeof(unsigned int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_k1, sizeof(unsigned int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_flags, sizeof(unsigned int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_pos, sizeof(unsigned int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_misc, sizeof(unsigned int) * 32));
    HIPCHK(h, hipMalloc(&h->pre_v0, sizeof(int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_v1, sizeof(int) * cap));
    HIPCHK(h, hipMalloc(&h->pre_starts, sizeof(int) * cap));
    size_t t1 = 0, t2 = 0;
    HIPCHK(h, lk_prim_sort_pairs(nullptr, t1, h->pre_k0, h->pre_k1, h->pre_v0, h->pre_v1, cap, 0, 32, h->stream));
    HIPCHK(h, lk_prim_exclusive_scan(nullptr, t2, h->pre_flags, h->pre_pos, cap, h->stream));
    h->pre_tmp_bytes = std::max(t1, t2);
    HIPCHK(h, hipMalloc(&h->pre_tmp, h->pre_tmp_bytes));
    h->pre_cap = cap;
    return LK_OK;
}

int lk_decode_scan_dev(lk_handle* h, const void* d_msg, size_t n_points, const lk_cloud_layout* layout, double time_scale,
                       int filter_num, float blind, double header_stamp, lk_point* d_out, size_t* n_out, double* begin_time,
                       double* end_time) {
    CHECK_H(h);
    if (!d_msg || !layout || !d_out || !n_out || n_points == 0 || filter_num < 1 || n_points > 0x7fffffffu)
        return fail(h, LK_ERR_INVALID, "lk_decode_scan: bad argument");
    if (layout->lidar_type < 1 || layout->lidar_type > 3) return fail(h, LK_ERR_INVALID, "lidar_type must be 1, 2 or 3");
    const uint32_t tsz = layout->lidar_type == 3 ? 8u : 4u;
    if (layout->off_x + 4 > layout->point_step || layout->off_y + 4 > layout->point_step || layout->off_z + 4 > layout->point_step ||
        layout->off_time + tsz > layout->point_step)
        return fail(h, LK_ERR_INVALID, "field offsets exceed point_step");
    int rc = pre_reserve(h, n_points);
    if (rc) return rc;
    LkDecodeArgs a;
    a.lay = *layout, a.time_scale = time_scale, a.filter_num = filter_num, a.blind = blind;
    const int n = (int)n_points, nb = (n + 255) / 256;
    unsigned int* n_out_d = h->pre_misc + 7;
    double* fl = reinterpret_cast<double*>(h->pre_misc + 8);
    LAUNCH(h, "decode_flags", hipLaunchKernelGGL(lk_decode_flags_kernel, dim3(nb), dim3(256), 0, h->stream,
                                                 (const unsigned char*)d_msg, n, a, h->pre_flags));
    size_t tb = h->pre_tmp_bytes;
    HIPCHK(h, lk_prim_exclusive_scan(h->pre_tmp, tb, h->pre_flags, h->pre_pos, n_points, h->stream));
    LAUNCH(h, "decode_scatter", hipLaunchKernelGGL(lk_decode_scatter_kernel, dim3(nb), dim3(256), 0, h->stream,
                                                   (const unsigned char*)d_msg, n, a, h->pre_flags, h->pre_pos, d_out, n_out_d, fl));
    unsigned int cnt = 0;
    double tfl[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(&cnt, n_out_d, sizeof(cnt), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(tfl, fl, sizeof(tfl), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_out = cnt;
    const double base = layout->lidar_type == 3 ? 0.0 : header_stamp;  // lidar_processing.cc:34-35,63-64 vs :91-92
    if (begin_time) *begin_time = base + tfl[0];
    if (end_time) *end_time = base + tfl[1];
    return LK_OK;
}

int lk_decode_scan(lk_handle* h, const void* msg, size_t n_points, const lk_cloud_layout* layout, double time_scale, int filter_num,
                   float blind, double header_stamp, lk_point* out, size_t* n_out, double* begin_time, double* end_time) {
    CHECK_H(h);
    if (!msg || !layout || !out || n_points == 0) return fail(h, LK_ERR_INVALID, "lk_decode_scan: bad argument");
    void* d_msg = nullptr;
    const size_t bytes = n_points * (size_t)layout->point_step;
    HIPCHK(h, hipMalloc(&d_msg, bytes));
    int rc = pre_reserve(h, n_points);
    if (rc == LK_OK) {
        hipMemcpyAsync(d_msg, msg, bytes, hipMemcpyHostToDevice, h->stream);
        rc = lk_decode_scan_dev(h, d_msg, n_points, layout, time_scale, filter_num, blind, header_stamp, h->pre_out, n_out, begin_time, end_time);
        if (rc == LK_OK) {
            hipMemcpyAsync(out, h->pre_out, sizeof(lk_point) * (*n_out), hipMemcpyDeviceToHost, h->stream);
            hipStreamSynchronize(h->stream);
        }
    }
    hipFree(d_msg);
    return rc;
}

int lk_preprocess_scan_dev(lk_handle* h, const lk_point* d_raw, size_t n_raw, float leaf, lk_point* d_out, size_t* n_out) {
    CHECK_H(h);
    if (!d_raw || !d_out || !n_out || n_raw == 0 || !(leaf > 0.f)) return fail(h, LK_ERR_INVALID, "lk_preprocess_scan: bad argument");
    if (n_raw > 0x7fffffffu) return fail(h, LK_ERR_INVALID, "too many points");
    int rc = pre_reserve(h, n_raw);
    if (rc) return rc;
    const int n = (int)n_raw;
    const int nb = (n + 255) / 256;
    const float inv = 1.0f / leaf;  // inverse_leaf_size_, float as in PCL
    int* mm = reinterpret_cast<int*>(h->pre_misc);
    const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
    HIPCHK(h, hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
    unsigned int* err = h->pre_misc + 6;
    unsigned int* ncells_d = h->pre_misc + 7;
    LAUNCH(h, "pre_minmax", hipLaunchKernelGGL(lk_pre_minmax_kernel, dim3(std::min(nb, 1024)), dim3(256), 0, h->stream, d_raw, n, mm));
    LAUNCH(h, "pre_cellidx", hipLaunchKernelGGL(lk_pre_cellidx_kernel, dim3(nb), dim3(256), 0, h->stream, d_raw, n, inv, mm, h->pre_k0,
                                                h->pre_v0, err));
    size_t tb = h->pre_tmp_bytes;
    HIPCHK(h, lk_prim_sort_pairs(h->pre_tmp, tb, h->pre_k0, h->pre_k1, h->pre_v0, h->pre_v1, n_raw, 0, 32, h->stream));
    LAUNCH(h, "pre_heads", hipLaunchKernelGGL(lk_pre_heads_kernel, dim3(nb), dim3(256), 0, h->stream, h->pre_k1, n, h->pre_flags));
    tb = h->pre_tmp_bytes;
    HIPCHK(h, lk_prim_exclusive_scan(h->pre_tmp, tb, h->pre_flags, h->pre_pos, n_raw, h->stream));
    LAUNCH(h, "pre_starts", hipLaunchKernelGGL(lk_pre_starts_kernel, dim3(nb), dim3(256), 0, h->stream, h->pre_flags, h->pre_pos, n,
                                               h->pre_starts, ncells_d));
    LAUNCH(h, "pre_centroid", hipLaunchKernelGGL(lk_pre_centroid_kernel, dim3(nb), dim3(256), 0, h->stream, d_raw, h->pre_v1,
                                                 h->pre_starts, ncells_d, n, h->pre_cells, h->pre_k0, h->pre_v0));
    unsigned int host_misc[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(host_misc, err, sizeof(host_misc), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (host_misc[0]) return fail(h, LK_ERR_INVALID, "voxel grid leaf too small for the cloud extent (index overflow)");
    const size_t nc = host_misc[1];
    tb = h->pre_tmp_bytes;
    HIPCHK(h, lk_prim_sort_pairs(h->pre_tmp, tb, h->pre_k0, h->pre_k1, h->pre_v0, h->pre_v1, nc, 0, 32, h->stream));
    LAUNCH(h, "pre_gather", hipLaunchKernelGGL(lk_pre_gather_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream,
                                               h->pre_cells, h->pre_v1, (int)nc, d_out));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_out = nc;
    return LK_OK;
}

int lk_preprocess_scan(lk_handle* h, const lk_point* raw, size_t n_raw, float leaf, lk_point* out_sorted, size_t* n_out) {
    CHECK_H(h);
    if (!raw || !out_sorted || n_raw == 0) return fail(h, LK_ERR_INVALID, "lk_preprocess_scan: bad argument");
    int rc = pre_reserve(h, n_raw);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->pre_raw, raw, sizeof(lk_point) * n_raw, hipMemcpyHostToDevice, h->stream));
    rc = lk_preprocess_scan_dev(h, h->pre_raw, n_raw, leaf, h->pre_out, n_out);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(out_sorted, h->pre_out, sizeof(lk_point) * (*n_out), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

int lk_process_raw_scan(lk_handle* h, const lk_point* raw, size_t n_raw, float leaf, double t_begin, const lk_imu* imus,
                        size_t n_imu, const lk_kin_imu* kins, size_t n_kin, size_t* n_down, lk_pose* out) {
    CHECK_H(h);
    if (!raw || n_raw == 0) return fail(h, LK_ERR_INVALID, "empty scan");
    if (n_imu && n_kin) return fail(h, LK_ERR_INVALID, "pass either IMU or kin+IMU messages, not both");
    int rc = pre_reserve(h, n_raw);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->pre_raw, raw, sizeof(lk_point) * n_raw, hipMemcpyHostToDevice, h->stream));
    size_t nd = 0;
    rc = lk_preprocess_scan_dev(h, h->pre_raw, n_raw, leaf, h->pre_out, &nd);
    if (rc) return rc;
    if (nd > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "downsampled scan exceeds max_scan_points");
    std::vector<lk_point> sorted(nd);
    HIPCHK(h, hipMemcpyAsync(sorted.data(), h->pre_out, sizeof(lk_point) * nd, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_down) *n_down = nd;
    return run_scan(h, sorted.data(), h->pre_out, nd, t_begin, imus, n_imu, kins, n_kin, nullptr, out);
}

// ------------------------------------------------------------------ batch replay against the frozen map
int lk_batch_set_priors(lk_handle* h, const double* x36, const double* P900, size_t n_scans) {
    CHECK_H(h);
    if (n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans exceeds n_slots");
    HIPCHK(h, hipMemcpy2DAsync(h->d_filters[0].x, sizeof(LkFilter), x36, sizeof(double) * 36, sizeof(double) * 36, n_scans,
                               hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(h->d_filters[0].P, sizeof(LkFilter), P900, sizeof(double) * 900, sizeof(double) * 900, n_scans,
                               hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

int lk_batch_set_priors_dev(lk_handle* h, const double* d_x36, const double* d_P900, size_t n_scans) {
    CHECK_H(h);
    if (n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans exceeds n_slots");
    if (!d_x36 || !d_P900) return fail(h, LK_ERR_INVALID, "null prior buffer");
    HIPCHK(h, hipMemcpy2DAsync(h->d_filters[0].x, sizeof(LkFilter), d_x36, sizeof(double) * 36, sizeof(double) * 36, n_scans,
                               hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipMemcpy2DAsync(h->d_filters[0].P, sizeof(LkFilter), d_P900, sizeof(double) * 900, sizeof(double) * 900, n_scans,
                               hipMemcpyDeviceToDevice, h->stream));
    return LK_OK;
}

// ---- a batch put into root-voxel order, bucket by bucket (lk_batch_sort_by_voxel_dev)
// key of point i of scan s: its root voxel under the slot's PRIOR pose (load_bucket_const / point_world / key_floor: what the first
// bucket's residual pass will compute), ten bits per axis - a wrap-around beyond 1 024 voxels only costs locality; value: the point's
// index in the batch
__global__ void __launch_bounds__(256)
    lk_sort_keys_kernel(LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts, size_t n_pts, unsigned int* __restrict__ keys,
                        unsigned int* __restrict__ vals) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pts) return;
    BucketConst bc;
    load_bucket_const<false>(&filters[blockIdx.y], pr, bc);
    const size_t j = (size_t)blockIdx.y * n_pts + i;
    const float4 p = reinterpret_cast<const float4*>(pts)[j];
    const V3 pw = point_world(p.x, p.y, p.z, bc, pr);
    int key[3];
    key_floor(pw, pr.voxel_size_f, key);
    keys[j] = (((unsigned int)key[2] & 1023u) << 20) | (((unsigned int)key[1] & 1023u) << 10) | ((unsigned int)key[0] & 1023u);
    vals[j] = (unsigned int)j;
}
__global__ void __launch_bounds__(256) lk_sort_gather_kernel(const lk_point* __restrict__ in, lk_point* __restrict__ out, const unsigned int* __restrict__ vals, size_t n) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n) reinterpret_cast<float4*>(out)[j] = reinterpret_cast<const float4*>(in)[vals[j]];
}
// Every time bucket of every scan of a device-resident batch in root-voxel order: a residual wave's 64 points then look at a handful of voxels
// instead of sixty (1.8 -> 0.6 L2 requests per point on the bench batch, 2.25 -> 1.6 ms per step).  Points keep their bucket; the order inside
// a bucket is one of the legal outcomes of KILO.cc:369's sort (equal curvature).  Stable segmented radix sort (rocPRIM) of (key, index) pairs +
// one gather; priors from lk_batch_set_priors(_dev).  Once per loaded batch, not per replay; in and out must not overlap.
static int sort_by_voxel(lk_handle* h, const lk_point* d_in, lk_point* d_out, uint32_t first_slot, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets);
int lk_batch_sort_by_voxel_dev(lk_handle* h, const lk_point* d_in, lk_point* d_out, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets) {
    CHECK_H(h);
    return sort_by_voxel(h, d_in, d_out, 0, n_scans, n_pts, bucket_off, n_buckets);
}
static int sort_by_voxel(lk_handle* h, const lk_point* d_in, lk_point* d_out, uint32_t first_slot, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets) {
    if (!d_in || !d_out || !bucket_off) return fail(h, LK_ERR_INVALID, "null argument");
    if (n_scans == 0 || n_pts == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty batch");
    if ((size_t)first_slot + n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans exceeds n_slots");
    if (bucket_off[0] != 0 || bucket_off[n_buckets] != n_pts) return fail(h, LK_ERR_INVALID, "bucket_off must cover the scan: bucket_off[0] == 0, bucket_off[n_buckets] == n_pts");
    for (size_t b = 0; b < n_buckets; ++b)
        if (bucket_off[b + 1] < bucket_off[b]) return fail(h, LK_ERR_INVALID, "bucket_off must be non-decreasing");
    const size_t total = n_scans * n_pts;
    if (total >= (size_t)1 << 32) return fail(h, LK_ERR_INVALID, "batch exceeds 2^32 points");
    const size_t nseg = n_scans * n_buckets;
    std::vector<unsigned int> offs(nseg + 1);
    for (size_t s = 0; s < n_scans; ++s)
        for (size_t b = 0; b < n_buckets; ++b) offs[s * n_buckets + b] = (unsigned int)(s * n_pts + bucket_off[b]);
    offs[nseg] = (unsigned int)total;
    DevTemps tmp;
    unsigned int *k0 = nullptr, *k1 = nullptr, *v0 = nullptr, *v1 = nullptr, *d_off = nullptr;
    void* d_tmp = nullptr;
    HIPCHK(h, tmp.alloc(&k0, sizeof(unsigned int) * total));
    HIPCHK(h, tmp.alloc(&k1, sizeof(unsigned int) * total));
    HIPCHK(h, tmp.alloc(&v0, sizeof(unsigned int) * total));
    HIPCHK(h, tmp.alloc(&v1, sizeof(unsigned int) * total));
    HIPCHK(h, tmp.alloc(&d_off, sizeof(unsigned int) * (nseg + 1)));
    HIPCHK(h, hipMemcpyAsync(d_off, offs.data(), sizeof(unsigned int) * (nseg + 1), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lk_sort_keys_kernel, dim3((unsigned int)((n_pts + 255) / 256), (unsigned int)n_scans), dim3(256), 0, h->stream, h->pr, h->d_filters + first_slot, d_in, n_pts, k0, v0);
    HIPCHK(h, hipGetLastError());
    size_t tmp_bytes = 0;
    HIPCHK(h, lk_prim_segmented_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, (unsigned int)total, (unsigned int)nseg, d_off, d_off + 1, 0, 30, h->stream));
    HIPCHK(h, tmp.alloc(&d_tmp, std::max<size_t>(tmp_bytes, 16)));
    HIPCHK(h, lk_prim_segmented_sort_pairs(d_tmp, tmp_bytes, k0, k1, v0, v1, (unsigned int)total, (unsigned int)nseg, d_off, d_off + 1, 0, 30, h->stream));
    hipLaunchKernelGGL(lk_sort_gather_kernel, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, h->stream, d_in, d_out, v1, total);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));   // offs and the temporaries go out of scope
    return LK_OK;
}

// ------------------------------------------------------------------ input order of device-resident batches (lk_batch_order)
// The batch residual kernel is 1.4 x slower on a batch whose buckets come in a random order than on one in voxel order (1.80 against 0.63 L2
// requests per point).  So that this is not the caller's problem, the frozen-map batch entries keep track of the batches they are given - (device
// pointer, shape, bucket bounds) + a stamp of 4 096 sampled points, taken ON THE DEVICE by a one-workgroup kernel ahead of the residual launches
// (no host round trip) - and once a batch has come back unchanged twice they sort it, unless every bucket already is in voxel order, ONCE into a
// library-owned copy (sort_by_voxel: 5.6 ms per 1 024 x 100 000 points) which its later replays read.  Any order inside a bucket is a legal outcome
// of the reference's sort of equal time stamps (KILO.cc:369), so the copy is a batch of the same scans.  A batch that is replayed once or twice - new
// scans streamed through one staging buffer - is never sorted; new content in a sorted batch's buffer is noticed by the same kernel, which tells the
// residual kernel to read the caller's buffer (ResidualOut::alt_use) and starts the count again.  (An in-place edit that changes none of the sampled
// points is not noticed: lk_batch_changed.)
__global__ void __launch_bounds__(256) lk_sort_check_kernel(LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts, size_t n_pts,
                                                            unsigned int* __restrict__ unsorted) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 || i >= n_pts) return;
    BucketConst bc;
    load_bucket_const<false>(&filters[blockIdx.y], pr, bc);
    const float4* p4 = reinterpret_cast<const float4*>(pts) + (size_t)blockIdx.y * n_pts;
    const float4 a = p4[i - 1], b = p4[i];
    if (a.w != b.w) return;   // a bucket boundary (runs of equal curvature, KILO.cc:375-378)
    int ka[3], kb[3];
    key_floor(point_world(a.x, a.y, a.z, bc, pr), pr.voxel_size_f, ka);
    key_floor(point_world(b.x, b.y, b.z, bc, pr), pr.voxel_size_f, kb);
    const unsigned int sa = (((unsigned int)ka[2] & 1023u) << 20) | (((unsigned int)ka[1] & 1023u) << 10) | ((unsigned int)ka[0] & 1023u);
    const unsigned int sb = (((unsigned int)kb[2] & 1023u) << 20) | (((unsigned int)kb[1] & 1023u) << 10) | ((unsigned int)kb[0] & 1023u);
    if (sb < sa) atomicAdd(unsorted, 1u);
}
// Stamp of 4 096 points spread over the batch, compared on the device with the stamp of the replay before (ref[0]); ref[1] = which buffer THIS replay's
// residual launches read (1: the library's copy); *seen (host-mapped) = how many replays in a row have found the same stamp (LK_ORD_SORTED once a copy
// of this very content exists).  mode LK_ORD_SET: the copy has just been made from the buffer.
#define LK_ORD_PROBE 0     // no copy (yet): count how often the content repeats
#define LK_ORD_CHECK 1     // a copy exists: use it if the buffer still holds what it was made from
#define LK_ORD_SET 2
#define LK_ORD_SORTED 1000000u
__global__ void __launch_bounds__(256) lk_batch_stamp_kernel(const lk_point* __restrict__ pts, size_t total, unsigned long long* __restrict__ ref,
                                                             unsigned int* __restrict__ seen, int mode) {
    __shared__ unsigned long long acc;
    if (threadIdx.x == 0) acc = 0ull;
    __syncthreads();
    const size_t stride = total / 4096 ? total / 4096 : 1;
    unsigned long long hsh = 0ull;
    for (int k = 0; k < 16; ++k) {
        const size_t idx = ((size_t)threadIdx.x * 16 + k) * stride;
        if (idx >= total) break;
        const uint4 v = reinterpret_cast<const uint4*>(pts)[idx];
        unsigned long long m = ((unsigned long long)v.x | ((unsigned long long)v.y << 32)) ^ (0x9E3779B97F4A7C15ull * (idx + 1));
        m = (m ^ (m >> 31)) * 0xBF58476D1CE4E5B9ull;
        m ^= ((unsigned long long)v.z | ((unsigned long long)v.w << 32)) * 0x94D049BB133111EBull;
        m = (m ^ (m >> 29)) * 0xC2B2AE3D27D4EB4Full;
        hsh += m ^ (m >> 32);   // a sum: the order the threads arrive in does not matter
    }
    atomicAdd(&acc, hsh);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long st = acc | 1ull;   // never 0: ref[0] == 0 means "no stamp yet"
        if (mode == LK_ORD_SET) {
            ref[0] = st, ref[1] = 1ull, *seen = LK_ORD_SORTED;
        } else if (st == ref[0]) {
            ref[1] = mode == LK_ORD_CHECK ? 1ull : 0ull;
            if (mode == LK_ORD_PROBE) *seen = *seen + 1u;
        } else {   // other content than last time: this replay reads the caller's buffer, and the count starts again
            ref[0] = st, ref[1] = 0ull, *seen = 1u;
        }
    }
}
static uint64_t hash_offsets(const uint32_t* off, size_t n) {
    uint64_t x = 1469598103934665603ull;
    for (size_t i = 0; i <= n; ++i) x = (x ^ off[i]) * 1099511628211ull;
    return x;
}
// Which buffer the residual launches of a frozen-map batch replay read.  `st`: the stream the slots' priors were armed on and the batch will run on.
// Fills ro->alt_pts / alt_use (null: the caller's buffer as given).  One small launch per call; the call that makes the copy synchronises.
static int batch_ordered(lk_handle* h, const lk_point* d_pts, uint32_t first_slot, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets,
                         hipStream_t st, ResidualOut* ro, bool now = false) {
    ResidualOut none;
    if (!ro) ro = &none;
    ro->alt_pts = nullptr, ro->alt_use = nullptr;
    if ((h->batch_order_mode == 0 && !now) || n_scans * n_pts < 4096 || n_scans * n_pts >= ((size_t)1 << 32)) return LK_OK;
    const uint64_t oh = hash_offsets(bucket_off, n_buckets);
    lk_handle::OrdEntry* e = nullptr;
    for (auto& c : h->ord)
        if (c.src == d_pts && c.n_scans == n_scans && c.n_pts == n_pts && c.n_buckets == n_buckets && c.off_hash == oh) e = &c;
    if (!e) {   // a batch not seen before (or forgotten): the least recently used entry goes; nothing is examined yet, only its stamp is taken
        e = h->ord[0].tick <= h->ord[1].tick ? &h->ord[0] : &h->ord[1];
        if (!e->d_ref) {
            HIPCHK(h, hipMalloc(&e->d_ref, 2 * sizeof(unsigned long long)));
            HIPCHK(h, hipHostMalloc(&e->h_seen, sizeof(unsigned int), hipHostMallocMapped));
            HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&e->d_seen), e->h_seen, 0));
        } else {
            HIPCHK(h, hipStreamSynchronize(h->stream));   // replays of the entry's former batch may still be using its stamp words
            for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)
                if (h->side[i]) HIPCHK(h, hipStreamSynchronize(h->side[i]));
        }
        *reinterpret_cast<volatile unsigned int*>(e->h_seen) = 0u;
        HIPCHK(h, hipMemsetAsync(e->d_ref, 0, 2 * sizeof(unsigned long long), st));
        e->src = d_pts, e->n_scans = n_scans, e->n_pts = n_pts, e->n_buckets = n_buckets, e->off_hash = oh;
        e->as_given = false, e->have_copy = false;
    }
    e->tick = ++h->ord_tick;
    if (e->as_given) return LK_OK;
    const unsigned int seen = *reinterpret_cast<volatile unsigned int*>(e->h_seen);   // written by the device; may lag behind the replays still in flight
    if (e->have_copy && seen < LK_ORD_SORTED) {   // a replay found other content in the caller's buffer: the copy is of no use any more
        e->have_copy = false;
        ++h->ord_stale;
    }
    if (!e->have_copy && (now || seen >= (unsigned int)h->batch_order_after) && seen < LK_ORD_SORTED) {
        // the same content has been replayed often enough to be worth 5.6 ms: examine it, and unless it already is in voxel order, sort it into the copy
        ++h->ord_examined;
        HIPCHK(h, hipStreamSynchronize(st));   // the priors are armed
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)   // (replays in flight on the other streams write the entry's stamp words)
            if (h->side[i]) HIPCHK(h, hipStreamSynchronize(h->side[i]));
        unsigned int* d_uns = reinterpret_cast<unsigned int*>(e->d_ref + 1);   // (borrowed: ref[1] is written again by the stamp kernel below)
        HIPCHK(h, hipMemsetAsync(d_uns, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(lk_sort_check_kernel, dim3((unsigned int)((n_pts + 255) / 256), (unsigned int)n_scans), dim3(256), 0, st, h->pr, h->d_filters + first_slot, d_pts, n_pts, d_uns);
        HIPCHK(h, hipGetLastError());
        unsigned int uns = 0;
        HIPCHK(h, hipMemcpyAsync(&uns, d_uns, sizeof(uns), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        HIPCHK(h, hipMemsetAsync(d_uns, 0, sizeof(unsigned long long), st));
        bool sort_it = uns != 0;
        if (sort_it) {
            const size_t bytes = sizeof(lk_point) * n_scans * n_pts;
            if (e->copy_bytes < bytes) {
                if (e->copy) hipFree(e->copy), e->copy = nullptr, e->copy_bytes = 0;
                if (hipMalloc(&e->copy, bytes) != hipSuccess) {   // no room for a copy: this batch is replayed as given
                    (void)hipGetLastError();
                    e->copy = nullptr;
                    sort_it = false;
                } else {
                    e->copy_bytes = bytes;
                }
            }
        }
        if (!sort_it) {
            e->as_given = true;   // in voxel order already (or no memory): replayed where it lies from now on, no stamps
            return LK_OK;
        }
        const int rc = sort_by_voxel(h, d_pts, e->copy, first_slot, n_scans, n_pts, bucket_off, n_buckets);   // on h->stream, synchronous
        if (rc) return rc;
        ++h->ord_sorted;
        e->have_copy = true;
        hipLaunchKernelGGL(lk_batch_stamp_kernel, dim3(1), dim3(256), 0, st, d_pts, n_scans * n_pts, e->d_ref, e->d_seen, LK_ORD_SET);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
    } else {
        hipLaunchKernelGGL(lk_batch_stamp_kernel, dim3(1), dim3(256), 0, st, d_pts, n_scans * n_pts, e->d_ref, e->d_seen, e->have_copy ? LK_ORD_CHECK : LK_ORD_PROBE);
        HIPCHK(h, hipGetLastError());
    }
    if (e->have_copy) ro->alt_pts = e->copy, ro->alt_use = e->d_ref;
    return LK_OK;
}

// one workgroup per slot: state and covariance of a filter slot into dense [n][36] / [n][900] arrays
__global__ void __launch_bounds__(256) lk_states_gather_kernel(const LkFilter* __restrict__ filters, double* __restrict__ x36, double* __restrict__ P900) {
    const LkFilter* f = &filters[blockIdx.x];
    if (x36 && threadIdx.x < LK_STATE_DOUBLES) x36[(size_t)blockIdx.x * LK_STATE_DOUBLES + threadIdx.x] = f->x[threadIdx.x];
    if (P900)
        for (int e = threadIdx.x; e < 900; e += 256) P900[(size_t)blockIdx.x * 900 + e] = f->P[e];
}
int join_side_streams(lk_handle* h) {   // the asynchronous batch entry may still be running on a side stream
    for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)
        if (h->side[i]) {
            HIPCHK(h, hipEventRecord(h->ev_join[i], h->side[i]));
            HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
        }
    return LK_OK;
}
int lk_batch_order(lk_handle* h, int mode) {
    CHECK_H(h);
    if (mode != 0 && mode != 1) return fail(h, LK_ERR_INVALID, "mode must be LK_BATCH_ORDER_AS_GIVEN (0) or LK_BATCH_ORDER_AUTO (1)");
    h->batch_order_mode = mode;
    return LK_OK;
}
int lk_batch_prepare_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, const uint32_t* bucket_off, size_t n_buckets) {
    CHECK_H(h);
    if (!d_pts || !bucket_off) return fail(h, LK_ERR_INVALID, "null argument");
    if (n_scans == 0 || n_scans > h->cfg.n_slots || n_pts == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty batch, or n_scans exceeds n_slots");
    if (bucket_off[0] != 0 || bucket_off[n_buckets] != n_pts) return fail(h, LK_ERR_INVALID, "bucket_off must cover the scan: bucket_off[0] == 0, bucket_off[n_buckets] == n_pts");
    int rc = join_side_streams(h);
    if (rc) return rc;
    rc = batch_ordered(h, d_pts, 0, n_scans, n_pts, bucket_off, n_buckets, h->stream, nullptr, true);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_batch_changed(lk_handle* h) {
    CHECK_H(h);
    for (auto& e : h->ord) e.src = nullptr;   // every batch is a first sight again (the copies' memory is kept for the next one)
    return LK_OK;
}
int lk_batch_order_stats(lk_handle* h, uint64_t* out3) {
    CHECK_H(h);
    if (!out3) return fail(h, LK_ERR_INVALID, "null argument");
    out3[0] = h->ord_examined, out3[1] = h->ord_sorted, out3[2] = h->ord_stale;
    return LK_OK;
}
int lk_batch_get_states_dev(lk_handle* h, uint32_t first_slot, size_t n, double* d_x36, double* d_P900) {
    CHECK_H(h);
    if (n == 0) return LK_OK;
    if ((size_t)first_slot + n > (size_t)h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "slot range exceeds n_slots");
    int rc = join_side_streams(h);
    if (rc) return rc;
    hipLaunchKernelGGL(lk_states_gather_kernel, dim3((unsigned int)n), dim3(256), 0, h->stream, h->d_filters + first_slot, d_x36, d_P900);
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}
int lk_batch_get_states(lk_handle* h, uint32_t first_slot, size_t n, double* x36, double* P900) {
    CHECK_H(h);
    if (n == 0) return LK_OK;
    if ((size_t)first_slot + n > (size_t)h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "slot range exceeds n_slots");
    DevTemps tmp;
    double *d_x = nullptr, *d_P = nullptr;
    if (x36) HIPCHK(h, tmp.alloc(&d_x, sizeof(double) * LK_STATE_DOUBLES * n));
    if (P900) HIPCHK(h, tmp.alloc(&d_P, sizeof(double) * 900 * n));
    int rc = lk_batch_get_states_dev(h, first_slot, n, d_x, d_P);
    if (rc) return rc;
    if (x36) HIPCHK(h, hipMemcpyAsync(x36, d_x, sizeof(double) * LK_STATE_DOUBLES * n, hipMemcpyDeviceToHost, h->stream));
    if (P900) HIPCHK(h, hipMemcpyAsync(P900, d_P, sizeof(double) * 900 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}

__global__ void lk_set_times_kernel(LkFilter* filters, int n, double t) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) filters[s].last_predict_t = t, filters[s].last_update_t = t;
}

int lk_batch_replay_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double t_begin,
                        const uint32_t* bucket_off, const double* bucket_dt, size_t n_buckets, lk_pose* out) {
    CHECK_H(h);
    if (n_scans == 0 || n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans must be in [1, n_slots]");
    if (n_pts == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty scans");
    const int S = (int)n_scans;
    LkMap fmap;
    int rc = frozen_map(h, &fmap);
    if (rc) return rc;
    rc = zero_scan_counters(h, 0, (uint32_t)n_scans);
    if (rc) return rc;
    const auto res_kernel = batch_residual_kernel(h, fmap);
    hipLaunchKernelGGL(lk_set_times_kernel, dim3((S + 63) / 64), dim3(64), 0, h->stream, h->d_filters, S, t_begin);
    HIPCHK(h, hipGetLastError());
    ResidualOut ro;
    memset(&ro, 0, sizeof(ro));
    rc = batch_ordered(h, d_pts, 0, n_scans, n_pts, bucket_off, n_buckets, h->stream, &ro);   // the caller's buffer, or the library's voxel-ordered copy of it
    if (rc) return rc;
    const lk_point* const alt_base = ro.alt_pts;
    // non-empty buckets; update(k) and predict(k+1) share one launch (the map is frozen: nothing reads the state in between)
    std::vector<size_t> live;
    for (size_t b = 0; b < n_buckets; ++b) {
        if (bucket_off[b + 1] <= bucket_off[b]) continue;
        if ((size_t)(bucket_off[b + 1] - bucket_off[b]) > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "bucket exceeds max_scan_points");
        live.push_back(b);
    }
    // Slot groups on separate HIP streams (default 3 — measured best of 1..8 on MI355X; LEGKILO_REPLAY_GROUPS=1..4 overrides): the single-workgroup update/predict kernels of one group (half of the CUs
    // idle, latency-bound) overlap the residual kernel of the other group.  Groups touch disjoint filters / partials and
    // only read the map.  Profiling mode (per-launch events + sync) and small batches stay on one stream.
    const int ngroups = (!h->profiling && S >= 2 * h->replay_groups) ? h->replay_groups : 1;
    hipStream_t streams[lk_handle::kMaxGroups];
    streams[0] = h->stream;
    for (int g = 1; g < lk_handle::kMaxGroups; ++g) streams[g] = h->side[g - 1];
    if (ngroups > 1) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        for (int g = 1; g < ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(streams[g], h->ev_fork, 0));
    }
    for (size_t k = 0; k < live.size(); ++k) {
        const size_t b = live[k];
        const int nb = (int)(bucket_off[b + 1] - bucket_off[b]);
        const double t = t_begin + bucket_dt[b];
        const bool has_next = k + 1 < live.size();
        const double t_next = has_next ? t_begin + bucket_dt[live[k + 1]] : 0.0;
        const int nblk = (nb + LK_RB - 1) / LK_RB;
        for (int grp = 0; grp < ngroups; ++grp) {
            const int s0 = (int)((long)S * grp / ngroups), sn = (int)((long)S * (grp + 1) / ngroups) - s0;
            hipStream_t st = streams[grp];
            LkFilter* fl = h->d_filters + s0;
            double* parts = h->d_partials + (size_t)s0 * h->part_stride;
            const lk_point* pts = d_pts + (size_t)s0 * n_pts + bucket_off[b];
            ro.alt_pts = alt_base ? alt_base + (size_t)s0 * n_pts + bucket_off[b] : nullptr;
            if (ngroups == 1) {
                if (k == 0) {
                    if (h->wave_update)
                        LAUNCH(h, "predict", hipLaunchKernelGGL(lk_update_wave_kernel, dim3(sn), dim3(LK_WAVE), 0, st, fl, parts, 0,
                                                                h->part_stride, 0.0, h->d_Q, t, 2));
                    else
                        LAUNCH(h, "predict", hipLaunchKernelGGL(lk_predict_kernel, dim3(sn), dim3(LK_FB), 0, st, fl, h->d_Q, t));
                }
                const dim3 rgrid = batch_residual_grid(nblk, sn, &ro);
                LAUNCH(h, "residual", hipLaunchKernelGGL(res_kernel, rgrid, dim3(LK_RB), 0, st, fmap, h->pr, fl,
                                                         pts, n_pts, nb, parts, h->part_stride, ro, (size_t)0));
                if (h->wave_update)
                    LAUNCH(h, "update", hipLaunchKernelGGL(lk_update_wave_kernel, dim3(sn), dim3(LK_WAVE), 0, st, fl, parts,
                                                           nblk * (LK_RB / LK_WAVE), h->part_stride, t, h->d_Q, t_next, has_next ? 3 : 1));
                else
                    LAUNCH(h, "update", hipLaunchKernelGGL(lk_update_kernel, dim3(sn), dim3(LK_FB), 0, st, fl, parts,
                                                           nblk * (LK_RB / LK_WAVE), h->part_stride, t, h->d_Q, t_next, has_next ? 1 : 0));
            } else {
                if (k == 0) {
                    if (h->wave_update)
                        hipLaunchKernelGGL(lk_update_wave_kernel, dim3(sn), dim3(LK_WAVE), 0, st, fl, parts, 0, h->part_stride, 0.0, h->d_Q, t, 2);
                    else
                        hipLaunchKernelGGL(lk_predict_kernel, dim3(sn), dim3(LK_FB), 0, st, fl, h->d_Q, t);
                }
                const dim3 rgrid = batch_residual_grid(nblk, sn, &ro);
                hipLaunchKernelGGL(res_kernel, rgrid, dim3(LK_RB), 0, st, fmap, h->pr, fl, pts, n_pts, nb, parts,
                                   h->part_stride, ro, (size_t)0);
                if (h->wave_update)
                    hipLaunchKernelGGL(lk_update_wave_kernel, dim3(sn), dim3(LK_WAVE), 0, st, fl, parts, nblk * (LK_RB / LK_WAVE),
                                       h->part_stride, t, h->d_Q, t_next, has_next ? 3 : 1);
                else
                    hipLaunchKernelGGL(lk_update_kernel, dim3(sn), dim3(LK_FB), 0, st, fl, parts, nblk * (LK_RB / LK_WAVE), h->part_stride, t,
                                       h->d_Q, t_next, has_next ? 1 : 0);
            }
        }
    }
    HIPCHK(h, hipGetLastError());
    for (int g = 1; g < ngroups; ++g) {  // join: everything after this point on h->stream sees every group's results
        HIPCHK(h, hipEventRecord(h->ev_join[g - 1], streams[g]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[g - 1], 0));
    }
    if (out) {
        std::vector<lk_pose> tmp(n_scans);
        rc = fetch_poses(h, tmp.data(), S);
        if (rc) return rc;
        memcpy(out, tmp.data(), sizeof(lk_pose) * n_scans);
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return LK_OK;
}

// Config 2 at bandwidth size: the residual build of KILO.cc:122-210 - transform, covariance terms, voxel lookup, plane match with the one
// neighbour retry, observation row - for n_scans x n_pts points in ONE launch, scan s under the CURRENT state of filter slot s (no predict,
// no update, no insert), rows MATERIALISED in HBM: one 64-B record [h(6) z R] per point + the valid byte lk_residuals returns.
// 16 B in + 65 B out per point; the map side is the frozen-map grid of the batch replay (L2 / Infinity-Cache resident).
int lk_batch_residuals_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double* d_rows8, uint8_t* d_valid) {
    CHECK_H(h);
    if (!d_pts || !d_rows8 || !d_valid) return fail(h, LK_ERR_INVALID, "null argument");
    if (((uintptr_t)d_rows8 & 15u) != 0) return fail(h, LK_ERR_INVALID, "d_rows8 must be 16-byte aligned");
    if (n_scans == 0 || n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans must be in [1, n_slots]");
    if (n_pts == 0) return fail(h, LK_ERR_INVALID, "empty scans");
    if (n_pts > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "n_pts exceeds max_scan_points");
    LkMap fmap;
    int rc = frozen_map(h, &fmap);
    if (rc) return rc;
    rc = join_side_streams(h);
    if (rc) return rc;
    static const bool xid_enable = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const ResidualKernelFn k = !fmap.grid_on ? lk_residual_kernel<true, 0, false>
                                             : ((h->pr.ext_identity && xid_enable) ? lk_residual_kernel<true, 1, true> : lk_residual_kernel<true, 1, false>);
    ResidualOut ro;
    memset(&ro, 0, sizeof(ro));
    ro.rows8 = d_rows8, ro.valid = d_valid;
    const int nblk = (int)((n_pts + LK_RB - 1) / LK_RB);
    const dim3 rgrid = batch_residual_grid(nblk, (int)n_scans, &ro);
    LAUNCH(h, "residual_rows", hipLaunchKernelGGL(k, rgrid, dim3(LK_RB), 0, h->stream, fmap, h->pr, h->d_filters, d_pts, n_pts, (int)n_pts,
                                                  h->d_partials, h->part_stride, ro, n_pts));
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}

// The launches of a ragged batch once its tables (padded or CSR, LkRagged) are in HBM.  msg_kind: 0 none, 1 lk_imu, 2 lk_kin_imu.
// max_n: largest bucket of every bucket index (null: `biggest` for all of them).
static int ragged_launch(lk_handle* h, const lk_point* d_pts, size_t S, const LkRagged& rg, const double* d_tbegin, int biggest, size_t ldb,
                         const int* max_n, int msg_kind, lk_pose* out) {
    LkMap fmap;
    int rc = frozen_map(h, &fmap);
    if (rc) return rc;
    rc = zero_scan_counters(h, 0, (uint32_t)S);
    if (rc) return rc;
    hipStream_t st = h->stream;
    LkFilter* fl = h->d_filters;
    hipLaunchKernelGGL(lk_set_times_ragged_kernel, dim3(((int)S + 63) / 64), dim3(64), 0, st, fl, (int)S, d_tbegin);
    if (biggest <= LK_SCAN_WAVE_MAX && (msg_kind || !getenv("LEGKILO_RAGGED_LEVELS"))) {
        // small buckets only (a real scan's 2 ms bins): each scan's whole bucket chain as one wave, one launch
        if (msg_kind == 2)
            hipLaunchKernelGGL(lk_scan_wave_kin_kernel, dim3((unsigned)S), dim3(LK_WAVE), 0, st, fmap, h->pr, fl, d_pts, rg, h->d_Q);
        else if (msg_kind == 1)
            hipLaunchKernelGGL(lk_scan_wave_imu_kernel, dim3((unsigned)S), dim3(LK_WAVE), 0, st, fmap, h->pr, fl, d_pts, rg, h->d_Q);
        else
            hipLaunchKernelGGL(lk_scan_wave_kernel, dim3((unsigned)S), dim3(LK_WAVE), 0, st, fmap, h->pr, fl, d_pts, rg, h->d_Q);
#ifdef LK_DEBUG_PHASES
        {
            unsigned long long hb[16];
            hipStreamSynchronize(st);
            hipMemcpyFromSymbol(hb, HIP_SYMBOL(lk_sw_dbg), sizeof(hb));
            const char* names[6] = {"head", "predict", "bc", "residual", "update", "tail"};
            fprintf(stderr, "[scan-wave phases] %llu buckets:", hb[15]);
            for (int k = 0; k < 6; ++k) fprintf(stderr, " %s %.2f us;", names[k], (double)hb[k] / (double)hb[15] * 0.01);
            fprintf(stderr, "\n");
            memset(hb, 0, sizeof(hb));
            hipMemcpyToSymbol(HIP_SYMBOL(lk_sw_dbg), hb, sizeof(hb));
        }
#endif
        ldb = 0;
    } else {
        hipLaunchKernelGGL(lk_update_wave_ragged_kernel, dim3((unsigned)S), dim3(LK_WAVE), 0, st, fl, h->d_partials, h->part_stride, h->d_Q, rg, -1);
    }
    for (size_t b = 0; b < ldb; ++b) {
        const int nblk = ((max_n ? max_n[b] : biggest) + LK_RB - 1) / LK_RB;
        const auto rag_kernel = fmap.grid_on ? lk_residual_ragged_kernel<1> : lk_residual_ragged_kernel<0>;
        hipLaunchKernelGGL(rag_kernel, dim3(nblk, (unsigned)S), dim3(LK_RB), 0, st, fmap, h->pr, fl, d_pts, rg, (int)b, h->d_partials,
                           h->part_stride);
        hipLaunchKernelGGL(lk_update_wave_ragged_kernel, dim3((unsigned)S), dim3(LK_WAVE), 0, st, fl, h->d_partials, h->part_stride, h->d_Q, rg,
                           (int)b);
    }
    HIPCHK(h, hipGetLastError());
    if (out) {
        rc = fetch_poses(h, out, (int)S);
        if (rc) return rc;
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return LK_OK;
}

// Ragged batch replay: the scans of a recorded run differ in size, in their time buckets and in their start time.  One
// launch per bucket INDEX over all scans (grid sized by the largest bucket of that index; scans that have run out of
// buckets leave at once), every scan reading its own tables (LkRagged).  Same kernels' arithmetic as the uniform entry:
// a ragged batch of equally shaped scans gives the same bits.  Synchronous; priors as for lk_batch_replay_dev.
int ragged_replay(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                         const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                         const double* t_begin, const uint32_t* n_imu, const void* imus, size_t msg_bytes, lk_pose* out, bool with_insert) {
    CHECK_H(h);
    if (n_scans == 0 || n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans must be in [1, n_slots]");
    if (!d_pts || !scan_off || !n_buckets || !bucket_off || !bucket_dt || !t_begin) return fail(h, LK_ERR_INVALID, "null argument");
    const size_t S = n_scans;
    // pass 1: validate, count the non-empty buckets of every scan (empty ones are skipped, as in the uniform entry), pitch
    std::vector<uint32_t> cnt(S);
    size_t row_o = 0, row_t = 0, ldb = 0;
    uint32_t biggest_bucket = 0;
    for (size_t s = 0; s < S; ++s) {
        const uint32_t* bo = bucket_off + row_o;
        const size_t nbs = n_buckets[s];
        if (nbs == 0) return fail(h, LK_ERR_INVALID, "a scan has no buckets");
        if (bo[0] != 0 || scan_off[s] + bo[nbs] != scan_off[s + 1]) return fail(h, LK_ERR_INVALID, "bucket offsets do not tile the scan");
        uint32_t c = 0;
        const double* bt = bucket_dt + row_t;
        for (size_t b = 0; b < nbs; ++b) {
            // the same ordering rule the device-built tables enforce (lk_rag_flag_kernel): time stamps finite and non-decreasing
            if (!std::isfinite(bt[b]) || (b > 0 && bt[b] < bt[b - 1])) return fail(h, LK_ERR_INVALID, "bucket times must be finite and non-decreasing (KILO.cc:367-370 sorts the scan by time)");
            if (bo[b + 1] < bo[b]) return fail(h, LK_ERR_INVALID, "bucket offsets must be non-decreasing");
            if ((size_t)(bo[b + 1] - bo[b]) > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "bucket exceeds max_scan_points");
            c += bo[b + 1] > bo[b];
            biggest_bucket = std::max(biggest_bucket, bo[b + 1] - bo[b]);
        }
        if (c == 0) return fail(h, LK_ERR_INVALID, "empty scan");
        cnt[s] = c;
        ldb = std::max(ldb, (size_t)c);
        row_o += nbs + 1, row_t += nbs;
    }
    // every check comes before the first memset / upload / launch: a refused call leaves the filter slots untouched
    if (n_imu && biggest_bucket > (uint32_t)LK_SCAN_WAVE_MAX && !with_insert)   // (the batch with insert runs bucket by bucket: lk_rag_advance_kernel takes the messages at any bucket size)
        return fail(h, LK_ERR_INVALID, "IMU / kinematic messages between buckets are only replayed for scans whose buckets hold <= 512 points");
    // tables: pt_off [S][ldb+1] u64 | t [S][ldb] f64 | t_begin [S] f64 | nb [S] u32, staged in pinned host memory
    size_t n_imu_total = 0;
    if (n_imu)
        for (size_t s = 0; s < S; ++s) n_imu_total += n_imu[s];
    if (n_imu_total && !imus) return fail(h, LK_ERR_INVALID, "null message array");
    // ... | imu [n][7] f64 | nb [S] u32 | imu_off [S+1] u32
    const size_t o_po = 0, o_t = o_po + 8 * S * (ldb + 1), o_tb = o_t + 8 * S * ldb, o_im = o_tb + 8 * S, o_nb = o_im + msg_bytes * n_imu_total,
                 o_io = o_nb + 4 * S, bytes = o_io + 4 * (S + 1);
    if (bytes > h->rag_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_rag) hipFree(h->d_rag), h->d_rag = nullptr;
        if (h->h_rag) hipHostFree(h->h_rag), h->h_rag = nullptr;
        h->rag_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_rag, bytes + bytes / 2));
        HIPCHK(h, hipHostMalloc(&h->h_rag, bytes + bytes / 2, hipHostMallocDefault));
        h->rag_cap = bytes + bytes / 2;
    } else {
        HIPCHK(h, hipStreamSynchronize(h->stream));  // a previous call's upload from the staging buffer has completed
    }
    unsigned char* stage = static_cast<unsigned char*>(h->h_rag);
    auto* hpo = reinterpret_cast<unsigned long long*>(stage + o_po);
    auto* ht = reinterpret_cast<double*>(stage + o_t);
    auto* htb = reinterpret_cast<double*>(stage + o_tb);
    auto* hnb = reinterpret_cast<unsigned int*>(stage + o_nb);
    std::vector<int> max_n(ldb, 0);
    row_o = 0, row_t = 0;
    for (size_t s = 0; s < S; ++s) {   // pass 2: fill
        const uint32_t* bo = bucket_off + row_o;
        const double* bd = bucket_dt + row_t;
        const size_t nbs = n_buckets[s];
        unsigned long long* po = hpo + s * (ldb + 1);
        double* tr = ht + s * ldb;
        size_t k = 0;
        for (size_t b = 0; b < nbs; ++b) {
            if (bo[b + 1] == bo[b]) continue;
            po[k] = scan_off[s] + bo[b];
            tr[k] = t_begin[s] + bd[b];
            max_n[k] = std::max(max_n[k], (int)(bo[b + 1] - bo[b]));
            ++k;
        }
        for (size_t b = k; b <= ldb; ++b) po[b] = scan_off[s + 1];
        for (size_t b = k; b < ldb; ++b) tr[b] = 0.0;
        htb[s] = t_begin[s];
        hnb[s] = cnt[s];
        row_o += nbs + 1, row_t += nbs;
    }
    if (n_imu) {
        if (n_imu_total) memcpy(stage + o_im, imus, msg_bytes * n_imu_total);
        auto* hio = reinterpret_cast<unsigned int*>(stage + o_io);
        hio[0] = 0;
        for (size_t s = 0; s < S; ++s) hio[s + 1] = hio[s] + n_imu[s];
    }
    HIPCHK(h, hipMemcpyAsync(h->d_rag, stage, bytes, hipMemcpyHostToDevice, h->stream));
    unsigned char* dr = static_cast<unsigned char*>(h->d_rag);
    LkRagged rg;
    rg.pt_off = reinterpret_cast<const unsigned long long*>(dr + o_po);
    rg.t = reinterpret_cast<const double*>(dr + o_t);
    rg.nb = reinterpret_cast<const unsigned int*>(dr + o_nb);
    rg.ldb = (int)ldb;
    rg.imu_off = n_imu ? reinterpret_cast<const unsigned int*>(dr + o_io) : nullptr;
    rg.imu = reinterpret_cast<const double*>(dr + o_im);
    rg.msg_stride = (int)(msg_bytes / sizeof(double));
    rg.kin_noise = h->cfg.kin_meas_noise;
    rg.q_diag = h->q_diag ? 1 : 0;
    rg.acc_scale = h->cfg.gravity / h->acc_norm;
    imu_noise(h->cfg, rg.Rn);
    rg.bstart = nullptr;
    if (with_insert) {
        size_t max_scan_pts = 0;
        for (size_t s = 0; s < S; ++s) max_scan_pts = std::max(max_scan_pts, (size_t)(scan_off[s + 1] - scan_off[s]));
        return overlay_ragged_launch(h, d_pts, S, rg, reinterpret_cast<const double*>(dr + o_tb), (int)biggest_bucket, ldb, max_n.data(), max_scan_pts,
                                     n_imu ? (msg_bytes == sizeof(lk_kin_imu) ? 2 : 1) : 0, out);
    }
    return ragged_launch(h, d_pts, S, rg, reinterpret_cast<const double*>(dr + o_tb), (int)biggest_bucket, ldb, max_n.data(),
                         n_imu ? (msg_bytes == sizeof(lk_kin_imu) ? 2 : 1) : 0, out);
}

int lk_batch_replay_ragged_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                               const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                               const double* t_begin, lk_pose* out) {
    return ragged_replay(h, d_pts, n_scans, scan_off, n_buckets, bucket_off, bucket_dt, t_begin, nullptr, nullptr, sizeof(lk_imu), out);
}
// The same with each scan's IMU messages (only_imu_use mode, KILO.cc:379-383): n_imu[s] messages of scan s, concatenated in
// `imus`, time-sorted per scan; a message stamped before a bucket's time is applied before that bucket, the rest of the
// scan's messages are left unused exactly as the bucket loop of KILO::process leaves them.
int lk_batch_replay_ragged_imu_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                                   const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                                   const double* t_begin, const uint32_t* n_imu, const lk_imu* imus, lk_pose* out) {
    CHECK_H(h);
    if (!n_imu) return fail(h, LK_ERR_INVALID, "null argument");
    return ragged_replay(h, d_pts, n_scans, scan_off, n_buckets, bucket_off, bucket_dt, t_begin, n_imu, imus, sizeof(lk_imu), out);
}
// Leg-fusion mode (only_imu_use: false, the reference's default): n_kin[s] kinematic + IMU messages of scan s, concatenated in
// `kins`, time-sorted per scan; a message stamped before a bucket's time is applied (predictUpdateKinImu, KILO.cc:260-314:
// two predicts, 6 IMU rows + 3 rows per foot in contact, updateByKinImu eskf.cc:137-145) before that bucket, as the loop at
// KILO.cc:384-390 does.  Same bucket-size limit as the IMU entry.
int lk_batch_replay_ragged_kin_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off,
                                   const uint32_t* n_buckets, const uint32_t* bucket_off, const double* bucket_dt,
                                   const double* t_begin, const uint32_t* n_kin, const lk_kin_imu* kins, lk_pose* out) {
    CHECK_H(h);
    if (!n_kin) return fail(h, LK_ERR_INVALID, "null argument");
    return ragged_replay(h, d_pts, n_scans, scan_off, n_buckets, bucket_off, bucket_dt, t_begin, n_kin, kins, sizeof(lk_kin_imu), out);
}

// Recorded-run replay WITHOUT host-side bucket tables: the scans lie back to back in HBM (scan s = d_pts[scan_off[s] .. scan_off[s+1]),
// time-sorted), and the runs of equal curvature that KILO::process turns into buckets (KILO.cc:375-378) are found on the device
// (lk_rag_flag / rocPRIM exclusive scan / lk_rag_scatter: CSR tables in HBM); the host sends S + 1 offsets and S start times and
// reads back three integers.  msgs: n_msg[s] lk_imu (msg_kind 1) or lk_kin_imu (msg_kind 2) records per scan, concatenated, or
// msg_kind 0.  Same kernels, same results as lk_batch_replay_ragged(_imu / _kin)_dev with host-built tables.
int lk_batch_replay_scans_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off, const double* t_begin,
                              int msg_kind, const uint32_t* n_msg, const void* msgs, lk_pose* out) {
    CHECK_H(h);
    if (n_scans == 0 || n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans must be in [1, n_slots]");
    if (!d_pts || !scan_off || !t_begin) return fail(h, LK_ERR_INVALID, "null argument");
    if (msg_kind < 0 || msg_kind > 2 || (msg_kind && !n_msg)) return fail(h, LK_ERR_INVALID, "msg_kind must be 0 (none), 1 (lk_imu) or 2 (lk_kin_imu) with n_msg given");
    const size_t S = n_scans;
    if (scan_off[0] != 0) return fail(h, LK_ERR_INVALID, "scan_off[0] must be 0");
    for (size_t s = 0; s < S; ++s)
        if (scan_off[s + 1] <= scan_off[s]) return fail(h, LK_ERR_INVALID, "empty scan");
    const size_t n = scan_off[S];
    if (n >= ((size_t)1 << 32)) return fail(h, LK_ERR_CAPACITY, "more than 2^32 points in one batch");
    const size_t msg_bytes = msg_kind == 2 ? sizeof(lk_kin_imu) : sizeof(lk_imu);
    size_t n_msg_total = 0;
    if (msg_kind)
        for (size_t s = 0; s < S; ++s) n_msg_total += n_msg[s];
    if (n_msg_total && !msgs) return fail(h, LK_ERR_INVALID, "null message array");
    // device layout: scan_off u64[S+1] | t_begin f64[S] | pt_start u64[n+1] | tb f64[n] | msgs | flag u32[n] | rank u32[n] | bstart u32[S+1]
    //                | msg_off u32[S+1] | stats u32[4]
    const size_t o_so = 0, o_tb0 = o_so + 8 * (S + 1), o_ps = o_tb0 + 8 * S, o_tb = o_ps + 8 * (n + 1), o_ms = o_tb + 8 * n,
                 o_fl = o_ms + ((msg_bytes * n_msg_total + 7) & ~(size_t)7), o_rk = o_fl + 4 * n, o_bs = o_rk + 4 * n, o_mo = o_bs + 4 * (S + 1),
                 o_st = o_mo + 4 * (S + 1), bytes = o_st + 16;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (bytes > h->ragdev_cap) {
        if (h->d_ragdev) hipFree(h->d_ragdev), h->d_ragdev = nullptr, h->ragdev_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ragdev, bytes + bytes / 4));
        h->ragdev_cap = bytes + bytes / 4;
    }
    unsigned char* d = static_cast<unsigned char*>(h->d_ragdev);
    auto* d_so = reinterpret_cast<unsigned long long*>(d + o_so);
    auto* d_t0 = reinterpret_cast<double*>(d + o_tb0);
    auto* d_ps = reinterpret_cast<unsigned long long*>(d + o_ps);
    auto* d_tb = reinterpret_cast<double*>(d + o_tb);
    auto* d_fl = reinterpret_cast<unsigned int*>(d + o_fl);
    auto* d_rk = reinterpret_cast<unsigned int*>(d + o_rk);
    auto* d_bs = reinterpret_cast<unsigned int*>(d + o_bs);
    auto* d_mo = reinterpret_cast<unsigned int*>(d + o_mo);
    auto* d_st = reinterpret_cast<unsigned int*>(d + o_st);
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "scan offsets are 64-bit");
    HIPCHK(h, hipMemcpyAsync(d_so, scan_off, 8 * (S + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(d_t0, t_begin, 8 * S, hipMemcpyHostToDevice, h->stream));
    std::vector<unsigned int> moff;
    if (msg_kind) {
        moff.resize(S + 1, 0);
        for (size_t s = 0; s < S; ++s) moff[s + 1] = moff[s] + n_msg[s];
        HIPCHK(h, hipMemcpyAsync(d_mo, moff.data(), 4 * (S + 1), hipMemcpyHostToDevice, h->stream));
        if (n_msg_total) HIPCHK(h, hipMemcpyAsync(d + o_ms, msgs, msg_bytes * n_msg_total, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, hipMemsetAsync(d_st, 0, 16, h->stream));
    const unsigned int nblk = (unsigned int)((n + 255) / 256);
    hipLaunchKernelGGL(lk_rag_flag_kernel, dim3(nblk), dim3(256), 0, h->stream, d_pts, (unsigned long long)n, d_so, (int)S, d_fl, d_st);
    size_t tmp_bytes = 0;
    HIPCHK(h, lk_prim_exclusive_scan(nullptr, tmp_bytes, d_fl, d_rk, n, h->stream));
    if (tmp_bytes > h->ragtmp_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->d_ragtmp) hipFree(h->d_ragtmp), h->d_ragtmp = nullptr, h->ragtmp_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ragtmp, tmp_bytes));
        h->ragtmp_cap = tmp_bytes;
    }
    HIPCHK(h, lk_prim_exclusive_scan(h->d_ragtmp, tmp_bytes, d_fl, d_rk, n, h->stream));
    hipLaunchKernelGGL(lk_rag_scatter_kernel, dim3(nblk), dim3(256), 0, h->stream, d_pts, (unsigned long long)n, d_so, (int)S, d_fl, d_rk, d_t0,
                       d_ps, d_tb, d_bs, d_st);
    hipLaunchKernelGGL(lk_rag_stats_kernel, dim3(nblk), dim3(256), 0, h->stream, d_ps, d_bs, (int)S, d_st);
    HIPCHK(h, hipGetLastError());
    unsigned int st[4];
    HIPCHK(h, hipMemcpyAsync(st, d_st, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const unsigned int biggest = st[1], most = st[2];
    if (st[3]) return fail(h, LK_ERR_INVALID, "a scan is not sorted by time (curvature must be non-decreasing within a scan, and finite)");
    if (biggest > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "bucket exceeds max_scan_points");
    if (msg_kind && biggest > (unsigned int)LK_SCAN_WAVE_MAX)
        return fail(h, LK_ERR_INVALID, "IMU / kinematic messages between buckets are only replayed for scans whose buckets hold <= 512 points");
    LkRagged rg;
    rg.pt_off = d_ps;
    rg.t = d_tb;
    rg.nb = nullptr;
    rg.ldb = (int)most;
    rg.bstart = d_bs;
    rg.imu_off = msg_kind ? d_mo : nullptr;
    rg.imu = reinterpret_cast<const double*>(d + o_ms);
    rg.msg_stride = (int)(msg_bytes / sizeof(double));
    rg.kin_noise = h->cfg.kin_meas_noise;
    rg.q_diag = h->q_diag ? 1 : 0;
    rg.acc_scale = h->cfg.gravity / h->acc_norm;
    imu_noise(h->cfg, rg.Rn);
    return ragged_launch(h, d_pts, S, rg, d_t0, (int)biggest, (size_t)most, nullptr, msg_kind, out);
}

// Asynchronous, double-buffered batch replay.  The batch uses filter slots [first_slot, first_slot + n_scans); calls whose
// slot ranges alternate (first_slot = 0, n_scans, 0, ...) run on alternate HIP streams, so the single-workgroup update /
// predict kernels of one batch overlap the full-size residual launches of the next instead of sitting between them.
// Everything of a batch - arming the priors (d_x36 / d_P900 may be NULL: keep the slots' current state), the bucket
// chain, the pose gather and its copy to (pinned) host memory - is stream-ordered on the batch's own stream; nothing
// synchronises.  lk_synchronize() waits for all streams.
int lk_batch_replay_async_dev(lk_handle* h, const lk_point* d_pts, uint32_t first_slot, size_t n_scans, size_t n_pts, double t_begin,
                              const uint32_t* bucket_off, const double* bucket_dt, size_t n_buckets, const double* d_x36,
                              const double* d_P900, lk_pose* host_out) {
    CHECK_H(h);
    if (n_scans == 0 || (size_t)first_slot + n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "slot range must lie in [0, n_slots]");
    if (n_pts == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty scans");
    if (!d_pts || !bucket_off || !bucket_dt) return fail(h, LK_ERR_INVALID, "null argument");
    if ((d_x36 == nullptr) != (d_P900 == nullptr)) return fail(h, LK_ERR_INVALID, "give both prior buffers or neither");
    const int S = (int)n_scans;
    for (size_t b = 0; b < n_buckets; ++b)   // all checks before the first enqueue
        if (bucket_off[b + 1] > bucket_off[b] && (size_t)(bucket_off[b + 1] - bucket_off[b]) > h->map.max_scan)
            return fail(h, LK_ERR_CAPACITY, "bucket exceeds max_scan_points");
    LkMap fmap;
    {
        const int rc = frozen_map(h, &fmap);   // synchronises once per map snapshot (grid rebuild), otherwise free
        if (rc) return rc;
    }
    // batches on slot ranges 0, n, 2n, ... rotate over up to three streams (the handle's + two side streams): with three batches in
    // flight there is (almost) always a residual launch ready while the other two sit in their update / predict launches
    // The stream is a function of (first_slot, n_scans): two batches in flight on overlapping slots are ordered only if they
    // share it.  So first_slot must be a multiple of n_scans (slot ranges of one size never overlap partially), and when the batch
    // size changes while batches may still be in flight the streams are drained first.
    if (first_slot % (uint32_t)n_scans != 0) return fail(h, LK_ERR_INVALID, "first_slot must be a multiple of n_scans");
    if (h->async_n != 0 && h->async_n != n_scans) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)
            if (h->side[i]) HIPCHK(h, hipStreamSynchronize(h->side[i]));
    }
    h->async_n = n_scans;
    const uint32_t ring = (first_slot / (uint32_t)n_scans) % 3u;
    hipStream_t st = ring == 0 ? h->stream : h->side[ring - 1];
    LkFilter* fl = h->d_filters + first_slot;
    double* parts = h->d_partials + (size_t)first_slot * h->part_stride;
    if (st != h->stream && !d_x36) {
        // the slots' state was armed elsewhere (lk_batch_set_priors_dev is asynchronous on the MAIN stream): order this batch
        // after it.  With d_x36 / d_P900 the batch arms its own slots on its own stream and needs no such edge - which is
        // what keeps two batches in flight.
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(st, h->ev_fork, 0));
    }
    if (d_x36) {
        HIPCHK(h, hipMemcpy2DAsync(fl[0].x, sizeof(LkFilter), d_x36, sizeof(double) * 36, sizeof(double) * 36, n_scans, hipMemcpyDeviceToDevice, st));
        HIPCHK(h, hipMemcpy2DAsync(fl[0].P, sizeof(LkFilter), d_P900, sizeof(double) * 900, sizeof(double) * 900, n_scans, hipMemcpyDeviceToDevice, st));
    }
    HIPCHK(h, hipMemset2DAsync(&fl[0].n_effect, sizeof(LkFilter), 0, 24, n_scans, st));
    hipLaunchKernelGGL(lk_set_times_kernel, dim3((S + 63) / 64), dim3(64), 0, st, fl, S, t_begin);
    ResidualOut ro;
    memset(&ro, 0, sizeof(ro));
    {
        const int rc = batch_ordered(h, d_pts, first_slot, n_scans, n_pts, bucket_off, n_buckets, st, &ro);   // first sight of a batch synchronises; afterwards one small launch
        if (rc) return rc;
    }
    const lk_point* const alt_base = ro.alt_pts;
    bool first = true;
    for (size_t b = 0; b < n_buckets; ++b) {
        if (bucket_off[b + 1] <= bucket_off[b]) continue;
        const int nb = (int)(bucket_off[b + 1] - bucket_off[b]);
        ro.alt_pts = alt_base ? alt_base + bucket_off[b] : nullptr;
        size_t nx = b + 1;
        while (nx < n_buckets && bucket_off[nx + 1] <= bucket_off[nx]) ++nx;
        const bool has_next = nx < n_buckets;
        const double t = t_begin + bucket_dt[b], t_next = has_next ? t_begin + bucket_dt[nx] : 0.0;
        const int nblk = (nb + LK_RB - 1) / LK_RB;
        if (first) {
            if (h->wave_update)
                hipLaunchKernelGGL(lk_update_wave_kernel, dim3(S), dim3(LK_WAVE), 0, st, fl, parts, 0, h->part_stride, 0.0, h->d_Q, t, 2);
            else
                hipLaunchKernelGGL(lk_predict_kernel, dim3(S), dim3(LK_FB), 0, st, fl, h->d_Q, t);
        }
        first = false;
        const auto res_kernel = batch_residual_kernel(h, fmap);
        const dim3 rgrid = batch_residual_grid(nblk, S, &ro);
        hipLaunchKernelGGL(res_kernel, rgrid, dim3(LK_RB), 0, st, fmap, h->pr, fl, d_pts + bucket_off[b], n_pts, nb, parts,
                           h->part_stride, ro, (size_t)0);
        if (h->wave_update)
            hipLaunchKernelGGL(lk_update_wave_kernel, dim3(S), dim3(LK_WAVE), 0, st, fl, parts, nblk * (LK_RB / LK_WAVE), h->part_stride, t,
                               h->d_Q, t_next, has_next ? 3 : 1);
        else
            hipLaunchKernelGGL(lk_update_kernel, dim3(S), dim3(LK_FB), 0, st, fl, parts, nblk * (LK_RB / LK_WAVE), h->part_stride, t, h->d_Q,
                               t_next, has_next ? 1 : 0);
    }
    HIPCHK(h, hipGetLastError());
    if (host_out) {
        hipLaunchKernelGGL(lk_pose_gather_kernel, dim3((S + 63) / 64), dim3(64), 0, st, fl, h->d_poses + first_slot, S);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(host_out, h->d_poses + first_slot, sizeof(lk_pose) * n_scans, hipMemcpyDeviceToHost, st));
    }
    return LK_OK;
}

// ------------------------------------------------------------------ measurement hooks
int lk_profile_enable(lk_handle* h, int on) {
    CHECK_H(h);
    h->profiling = on != 0;
    return LK_OK;
}
int lk_profile_get(lk_handle* h, const char* kernel, uint64_t* launches, double* total_ms) {
    CHECK_H(h);
    auto it = h->prof.find(kernel ? kernel : "");
    if (it == h->prof.end()) {
        if (launches) *launches = 0;
        if (total_ms) *total_ms = 0.0;
        return LK_OK;
    }
    if (launches) *launches = it->second.launches;
    if (total_ms) *total_ms = it->second.total_ms;
    return LK_OK;
}
int lk_profile_reset(lk_handle* h) {
    CHECK_H(h);
    h->prof.clear();
    return LK_OK;
}
int lk_device_malloc(lk_handle* h, void** d_ptr, size_t bytes) {
    CHECK_H(h);
    HIPCHK(h, hipMalloc(d_ptr, bytes));
    return LK_OK;
}
int lk_device_free(lk_handle* h, void* d_ptr) {
    CHECK_H(h);
    HIPCHK(h, hipFree(d_ptr));
    return LK_OK;
}
int lk_memcpy_h2d(lk_handle* h, void* d_dst, const void* src, size_t bytes) {
    CHECK_H(h);
    HIPCHK(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_memcpy_d2h(lk_handle* h, void* dst, const void* d_src, size_t bytes) {
    CHECK_H(h);
    HIPCHK(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
int lk_synchronize(lk_handle* h) {
    CHECK_H(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < lk_handle::kMaxGroups - 1; ++i)
        if (h->side[i]) HIPCHK(h, hipStreamSynchronize(h->side[i]));  // double-buffered async batches live there
    h->async_n = 0;
    return LK_OK;
}
int lk_stream_pipeline(lk_handle* h, int on) {
    CHECK_H(h);   // joins the inserts in flight
    h->spec_enable = on != 0;
    return LK_OK;
}
int lk_stream_resident(lk_handle* h, int on) {
    CHECK_H(h);
    h->resident_enable = on != 0;
    return LK_OK;
}
int lk_stream_grid(lk_handle* h, int on) {
    CHECK_H(h);
    h->gridscan_mode = std::min(std::max(on, 0), 2);
    return LK_OK;
}
int lk_test_stall(lk_handle* h, unsigned int bound_ms) {
    CHECK_H(h);
    h->test_stall_ms = bound_ms & 0x7fffffffu;
    return LK_OK;
}
int lk_stream_resident_stats(lk_handle* h, uint64_t* out2) {
    CHECK_H(h);
    if (!out2) return fail(h, LK_ERR_INVALID, "out2 is null");
    out2[0] = h->resident_scans + h->grid_scans, out2[1] = h->resident_relaunches + h->grid_relaunches;
    return LK_OK;
}
int lk_stream_stats(lk_handle* h, uint64_t* out4) {
    CHECK_H(h);
    if (!out4) return fail(h, LK_ERR_INVALID, "out4 is null");
#ifdef LK_DEBUG_LI
    {
        unsigned long long hb[64];
        hipStreamSynchronize(h->stream);
        hipMemcpyFromSymbol(hb, HIP_SYMBOL(lk_li_dbg), sizeof(hb));
        fprintf(stderr, "[li] %llu mismatches, %llu traced\n", hb[0], hb[60]);
        for (int k = 0; k < 4 && k < (int)hb[60]; ++k) {
            const unsigned long long* o = hb + 1 + 14 * k;
            fprintf(stderr, "[tr] leaf %d root %d Tn %d Tp %d To %d g %d r.npts %d r.block %d layer %d newp %d state %llx consumed %d need_init %d off %d\n", (int)o[0], (int)o[1],
                    (int)o[2], (int)o[3], (int)o[4], (int)o[5], (int)o[6], (int)o[7], (int)o[8], (int)o[9], o[10], (int)o[11], (int)o[12], (int)o[13]);
        }
        for (int k = 0; k < 4 && k < (int)hb[0]; ++k) {
            const unsigned long long* o = hb + 1 + 14 * k;
            fprintf(stderr, "[li] leaf %lld root %lld npts %d/%d newp %d/%d block %d/%d layer %lld/%lld state %llx/%llx plane %lld off %lld\n", (long long)o[0], (long long)o[1],
                    (int)o[2], (int)o[3], (int)o[4], (int)o[5], (int)o[6], (int)o[7], (long long)o[8], (long long)o[9], o[10], o[11], (long long)o[12], (long long)o[13]);
        }
    }
#endif
#ifdef LK_DEBUG_INS
    {
        unsigned long long hb[16];
        unsigned int ctr[LK_CTR_COUNT];
        hipStreamSynchronize(h->stream);
        hipMemcpyFromSymbol(hb, HIP_SYMBOL(lk_ins_dbg), sizeof(hb));
        hipMemcpy(ctr, h->map.counters, sizeof(ctr), hipMemcpyDeviceToHost);
        const char* names[8] = {"desc+node", "points", "simulate", "stores", "eigen", "plane_var", "commit", "tail"};
        fprintf(stderr, "[ins] %llu groups, %llu fitted; per group (us):", hb[15], hb[14]);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.2f;", names[k], (double)hb[k] / (double)(hb[15] ? hb[15] : 1) * 0.01);
        fprintf(stderr, " block-load wait %.2f;", (double)hb[8] / (double)(hb[15] ? hb[15] : 1) * 0.01);
        { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(lk_ins_dbg), z, sizeof(z)); }
        fprintf(stderr, "\n[ins] plane_var: %.0f shader cycles and %.2f us per fit -> %.2f GHz", (double)hb[12] / (double)(hb[14] ? hb[14] : 1),
                (double)hb[13] / (double)(hb[14] ? hb[14] : 1) * 0.01, (double)hb[12] / ((double)hb[13] * 10.0 + 1e-9));
        fprintf(stderr, "\n[ins] last bucket: touched %u heavy %u groups %u gidx %u fallback %u\n", ctr[LK_CTR_TOUCHED], ctr[LK_CTR_HEAVY],
                ctr[LK_CTR_GROUPS], ctr[LK_CTR_GIDX], ctr[LK_CTR_FALLBACK]);
        unsigned int rh[6 * 32];
        hipMemcpyFromSymbol(rh, HIP_SYMBOL(lk_root_hist), sizeof(rh));
        const char* rows[6] = {"light", "one group", "several groups", "long", "wave before 2nd root", "load+sort+walk"};
        for (int r = 0; r < 6; ++r) {
            fprintf(stderr, "[root] %-22s 2-us bins:", rows[r]);
            for (int b = 0; b < 32; ++b) fprintf(stderr, " %u", rh[r * 32 + b]);
            fprintf(stderr, "\n");
        }
        memset(rh, 0, sizeof(rh));
        hipMemcpyToSymbol(HIP_SYMBOL(lk_root_hist), rh, sizeof(rh));
        {
            unsigned long long sd[32];
            hipMemcpyFromSymbol(sd, HIP_SYMBOL(lk_slow_dbg), sizeof(sd));
            const char* pn[10] = {"desc+node", "points", "simulate", "stores", "eigen", "plane_var", "commit", "tail", "block-load", "-"};
            for (int w = 0; w < 2; ++w) {
                fprintf(stderr, "[apply] plane-fit groups %s 20 us: %llu; per group (us):", w == 0 ? ">=" : "<", sd[16 * w + 15]);
                for (int k = 0; k < 9; ++k) fprintf(stderr, " %s %.2f;", pn[k], (double)sd[16 * w + k] / (double)(sd[16 * w + 15] ? sd[16 * w + 15] : 1) * 0.01);
                fprintf(stderr, "\n");
            }
            memset(sd, 0, sizeof(sd));
            hipMemcpyToSymbol(HIP_SYMBOL(lk_slow_dbg), sd, sizeof(sd));
        }
        unsigned int eh[16 * 32];
        hipMemcpyFromSymbol(eh, HIP_SYMBOL(lk_ev_hist), sizeof(eh));
        for (int r = 0; r < 16; ++r) {
            fprintf(stderr, "[apply] flags %2d (1 new child, 2 new block, 4 frozen, 8 plane fit) 2-us bins:", r);
            for (int b = 0; b < 32; ++b) fprintf(stderr, " %u", eh[r * 32 + b]);
            fprintf(stderr, "\n");
        }
        memset(eh, 0, sizeof(eh));
        hipMemcpyToSymbol(HIP_SYMBOL(lk_ev_hist), eh, sizeof(eh));
    }
#endif
    unsigned int redo[3] = {0, 0, 0};   // LK_CTR_SPEC_REDO, [13], LK_CTR_RES_REDO
    HIPCHK(h, hipMemcpyAsync(redo, h->map.counters + LK_CTR_SPEC_REDO, sizeof(redo), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (redo[0] < h->spec_redo_seen) h->spec_redo_seen = 0;          // the device words are reset with the pools (map import)
    h->spec_redo_total += (uint64_t)(redo[0] - h->spec_redo_seen);
    h->spec_redo_seen = redo[0];
    if (redo[2] < h->res_redo_seen) h->res_redo_seen = 0;
    h->res_redo_total += (uint64_t)(redo[2] - h->res_redo_seen);
    h->res_redo_seen = redo[2];
    out4[0] = h->spec_buckets, out4[1] = h->spec_tiles, out4[2] = h->spec_redo_total, out4[3] = h->res_redo_total;
    return LK_OK;
}
int lk_stream_grid_placement(lk_handle* h, uint32_t* xcc_mask) {
    CHECK_H(h);
    if (!xcc_mask) return fail(h, LK_ERR_INVALID, "xcc_mask is null");
    HIPCHK(h, hipMemcpyAsync(xcc_mask, h->map.counters + LK_CTR_GRID_XCC, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return LK_OK;
}
void* lk_stream(lk_handle* h) { return h ? (void*)h->stream : nullptr; }

}  // extern "C"
