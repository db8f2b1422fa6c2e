// lk_overlay.hip - the overlay translation unit of liblegkilo_hip.so (see lk_internal.h): batch replay WITH the map insert - every scan on its own
// copy-on-write overlay of the handle's map (lk_overlay_kernels.h) - and the entry points that launch it.
#define LK_TU_OVERLAY 1
#include "lk_internal.h"

static int ov_root_bits() {   // LEGKILO_OV_ROOT_BITS=0: no "takes the point at the root" bit for the re-projection pass (A/B)
    static const int v = getenv("LEGKILO_OV_ROOT_BITS") == nullptr || atoi(getenv("LEGKILO_OV_ROOT_BITS")) != 0;
    return v;
}

// Round 6: the FRONT of a bucket index of the ragged batch with insert as ONE launch when every bucket holds <= LK_SCAN_WAVE_MAX points (a recorded
// scan's 2 ms bins): lk_rag_advance_kernel (messages + predict), lk_ov_residual_kernel, lk_update_wave_ragged_kernel, lk_ov_begin_kernel and
// lk_ov_reproject_kernel were five one-wave-per-scan launches, each paying a launch boundary (~5 us for 1 024 one-wave workgroups whatever they do) and
// its own load / store of the filter's 7.6 KB.  Here one wave per scan runs the five bodies back to back - the one-wave filter cores and the tile code
// of dev_scan_wave, the overlay lookup of lk_ov_residual_kernel - with state and covariance in LDS from the first message to the update.  Same device
// functions, same order of sums (tile totals in tile order, as lk_update_wave_kernel adds up to eight of them): bit-identical to the five launches
// (test_batch_replay_overlay_ragged compares both against the oracle; LEGKILO_RAG_FUSE=0 is the A/B).
extern "C++" {
template <bool XID>
__global__ void __launch_bounds__(LK_WAVE, 2)
    lk_rag_ov_front_kernel(LkMap base, LkOverlay ov, LkParams pr, LkFilter* filters, const double* __restrict__ Q, LkRagged rg, const lk_point* __restrict__ d_pts,
                           int b, int msg_kind) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const LkMap pm = ov_slot_map(ov, (unsigned int)slot);
    if (b >= rag_nb(rg, slot)) {   // this scan has run out of buckets: the passes behind this one must find its work lists empty (lk_ov_begin_kernel ran for every slot)
        dev_bucket_begin_wave(pm);
        return;
    }
    LkFilter* f = &filters[slot];
    const double* T = rag_t(rg, slot);
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    __syncthreads();
    const double tb = T[b];
    if (msg_kind) {   // lk_rag_advance_kernel: the scan's messages stamped before this bucket that no earlier bucket has consumed (KILO.cc:379-390)
        const size_t mstride = msg_kind == 2 ? 33 : 7;
        const unsigned int q0 = rg.imu_off[slot], q1 = rg.imu_off[slot + 1];
        for (unsigned int q = q0; q < q1; ++q) {
            const double* m = rg.imu + mstride * (size_t)q;
            const double tm = m[0];
            if (!(tm < tb)) break;
            if (b > 0 && tm < T[b - 1]) continue;
            wave_predict_core(sm, Q, tm - t_upd, tm - t_pred, lane, rg.q_diag != 0);
            t_pred = tm;
            if (msg_kind == 2) wave_kin_update_core(sm, rows, m, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else wave_imu_update_core(sm, m + 1, m + 4, rg.acc_scale, rg.Rn, lane);
            t_upd = tm;
        }
    }
    wave_predict_core(sm, Q, tb - t_upd, tb - t_pred, lane, rg.q_diag != 0);   // KILO.cc:111-115
    t_pred = tb;
    // lk_ov_residual_kernel: the bucket's tiles against base map + the scan's overlay, under the predicted state in LDS
    const unsigned long long* po = rag_pt_off(rg, slot);
    const lk_point* pts = d_pts + po[b];
    const int n = (int)(po[b + 1] - po[b]);
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
    LkOvView ovv;
    ovv.keys = ov.keys + (size_t)slot * ov.hash_cap;
    ovv.hash_mask = ov.hash_cap - 1;
    ovv.match = ov.match + (size_t)slot * ov.nodes_cap;
    ovv.nodes = ov.nodes + (size_t)slot * ov.nodes_cap;
    ovv.bits = ov.bits + (size_t)slot * ov.bit_words;
    ResidualOut ro;
    ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = nullptr, ro.ids = nullptr;
    double totv = 0.0;   // tot[j] in lanes 0..31
    for (int i0 = 0; i0 < n; i0 += LK_WAVE) {
        __builtin_amdgcn_wave_barrier();   // the previous tile's reads of the rows are complete
        const double a = residual_tile<false, 3, XID, true, false>(base, pr, bc, reinterpret_cast<const float4*>(pts), i0 + lane, n, rows, lane, ro, (size_t)0, &ovv);
        totv += (lane < 29) ? a : 0.0;
    }
    // lk_update_wave_ragged_kernel (update_only): the posterior the insert reads
    const int N = (int)(lane_bcast<28>(totv) + 0.5);
    if (lane == 0) {
        f->last_predict_t = t_pred;
        f->n_buckets += 1;
        f->last_N = N;
        f->updated = N > 0;
        if (N > 0) {
            f->n_updates += 1;
            f->n_effect += (unsigned long long)N;
        }
        f->last_update_t = N > 0 ? tb : t_upd;   // KILO.cc:212
    }
    if (N > 0) wave_update_core(sm, totv, N, lane);
    __syncthreads();
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    if (lane < 36) f->x[lane] = sm.x[lane];
    // the re-projection below reads the posterior through the filter record, like every other kernel of the insert - written by THIS wave: its stores
    // have to be acknowledged before its loads go out (workgroup scope = s_waitcnt; an agent-scope fence here is an L2 write-back + invalidate per
    // wave, 1 024 of them per launch: the first version of this kernel was 6 ms SLOWER than the five launches for it)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    // lk_ov_begin_kernel, lk_ov_reproject_kernel
    dev_bucket_begin_wave(pm);
    for (int i = lane; i < n; i += LK_WAVE) {
        const int r = ov_reproject_point(base, ov, pr, filters, pts, i, (unsigned int)slot);
        ov.ptroot[(size_t)slot * ov.scan_cap + i] = r;   // for lk_ov_point_geom_kernel
    }
}

}   // extern "C++"

extern "C" {

int lk_overlay_export(lk_handle* h, uint32_t slot, void* blob, size_t* bytes) {
    CHECK_H(h);
    if (!h->ov.counters || !h->ov_last_slots) return fail(h, LK_ERR_STATE, "no overlay replay's pools are held by this handle (none has run, or lk_overlay_reserve released them)");
    if (slot >= h->ov_last_slots) return fail(h, LK_ERR_INVALID, "slot was not part of the last overlay replay");
    // an overlay is not self-contained: split leaves keep their first points in the BASE map's blocks, lazily copied voxels their plane in the base
    // map's plane records.  Once the handle's map has changed (lk_process_scan, lk_map_update, lk_map_slide, lk_map_import, ...) those ids may
    // name other voxels: refuse instead of exporting them
    if (!h->grid_valid || h->ov_gen != h->map_gen)
        return fail(h, LK_ERR_STATE, "the handle's map has changed since the overlay replay: its overlays can no longer be exported (export before the map is updated, slid or imported, or replay again)");
    const LkOverlay& ov = h->ov;
    // leaves the fast root pass left split (old points still in the handle's blocks) are made whole first
    hipLaunchKernelGGL(lk_ov_merge_split_kernel, dim3(64), dim3(LK_MB), 0, h->stream, h->map, ov, slot);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> keys(ov.hash_cap);
    HIPCHK(h, hipMemcpy(keys.data(), ov.keys + (size_t)slot * ov.hash_cap, sizeof(unsigned long long) * ov.hash_cap, hipMemcpyDeviceToHost));
    std::vector<int4> table(ov.hash_cap);
    for (unsigned int i = 0; i < ov.hash_cap; ++i) {
        if (keys[i] == LK_OV_EMPTY) {
            table[i] = make_int4(INT_MIN, INT_MIN, INT_MIN, LK_EMPTY);
        } else {
            int k3[3];
            ov_unpack_key(keys[i], k3);
            table[i] = make_int4(k3[0], k3[1], k3[2], (int)i);
        }
    }
    LkMap m = ov_slot_map(ov, slot);
    DevTemps tmp;
    HIPCHK(h, tmp.alloc(&m.hash, sizeof(int4) * ov.hash_cap));
    HIPCHK(h, hipMemcpy(m.hash, table.data(), sizeof(int4) * ov.hash_cap, hipMemcpyHostToDevice));
    return export_map_blob(h, m, ov.hash_cap, blob, bytes);
}

// ------------------------------------------------------------------ batch replay with a per-scan insert overlay
// (lk_overlay_kernels.h) KILO::process for every scan of the batch - predict, residual, update AND map insert per bucket (KILO.cc:108-233,
// :375-395) - each scan on its own copy-on-write overlay of the handle's map, which itself stays untouched.
void ov_free(lk_handle* h) {
    LkOverlay& o = h->ov;
    void* ptrs[] = {o.keys, o.planes, o.match, o.nodes, o.blocks, o.counters, o.touched, o.next, o.scratch, o.gidx, o.groups, o.slots,
                    o.free_list, o.freed_next, o.dirty, o.newroot, o.spec, o.bits, o.jobs, o.jobhdr, o.frozen, o.sums, o.base_sums, o.cplx, o.ptroot};
    for (void* q : ptrs)
        if (q) hipFree(q);
    memset(&o, 0, sizeof(o));
    h->ov_slots = 0;
    h->ov_pool_bytes = 0;
    h->ov_last_slots = 0;   // nothing of the last replay is left to export / count (lk_overlay_export, lk_overlay_stats)
}
// LEGKILO_POISON_POOLS (test aid): node records that look plausible - a few points, no children - and point at a block far outside any pool
__global__ void __launch_bounds__(256) lk_ov_poison_nodes_kernel(lk_node_rec* nodes, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    nodes[i].npts = 3, nodes[i].new_points = 1, nodes[i].block = 0x3fffff00, nodes[i].layer = 0, nodes[i].state = LK_NODE_UPDATE_ENABLE | LK_NODE_INIT_OCTO;
    for (int c = 0; c < 8; ++c) nodes[i].child[c] = -1;
}
// Per-scan capacities.  lk_overlay_reserve's numbers if given; else, when an earlier replay of scans of this size has left its
// high-water marks, those + 25 % (pools more than twice that are released and re-made: round 4 reserved n_pts / 6 roots = 110 MB per scan,
// 113 GB for 1 024 scans, where the bench's scans use 4 700 roots); else a first guess of n_pts / 18 roots.  `grow` (bits of the slots' error
// word: 1 private root table, 2 nodes, 4 point blocks) doubles what overflowed - the replay is then run again (lk_batch_replay_overlay_dev).
static int ov_reserve(lk_handle* h, uint32_t S, size_t n_pts_scan, size_t biggest_bucket, const LkMap& fmap, unsigned int grow = 0) {
    const bool hist = h->ov_hw_roots > 0 && h->ov_hw_npts == n_pts_scan;
    uint32_t roots, nodes_extra, blocks;
    if (h->ov_want_roots) {
        roots = h->ov_want_roots;
        nodes_extra = h->ov_want_nodes ? std::max(h->ov_want_nodes, roots + 64u) - roots : roots / 2;
        blocks = h->ov_want_blocks ? h->ov_want_blocks : roots;
    } else if (hist) {
        roots = h->ov_hw_roots + h->ov_hw_roots / 4 + 64;
        const uint32_t child = h->ov_hw_nodes > h->ov_hw_roots ? h->ov_hw_nodes - h->ov_hw_roots : 0u;
        nodes_extra = child + child / 4 + 256;
        blocks = h->ov_hw_blocks + h->ov_hw_blocks / 4 + 64;
    } else {
        roots = (uint32_t)std::min<size_t>(std::max<size_t>(2048, n_pts_scan / 18), std::max<size_t>(1024, n_pts_scan));
        nodes_extra = roots / 2;
        blocks = roots;
    }
    uint32_t hash_cap = next_pow2(roots + roots / 4);       // the roots' records ARE the table entries: node ids [0, hash_cap); load <= 0.8 (0.57 for the bench's scans)
    LkOverlay& o = h->ov;
    if (grow) {   // never below what is there; what overflowed is doubled
        hash_cap = std::max(hash_cap, o.hash_cap), nodes_extra = std::max(nodes_extra, o.nodes_cap - o.hash_cap), blocks = std::max(blocks, o.blocks_cap);
        if (grow & LK_E_HASH_FULL) hash_cap *= 2;
        if (grow & LK_E_NODES_FULL) nodes_extra = nodes_extra * 2 + 256;
        if (grow & LK_E_BLOCKS_FULL) blocks *= 2;
    }
    const uint32_t nodes_cap = hash_cap + nodes_extra;        // children from hash_cap upwards
    const uint32_t scan_cap = (uint32_t)((biggest_bucket + 63) & ~(size_t)63);
    const size_t cells = (size_t)fmap.gdim[0] * (size_t)fmap.gdim[1] * (size_t)fmap.gdim[2];
    const uint32_t bit_words = (uint32_t)((cells + 31) / 32);
    const bool fits = S <= h->ov_slots && hash_cap <= o.hash_cap && nodes_cap - hash_cap <= o.nodes_cap - o.hash_cap && blocks <= o.blocks_cap &&
                      scan_cap <= o.scan_cap && bit_words <= o.bit_words;
    // far too large for what the scans use (measured by an earlier replay, or asked for explicitly): released and re-made
    const bool oversized = (hist || h->ov_want_roots) && !grow && (o.hash_cap > 2 * hash_cap || (size_t)o.blocks_cap > 2 * (size_t)blocks + 1024);
    if (fits && !oversized) return LK_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const uint32_t S2 = oversized ? S : std::max(S, h->ov_slots);
    LkOverlay n = {};
    if (oversized) {
        n.hash_cap = hash_cap, n.nodes_cap = nodes_cap, n.blocks_cap = blocks, n.scan_cap = scan_cap, n.bit_words = bit_words;
    } else {
        n.hash_cap = std::max(hash_cap, o.hash_cap);
        n.nodes_cap = n.hash_cap + std::max(nodes_extra, o.nodes_cap > o.hash_cap ? o.nodes_cap - o.hash_cap : 0u);
        n.blocks_cap = std::max(blocks, o.blocks_cap), n.scan_cap = std::max(scan_cap, o.scan_cap), n.bit_words = std::max(bit_words, o.bit_words);
    }
    ov_free(h);
    h->ov = n;
    const size_t s = S2;
    size_t total = 0;
    auto get = [&](auto** q, size_t bytes) -> hipError_t {
        total += bytes;
        return hipMalloc((void**)q, bytes);
    };
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = get(&o.keys, s * n.hash_cap * sizeof(unsigned long long));
    if (e == hipSuccess) e = get(&o.planes, s * n.nodes_cap * sizeof(lk_plane_rec));
    if (e == hipSuccess) e = get(&o.match, s * n.nodes_cap * sizeof(lk_match_rec));
    if (e == hipSuccess) e = get(&o.nodes, s * n.nodes_cap * sizeof(lk_node_rec));
    if (e == hipSuccess) e = get(&o.blocks, s * n.blocks_cap * sizeof(lk_block_rec));
    if (e == hipSuccess) e = get(&o.counters, s * LK_CTR_COUNT * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.touched, s * n.scan_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.next, s * n.scan_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.scratch, s * n.scan_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.gidx, s * n.scan_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.groups, s * n.scan_cap * 2 * sizeof(LkGroup));
    if (e == hipSuccess) e = get(&o.slots, s * n.hash_cap * LK_SLOTS * sizeof(float4));
    if (e == hipSuccess) e = get(&o.free_list, s * n.blocks_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.freed_next, s * n.blocks_cap * sizeof(int));
    if (e == hipSuccess) e = get(&o.dirty, s * n.hash_cap * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.newroot, (size_t)(LK_NEWROOT_MASK + 1) * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.spec, LK_SPEC_WORDS * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.bits, s * n.bit_words * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.frozen, (size_t)2 * n.bit_words * sizeof(unsigned int));
    if (e == hipSuccess) e = get(&o.jobs, s * n.hash_cap * LK_INLINE_GROUPS * sizeof(LkFitJob));
    if (e == hipSuccess) e = get(&o.jobhdr, s * n.hash_cap * LK_INLINE_GROUPS * sizeof(int4));
    if (e == hipSuccess) e = get(&o.sums, s * n.hash_cap * sizeof(LkLeafSum));
    if (e == hipSuccess) e = get(&o.base_sums, (size_t)h->map.max_nodes * sizeof(LkLeafSum));
    if (e == hipSuccess) e = get(&o.cplx, s * n.scan_cap * 2 * sizeof(int));
    if (e == hipSuccess) e = get(&o.ptroot, s * n.scan_cap * sizeof(int));
    if (e == hipSuccess && !h->d_ov_status) e = hipMalloc(&h->d_ov_status, 8 * sizeof(unsigned int));
    if (e == hipSuccess && getenv("LEGKILO_POISON_POOLS")) {
        // test aid: fresh pools hold 0x5a bytes instead of whatever the allocator hands out (usually zeros) - a kernel that trusts a record
        // nobody has written then faults HERE AND NOW, not in the one process whose allocation history leaves garbage there
        hipLaunchKernelGGL(lk_ov_poison_nodes_kernel, dim3((unsigned int)((s * n.nodes_cap + 255) / 256)), dim3(256), 0, h->stream, o.nodes, s * n.nodes_cap);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ov_free(h);
        char buf[256];
        snprintf(buf, sizeof(buf), "overlay pools for %u scans (%u root entries / %u child nodes / %u point blocks each) do not fit: %s (lk_overlay_reserve sets smaller per-scan capacities)",
                 S2, n.hash_cap, n.nodes_cap - n.hash_cap, n.blocks_cap, hipGetErrorString(e));
        return fail(h, LK_ERR_CAPACITY, buf);
    }
    h->ov_slots = S2;
    h->ov_pool_bytes = total;
    hipLaunchKernelGGL(lk_ov_init_kernel, dim3((n.hash_cap + 255) / 256, S2), dim3(256), 0, h->stream, h->ov);
    HIPCHK(h, hipGetLastError());
    return LK_OK;
}

int lk_overlay_reserve(lk_handle* h, uint32_t roots_per_scan, uint32_t nodes_per_scan, uint32_t blocks_per_scan) {
    CHECK_H(h);
    if ((roots_per_scan && roots_per_scan < 16) || (nodes_per_scan && nodes_per_scan < roots_per_scan) || (blocks_per_scan && blocks_per_scan < 16))
        return fail(h, LK_ERR_INVALID, "overlay capacities too small (0 = derive from the scan size)");
    h->ov_want_roots = roots_per_scan, h->ov_want_nodes = nodes_per_scan, h->ov_want_blocks = blocks_per_scan;
    if (h->ov_slots) {   // pools of another shape are released; the next replay allocates what it needs
        HIPCHK(h, hipStreamSynchronize(h->stream));
        ov_free(h);
    }
    return LK_OK;
}

}  // extern "C"
// the overlay pools as a group of slots starting at slot s0 sees them: every per-slot array advanced by s0 slots (the kernels index by blockIdx.y)
static LkOverlay ov_at(const LkOverlay& o, size_t s0) {
    LkOverlay r = o;
    r.keys += s0 * o.hash_cap;
    r.planes += s0 * o.nodes_cap, r.match += s0 * o.nodes_cap, r.nodes += s0 * o.nodes_cap;
    r.blocks += s0 * o.blocks_cap;
    r.counters += s0 * LK_CTR_COUNT;
    r.touched += s0 * o.scan_cap, r.next += s0 * o.scan_cap, r.scratch += s0 * o.scan_cap, r.gidx += s0 * o.scan_cap;
    r.groups += s0 * o.scan_cap * 32;
    r.slots += s0 * o.hash_cap * LK_SLOTS * 4;
    r.free_list += s0 * o.blocks_cap, r.freed_next += s0 * o.blocks_cap;
    r.dirty += s0 * o.hash_cap;
    r.bits += s0 * o.bit_words;
    r.jobs += s0 * o.hash_cap * LK_INLINE_GROUPS;
    r.jobhdr += s0 * o.hash_cap * LK_INLINE_GROUPS;
    r.sums += s0 * o.hash_cap;
    r.cplx += s0 * o.scan_cap * 2;
    r.ptroot += s0 * o.scan_cap;
    return r;   // frozen, base_sums, newroot, spec: shared by all slots
}
extern "C" {
int lk_batch_replay_overlay_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, size_t n_pts, double t_begin, const uint32_t* bucket_off,
                                const double* bucket_dt, size_t n_buckets, lk_pose* out) {
    CHECK_H(h);
    if (n_scans == 0 || n_scans > h->cfg.n_slots) return fail(h, LK_ERR_INVALID, "n_scans must be in [1, n_slots]");
    if (n_pts == 0 || n_buckets == 0) return fail(h, LK_ERR_INVALID, "empty scans");
    if (!d_pts || !bucket_off || !bucket_dt) return fail(h, LK_ERR_INVALID, "null argument");
    const int S = (int)n_scans;
    std::vector<size_t> live;
    size_t biggest = 0;
    for (size_t b = 0; b < n_buckets; ++b) {
        if (bucket_off[b + 1] < bucket_off[b] || bucket_off[b + 1] > n_pts) return fail(h, LK_ERR_INVALID, "bucket offsets must be non-decreasing and end inside the scan");
        if (!std::isfinite(bucket_dt[b]) || (b > 0 && bucket_dt[b] < bucket_dt[b - 1])) return fail(h, LK_ERR_INVALID, "bucket times must be finite and non-decreasing");
        if (bucket_off[b + 1] == bucket_off[b]) continue;
        if ((size_t)(bucket_off[b + 1] - bucket_off[b]) > h->map.max_scan) return fail(h, LK_ERR_CAPACITY, "bucket exceeds max_scan_points");
        biggest = std::max(biggest, (size_t)(bucket_off[b + 1] - bucket_off[b]));
        live.push_back(b);
    }
    if (live.empty()) return fail(h, LK_ERR_INVALID, "empty scans");
    int rc = join_side_streams(h);   // an asynchronous frozen-map batch may still be using the filter slots
    if (rc) return rc;
    LkMap fmap;
    rc = frozen_map(h, &fmap);
    if (rc) return rc;
    if (!fmap.grid_on) return fail(h, LK_ERR_STATE, "overlay replay needs the frozen-map grid (root keys' bounding box too large, LEGKILO_GRID=0, or out of device memory)");
    rc = ov_reserve(h, (uint32_t)S, n_pts, biggest, fmap);
    if (rc) return rc;
    // the batch's priors, kept for a second attempt: a scan whose overlay outgrows pools that were sized by this library (first guess, or
    // the previous replay's high-water marks) makes the pools grow and the whole batch run again - only capacities the caller has set
    // explicitly (lk_overlay_reserve) fail with LK_ERR_CAPACITY
    if (h->ov_priors_cap < (size_t)S) {
        if (h->d_ov_priors) hipFree(h->d_ov_priors), h->d_ov_priors = nullptr, h->ov_priors_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ov_priors, sizeof(LkFilter) * (size_t)S));
        h->ov_priors_cap = (size_t)S;
    }
    HIPCHK(h, hipMemcpyAsync(h->d_ov_priors, h->d_filters, sizeof(LkFilter) * (size_t)S, hipMemcpyDeviceToDevice, h->stream));
    unsigned int stt[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int attempt = 0;; ++attempt) {
    const LkOverlay ov = h->ov;
    hipStream_t st = h->stream;
    rc = zero_scan_counters(h, 0, (uint32_t)S);
    if (rc) return rc;
    hipLaunchKernelGGL(lk_set_times_kernel, dim3((S + 63) / 64), dim3(64), 0, st, h->d_filters, S, t_begin);
    HIPCHK(h, hipMemsetAsync(ov.frozen, 0, (size_t)2 * ov.bit_words * sizeof(unsigned int), st));
    static const bool frozen_bits = getenv("LEGKILO_OV_FROZEN_BITS") == nullptr || atoi(getenv("LEGKILO_OV_FROZEN_BITS")) != 0;   // 0: every point through the probes and the walk (A/B)
    if (frozen_bits) LAUNCH(h, "ov_frozen_bits", hipLaunchKernelGGL(lk_ov_frozen_bits_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, st, fmap, h->hash_cap, h->pr.max_layer, ov.frozen, ov_root_bits()));
    static const bool xid_enable = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const auto res_kernel = (h->pr.ext_identity && xid_enable) ? lk_ov_residual_kernel<true> : lk_ov_residual_kernel<false>;
    // root pass: the fast path (lk_ov_point_geom_kernel + lk_ov_root_lane_kernel: root leaves that append / refit / freeze) and the generic pass over what it leaves
    // (LEGKILO_OV_FAST=0: the generic pass over every touched root, round 4's path; A/B)
    // LEGKILO_OV_FAST: 1 (default) = the fast path - one thread per point (geometry) + one lane per root (lk_ov_point_geom_kernel,
    // lk_ov_root_lane_kernel), the generic pass for what they leave; 0 = the generic pass over every touched root (round 4's path; A/B)
    static const bool ov_fast = getenv("LEGKILO_OV_FAST") == nullptr || atoi(getenv("LEGKILO_OV_FAST")) != 0;
    static const int root_waves = getenv("LEGKILO_OV_ROOT_WAVES") ? atoi(getenv("LEGKILO_OV_ROOT_WAVES")) : 3;   // generic pass without the fit: 184 VGPRs at 2 waves, 168 at 3
    const auto root_kernel = ov_fast ? (root_waves >= 4 ? lk_ov_insert_root_kernel<4, true> : root_waves == 3 ? lk_ov_insert_root_kernel<3, true> : lk_ov_insert_root_kernel<2, true>)
                                     : (root_waves >= 4 ? lk_ov_insert_root_kernel<4, false> : root_waves == 3 ? lk_ov_insert_root_kernel<3, false> : lk_ov_insert_root_kernel<2, false>);
    if (ov_fast) LAUNCH(h, "ov_base_sums", hipLaunchKernelGGL(lk_ov_base_sums_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, st, fmap, h->hash_cap, ov.base_sums));
    static const int ov_mat_wg = getenv("LEGKILO_OV_MAT_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_MAT_WG"))) : 0;
    static const int ov_root_wg = getenv("LEGKILO_OV_ROOT_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_ROOT_WG"))) : 0;
    static const int fit_blocks = getenv("LEGKILO_OV_FIT_BLOCKS") ? std::max(1, atoi(getenv("LEGKILO_OV_FIT_BLOCKS"))) : 12;   // round 6: with the job headers in a dense array 6 / 8 / 12 / 16 / 24 waves per scan -> fit pass 1.75 / 1.92 / 1.61 / 2.12 / 1.71 ms (before: best at 6, 1.89)
    static const int ov_waves_per_slot = getenv("LEGKILO_OV_WG_PER_SLOT") ? std::max(1, atoi(getenv("LEGKILO_OV_WG_PER_SLOT"))) : 0;
    // Slot groups on separate HIP streams (LEGKILO_OV_GROUPS, default 4 - round 6, same box: 15.14 / 13.48 / 13.02 / 12.74 ms with 1 / 2 / 3 / 4 groups, 14.8 / 14.2 with
    // 6 / 8: beyond four streams the queues share hardware): the scans are independent, and the passes of a bucket are of two
    // kinds - the root pass issues VALU work at 2.8 TB/s of HBM traffic, the others (re-projection, copy-on-write, plane fits) only move
    // bytes - so one group's root pass runs beside the other group's memory passes.  A group is the same launches with every per-slot
    // array offset to its first slot (ov_at).  Profiling mode (per-launch events + sync) and small batches stay on one stream.
    static const int ov_groups_env = getenv("LEGKILO_OV_GROUPS") ? std::min(std::max(atoi(getenv("LEGKILO_OV_GROUPS")), 1), (int)lk_handle::kMaxGroups) : 4;
    const int ngroups = (!h->profiling && S >= 64 * ov_groups_env) ? ov_groups_env : 1;
    hipStream_t streams[lk_handle::kMaxGroups];
    streams[0] = h->stream;
    for (int g = 1; g < lk_handle::kMaxGroups; ++g) streams[g] = h->side[g - 1];
    if (ngroups > 1) {
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        for (int g = 1; g < ngroups; ++g) HIPCHK(h, hipStreamWaitEvent(streams[g], h->ev_fork, 0));
    }
    const LkOverlay ov_all = ov;
    // the enqueue of every group's launches; whatever it returns, the side streams are joined below before this call returns (a failed
    // launch must not leave them writing filters, partials and pools behind the caller's back)
    auto enqueue_all = [&]() -> int {
    for (int grp = 0; grp < ngroups; ++grp) {
        const int s0 = (int)((long)S * grp / ngroups), sn = (int)((long)S * (grp + 1) / ngroups) - s0;
        const unsigned int per = std::max(std::max(ov_all.hash_cap, ov_all.bit_words), (unsigned int)LK_CTR_COUNT);   // root records, bitmap words, counters
        LAUNCH(h, "ov_reset", hipLaunchKernelGGL(lk_ov_reset_kernel, dim3((per + 255) / 256, sn), dim3(256), 0, streams[grp], ov_at(ov_all, (size_t)s0)));
    }
    for (size_t k = 0; k < live.size(); ++k)
    for (int grp = 0; grp < ngroups; ++grp) {
        const int s0 = (int)((long)S * grp / ngroups), Sg = (int)((long)S * (grp + 1) / ngroups) - s0;   // this group's slots
        const LkOverlay ov = ov_at(ov_all, (size_t)s0);
        hipStream_t st = streams[grp];
        LkFilter* fl = h->d_filters + s0;
        double* parts = h->d_partials + (size_t)s0 * h->part_stride;
        const size_t b = live[k];
        const int nb = (int)(bucket_off[b + 1] - bucket_off[b]);
        const double t = t_begin + bucket_dt[b];
        const int nblk = (nb + LK_RB - 1) / LK_RB;
        const lk_point* pts = d_pts + (size_t)s0 * n_pts + bucket_off[b];
        const LkPtSrc src = {pts, n_pts, nb, nullptr, nullptr, 0, 0, nullptr};
        if (k == 0) LAUNCH(h, "predict", hipLaunchKernelGGL(lk_update_wave_kernel, dim3(Sg), dim3(LK_WAVE), 0, st, fl, parts, 0, h->part_stride, 0.0, h->d_Q, t, 2));
        LAUNCH(h, "ov_residual", hipLaunchKernelGGL(res_kernel, dim3(nblk, Sg), dim3(LK_RB), 0, st, fmap, ov, h->pr, fl, src, parts, h->part_stride));
        LAUNCH(h, "update", hipLaunchKernelGGL(lk_update_wave_kernel, dim3(Sg), dim3(LK_WAVE), 0, st, fl, parts, nblk * (LK_RB / LK_WAVE), h->part_stride, t, h->d_Q, 0.0, 1));
        // the bucket's insert into every slot's overlay, from the posterior (KILO.cc:216-233)
        LAUNCH(h, "ov_begin", hipLaunchKernelGGL(lk_ov_begin_kernel, dim3(Sg), dim3(LK_WAVE), 0, st, ov));
        LAUNCH(h, "ov_reproject", hipLaunchKernelGGL(lk_ov_reproject_kernel, dim3((nb + LK_WAVE - 1) / LK_WAVE, Sg), dim3(LK_WAVE), 0, st, fmap, ov, h->pr, fl, src));
        // per-root passes: enough waves per slot to cover its touched roots a few at a time, ~4096 workgroups per launch at least
        const int per_slot = ov_waves_per_slot ? ov_waves_per_slot : std::max(1, std::min((nb + 255) / 256, std::max(2, (4096 + S - 1) / S)));
        // (measured at 1024 slots x 20 000-point buckets, workgroups per slot: copy-on-write 2.8 / 6.6 / 12.2 ms per batch at 4 / 16 / 32 - a wave takes 64
        // roots, more waves only find nothing to do; root pass 12.8 / 11.0 / 11.7 - a wave works through its roots one after the other)
        const int mat_per_slot = ov_mat_wg ? ov_mat_wg : std::max(1, per_slot / 2), root_per_slot = ov_root_wg ? ov_root_wg : 3 * per_slot;
        LAUNCH(h, "ov_materialise", hipLaunchKernelGGL(ov_fast ? lk_ov_materialise_kernel<true> : lk_ov_materialise_kernel<false>, dim3(mat_per_slot, Sg), dim3(LK_MB), 0, st, fmap, ov, h->pr));
        // one WAVE per touched root (the leaf's plane fit only decided), then the fits one LANE each
        if (ov_fast) {
            LAUNCH(h, "ov_point_geom", hipLaunchKernelGGL(lk_ov_point_geom_kernel, dim3((nb + 255) / 256, Sg), dim3(256), 0, st, ov, h->pr, fl, src));
            static const int lane_blocks = getenv("LEGKILO_OV_LANE_BLOCKS") ? std::max(1, atoi(getenv("LEGKILO_OV_LANE_BLOCKS"))) : 0;
            LAUNCH(h, "ov_root_lane", hipLaunchKernelGGL(lk_ov_root_lane_kernel, dim3(lane_blocks ? lane_blocks : std::max(4, (nb + 16 * LK_WAVE - 1) / (16 * LK_WAVE)), Sg), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
            LAUNCH(h, "ov_insert_root", hipLaunchKernelGGL(root_kernel, dim3(std::max(1, per_slot / 2), Sg), dim3(LK_MB), 0, st, fmap, ov, h->pr, fl, src));
        } else {
            LAUNCH(h, "ov_insert_root", hipLaunchKernelGGL(root_kernel, dim3(root_per_slot, Sg), dim3(LK_MB), 0, st, fmap, ov, h->pr, fl, src));
        }
        LAUNCH(h, "ov_fit_eig", hipLaunchKernelGGL(lk_ov_fit_eig_kernel, dim3(fit_blocks, Sg), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
        static const bool fit_group = getenv("LEGKILO_OV_FIT_GROUP") == nullptr || atoi(getenv("LEGKILO_OV_FIT_GROUP")) != 0;   // 0: round 5's one lane per fit (A/B)
        if (fit_group)
            LAUNCH(h, "ov_fit_lane", hipLaunchKernelGGL(lk_ov_fit_group_kernel, dim3(fit_blocks, Sg), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
        else
            LAUNCH(h, "ov_fit_lane", hipLaunchKernelGGL(lk_ov_fit_lane_kernel, dim3(fit_blocks, Sg), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
        static const int ov_apply_wg = getenv("LEGKILO_OV_APPLY_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_APPLY_WG"))) : 0;
        static const int ov_fb_wg = getenv("LEGKILO_OV_FB_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_FB_WG"))) : 0;
        LAUNCH(h, "ov_insert_apply", hipLaunchKernelGGL(lk_ov_insert_apply_kernel, dim3(ov_apply_wg ? ov_apply_wg : per_slot, Sg), dim3(LK_MB), 0, st, ov, h->pr, fl, src));
        LAUNCH(h, "ov_insert_fallback", hipLaunchKernelGGL(lk_ov_insert_fallback_kernel, dim3(std::min(Sg, ov_fb_wg ? ov_fb_wg : 128)), dim3(LK_MB), 0, st, ov, h->pr, fl, src, Sg));   // (1 024 slots, workgroups 8 / 32 / 128 / 256 / 512: 0.54 / 0.26 / 0.15 / 0.17 / 0.16 ms per batch; a workgroup or more per slot: 0.34)
        if (k + 1 < live.size())
            LAUNCH(h, "predict", hipLaunchKernelGGL(lk_update_wave_kernel, dim3(Sg), dim3(LK_WAVE), 0, st, fl, parts, 0, h->part_stride, 0.0, h->d_Q,
                                                    t_begin + bucket_dt[live[k + 1]], 2));
    }
    HIPCHK(h, hipGetLastError());
    return LK_OK;
    };
    rc = enqueue_all();
    for (int g = 1; g < ngroups; ++g) {  // join: everything after this point on h->stream sees every group's results
        if (rc) {
            (void)hipStreamSynchronize(streams[g]);   // error path: nothing of this call keeps running
            continue;
        }
        HIPCHK(h, hipEventRecord(h->ev_join[g - 1], streams[g]));
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join[g - 1], 0));
    }
    if (rc) {
        (void)hipStreamSynchronize(h->stream);
        return rc;
    }
    h->ov_last_slots = (uint32_t)S, h->ov_gen = h->map_gen;
    const unsigned int init[8] = {0u, 0u, 0u, 0u, 0xffffffffu, 0u, 0u, 0u};
    HIPCHK(h, hipMemcpyAsync(h->d_ov_status, init, sizeof(init), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(lk_ov_status_kernel, dim3(std::min((S + 255) / 256, 64)), dim3(256), 0, st, ov, (unsigned int)S, h->d_ov_status);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(stt, h->d_ov_status, sizeof(stt), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    const bool growable = !h->ov_want_roots && !(stt[0] & ~(LK_E_HASH_FULL | LK_E_NODES_FULL | LK_E_BLOCKS_FULL)) && attempt < 4;
    if (!stt[0] || !growable) break;
    HIPCHK(h, hipMemcpyAsync(h->d_filters, h->d_ov_priors, sizeof(LkFilter) * (size_t)S, hipMemcpyDeviceToDevice, st));
    rc = ov_reserve(h, (uint32_t)S, n_pts, biggest, fmap, stt[0]);
    if (rc) return rc;
    }   // attempts
    const LkOverlay& ov = h->ov;
    hipStream_t st = h->stream;
    if (!stt[0]) h->ov_hw_roots = stt[3], h->ov_hw_nodes = stt[1], h->ov_hw_blocks = stt[2], h->ov_hw_npts = n_pts;
    if (out) {
        std::vector<lk_pose> tmp(n_scans);
        rc = fetch_poses(h, tmp.data(), S);   // synchronises
        if (rc) return rc;
        memcpy(out, tmp.data(), sizeof(lk_pose) * n_scans);
    } else {
        HIPCHK(h, hipStreamSynchronize(st));
    }
    if (stt[0] & LK_E_KEY_RANGE) {
        char buf[200];
        snprintf(buf, sizeof(buf), "overlay replay: a point of slot %u lies in a voxel whose key is outside the +-2^20 range of the private root tables' packed keys (%.0f km from the origin at this voxel size)",
                 stt[4], 1048576.0 * h->cfg.max_voxel_size / 1000.0);
        return fail(h, LK_ERR_INVALID, buf);
    }
    if (stt[0]) {
        char buf[256];
        snprintf(buf, sizeof(buf), "overlay pool overflow in slot %u (bits 0x%x: 1 private root table, 2 nodes, 4 point blocks, 8 work lists); largest use over the slots: %u nodes, %u blocks, %u roots; per-scan pools: %u root entries, %u child nodes, %u blocks (lk_overlay_reserve)",
                 stt[4], stt[0], stt[1], stt[2], stt[3], ov.hash_cap, ov.nodes_cap - ov.hash_cap, ov.blocks_cap);
        return fail(h, LK_ERR_CAPACITY, buf);
    }
    return LK_OK;
}

int lk_batch_replay_overlay_ragged_dev(lk_handle* h, const lk_point* d_pts, size_t n_scans, const uint64_t* scan_off, const uint32_t* n_buckets,
                                       const uint32_t* bucket_off, const double* bucket_dt, const double* t_begin, const uint32_t* n_msg, const void* msgs,
                                       int msg_kind, lk_pose* out) {
    CHECK_H(h);
    if (msg_kind < 0 || msg_kind > 2) return fail(h, LK_ERR_INVALID, "msg_kind must be 0 (no messages), 1 (lk_imu) or 2 (lk_kin_imu)");
    if (msg_kind && !n_msg) return fail(h, LK_ERR_INVALID, "null argument");
    return ragged_replay(h, d_pts, n_scans, scan_off, n_buckets, bucket_off, bucket_dt, t_begin, msg_kind ? n_msg : nullptr, msgs,
                         msg_kind == 2 ? sizeof(lk_kin_imu) : sizeof(lk_imu), out, true);
}
}  // extern "C"
// The ragged batch WITH insert, bucket INDEX after bucket index over all scans (one launch of every pass per index, grids sized by that
// index's longest bucket; a scan that has run out of buckets leaves every launch at once): per index b - the scan's messages up to the
// bucket's time + predict (lk_rag_advance_kernel), residual with the overlay lookup, update, then the insert passes of
// lk_batch_replay_overlay_dev on each scan's own bucket (LkPtSrc).  One stream: a recorded run's buckets are small, the launches are what it costs.
int overlay_ragged_launch(lk_handle* h, const lk_point* d_pts, size_t S_, const LkRagged& rg, const double* d_tbegin, int biggest, size_t ldb,
                                 const int* max_n, size_t max_scan_pts, int msg_kind, lk_pose* out) {
    const int S = (int)S_;
    int rc = join_side_streams(h);
    if (rc) return rc;
    LkMap fmap;
    rc = frozen_map(h, &fmap);
    if (rc) return rc;
    if (!fmap.grid_on) return fail(h, LK_ERR_STATE, "overlay replay needs the frozen-map grid (root keys' bounding box too large, LEGKILO_GRID=0, or out of device memory)");
    rc = ov_reserve(h, (uint32_t)S, max_scan_pts, (size_t)biggest, fmap);
    if (rc) return rc;
    if (h->ov_priors_cap < (size_t)S) {
        if (h->d_ov_priors) hipFree(h->d_ov_priors), h->d_ov_priors = nullptr, h->ov_priors_cap = 0;
        HIPCHK(h, hipMalloc(&h->d_ov_priors, sizeof(LkFilter) * (size_t)S));
        h->ov_priors_cap = (size_t)S;
    }
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_ov_priors, h->d_filters, sizeof(LkFilter) * (size_t)S, hipMemcpyDeviceToDevice, st));
    static const bool xid_enable = getenv("LEGKILO_XID") == nullptr || atoi(getenv("LEGKILO_XID")) != 0;
    const auto res_kernel = (h->pr.ext_identity && xid_enable) ? lk_ov_residual_kernel<true> : lk_ov_residual_kernel<false>;
    unsigned int stt[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    for (int attempt = 0;; ++attempt) {
        const LkOverlay ov = h->ov;
        LkFilter* fl = h->d_filters;
        rc = zero_scan_counters(h, 0, (uint32_t)S);
        if (rc) return rc;
        hipLaunchKernelGGL(lk_set_times_ragged_kernel, dim3((S + 63) / 64), dim3(64), 0, st, fl, S, d_tbegin);
        HIPCHK(h, hipMemsetAsync(ov.frozen, 0, (size_t)2 * ov.bit_words * sizeof(unsigned int), st));
        LAUNCH(h, "ov_frozen_bits", hipLaunchKernelGGL(lk_ov_frozen_bits_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, st, fmap, h->hash_cap, h->pr.max_layer, ov.frozen, ov_root_bits()));
        LAUNCH(h, "ov_base_sums", hipLaunchKernelGGL(lk_ov_base_sums_kernel, dim3((h->hash_cap + 255) / 256), dim3(256), 0, st, fmap, h->hash_cap, ov.base_sums));
        {
            const unsigned int per = std::max(std::max(ov.hash_cap, ov.bit_words), (unsigned int)LK_CTR_COUNT);
            LAUNCH(h, "ov_reset", hipLaunchKernelGGL(lk_ov_reset_kernel, dim3((per + 255) / 256, S), dim3(256), 0, st, ov));
        }
        const bool rag_resident = getenv("LEGKILO_RAG_RESIDENT") == nullptr || atoi(getenv("LEGKILO_RAG_RESIDENT")) != 0;   // 0: launch by launch (A/B, and the bit-identity reference of the tests: read at every call)
        h->ov_res_rounds = 0;
        const bool resident = rag_resident && biggest <= LK_SCAN_WAVE_MAX && !rg.bstart;
        if (resident) {
            // [S] next bucket of every scan, [S] the bucket whose fallback items wait, one counter: scans stopped by fallback items in the last launch
            if (h->ov_res_cap < (size_t)S) {
                if (h->d_ov_res) hipFree(h->d_ov_res), h->d_ov_res = nullptr, h->ov_res_cap = 0;
                HIPCHK(h, hipMalloc(&h->d_ov_res, sizeof(int) * (2 * (size_t)S + 4)));
                h->ov_res_cap = (size_t)S;
            }
            int* cur = h->d_ov_res;
            int* fb_b = cur + S;
            unsigned int* pending = reinterpret_cast<unsigned int*>(cur + 2 * (size_t)S);
            HIPCHK(h, hipMemsetAsync(cur, 0, sizeof(int) * (2 * (size_t)S + 4), st));
            const LkPtSrc fsrc = {d_pts, 0, 0, rg.pt_off, rg.nb, rg.ldb, 0, fb_b};
            static const int ov_fb_wg_s = getenv("LEGKILO_OV_FB_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_FB_WG"))) : 0;
            unsigned int rounds = 0;
            for (;; ++rounds) {
                rc = ov_scan_launch(h, h->pr.ext_identity && xid_enable, S, st, fmap, ov, fl, rg, d_pts, msg_kind, cur, fb_b, pending);   // lk_ovscan.hip
                if (rc) return rc;
                unsigned int n_pending = 0;
                HIPCHK(h, hipMemcpyAsync(&n_pending, pending, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                HIPCHK(h, hipStreamSynchronize(st));
                if (!n_pending) break;
                if (rounds > ldb + 1) return fail(h, LK_ERR_STATE, "scan-resident overlay replay: more fallback rounds than buckets");
                HIPCHK(h, hipMemsetAsync(pending, 0, sizeof(unsigned int), st));
                LAUNCH(h, "ov_insert_fallback", hipLaunchKernelGGL(lk_ov_insert_fallback_kernel, dim3(std::min((int)S, ov_fb_wg_s ? ov_fb_wg_s : 128)), dim3(LK_MB), 0, st, ov, h->pr, fl, fsrc, (int)S));
            }
            h->ov_res_rounds = rounds + 1;
            if (getenv("LEGKILO_RAG_VERBOSE")) fprintf(stderr, "[legkilo] scan-resident overlay replay: %u launches of the scan kernel for %d scans x <= %zu buckets\n", rounds + 1, S, ldb);
        }
        for (size_t b = 0; b < (resident ? 0 : ldb); ++b) {
            const int nb = std::max(1, max_n ? max_n[b] : biggest);
            const int nblk = (nb + LK_RB - 1) / LK_RB;
            const LkPtSrc src = {d_pts, 0, 0, rg.pt_off, rg.nb, rg.ldb, (int)b, nullptr};
            static const bool rag_fuse = getenv("LEGKILO_RAG_FUSE") == nullptr || atoi(getenv("LEGKILO_RAG_FUSE")) != 0;   // 0: the five launches (A/B, and the bit-identity reference)
            if (rag_fuse && biggest <= LK_SCAN_WAVE_MAX) {
                const auto front = (h->pr.ext_identity && xid_enable) ? lk_rag_ov_front_kernel<true> : lk_rag_ov_front_kernel<false>;
                LAUNCH(h, "rag_ov_front", hipLaunchKernelGGL(front, dim3(S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr, fl, h->d_Q, rg, d_pts, (int)b, msg_kind));
            } else {
                LAUNCH(h, "rag_advance", hipLaunchKernelGGL(lk_rag_advance_kernel, dim3(S), dim3(LK_WAVE), 0, st, fl, h->d_Q, rg, (int)b, msg_kind));
                LAUNCH(h, "ov_residual", hipLaunchKernelGGL(res_kernel, dim3(nblk, S), dim3(LK_RB), 0, st, fmap, ov, h->pr, fl, src, h->d_partials, h->part_stride));
                LAUNCH(h, "update", hipLaunchKernelGGL(lk_update_wave_ragged_kernel, dim3(S), dim3(LK_WAVE), 0, st, fl, h->d_partials, h->part_stride, h->d_Q, rg, (int)b, 1));
                LAUNCH(h, "ov_begin", hipLaunchKernelGGL(lk_ov_begin_kernel, dim3(S), dim3(LK_WAVE), 0, st, ov));
                LAUNCH(h, "ov_reproject", hipLaunchKernelGGL(lk_ov_reproject_kernel, dim3((nb + LK_WAVE - 1) / LK_WAVE, S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr, fl, src));
            }
            const int per_slot = std::max(1, std::min((nb + 255) / 256, std::max(2, (4096 + S - 1) / S)));
            if (rag_fuse && biggest <= LK_SCAN_WAVE_MAX) {
                static const int mid_threads = getenv("LEGKILO_RAG_MID_THREADS") ? (atoi(getenv("LEGKILO_RAG_MID_THREADS")) <= 64 ? 64 : atoi(getenv("LEGKILO_RAG_MID_THREADS")) <= 128 ? 128 : 256) : LK_MB;
                LAUNCH(h, "ov_mid", hipLaunchKernelGGL(lk_ov_mid_kernel<true>, dim3(S), dim3(mid_threads), 0, st, fmap, ov, h->pr, fl, src));
            } else {
                LAUNCH(h, "ov_materialise", hipLaunchKernelGGL(lk_ov_materialise_kernel<true>, dim3(std::max(1, per_slot / 2), S), dim3(LK_MB), 0, st, fmap, ov, h->pr));
                LAUNCH(h, "ov_point_geom", hipLaunchKernelGGL(lk_ov_point_geom_kernel, dim3((nb + 255) / 256, S), dim3(256), 0, st, ov, h->pr, fl, src));
                LAUNCH(h, "ov_root_lane", hipLaunchKernelGGL(lk_ov_root_lane_kernel, dim3(std::max(1, (nb + 16 * LK_WAVE - 1) / (16 * LK_WAVE)), S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
            }
            static const bool fit_group_r = getenv("LEGKILO_OV_FIT_GROUP") == nullptr || atoi(getenv("LEGKILO_OV_FIT_GROUP")) != 0;
            static const bool rag_tail = getenv("LEGKILO_RAG_TAIL") == nullptr || atoi(getenv("LEGKILO_RAG_TAIL")) != 0;   // 0: the four launches (A/B)
            if (rag_fuse && rag_tail && fit_group_r && biggest <= LK_SCAN_WAVE_MAX) {
                static const int tail_threads = getenv("LEGKILO_RAG_TAIL_THREADS") ? std::min(LK_MB, std::max(64, atoi(getenv("LEGKILO_RAG_TAIL_THREADS")) & ~63)) : LK_WAVE;   // one wave per slot: 29.2 -> 26.4 ms against four (fewer waves to dispatch)
                LAUNCH(h, "ov_tail", hipLaunchKernelGGL(lk_ov_tail_kernel, dim3(S), dim3(tail_threads), 0, st, fmap, ov, h->pr, fl, src));
            } else {
                LAUNCH(h, "ov_insert_root", hipLaunchKernelGGL((lk_ov_insert_root_kernel<3, true>), dim3(std::max(1, per_slot / 2), S), dim3(LK_MB), 0, st, fmap, ov, h->pr, fl, src));
                LAUNCH(h, "ov_fit_eig", hipLaunchKernelGGL(lk_ov_fit_eig_kernel, dim3(std::max(1, std::min(8, (nb + 63) / 64)), S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
                if (fit_group_r)
                    LAUNCH(h, "ov_fit_lane", hipLaunchKernelGGL(lk_ov_fit_group_kernel, dim3(std::max(1, std::min(8, (nb + 63) / 64)), S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
                else
                    LAUNCH(h, "ov_fit_lane", hipLaunchKernelGGL(lk_ov_fit_lane_kernel, dim3(std::max(1, std::min(8, (nb + 63) / 64)), S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr));
                LAUNCH(h, "ov_insert_apply", hipLaunchKernelGGL(lk_ov_insert_apply_kernel, dim3(per_slot, S), dim3(LK_MB), 0, st, ov, h->pr, fl, src));
            }
            static const int ov_fb_wg_r = getenv("LEGKILO_OV_FB_WG") ? std::max(1, atoi(getenv("LEGKILO_OV_FB_WG"))) : 0;
            LAUNCH(h, "ov_insert_fallback", hipLaunchKernelGGL(lk_ov_insert_fallback_kernel, dim3(std::min((int)S, ov_fb_wg_r ? ov_fb_wg_r : 128)), dim3(LK_MB), 0, st, ov, h->pr, fl, src, (int)S));
        }
        HIPCHK(h, hipGetLastError());
        h->ov_last_slots = (uint32_t)S, h->ov_gen = h->map_gen;
        const unsigned int init[8] = {0u, 0u, 0u, 0u, 0xffffffffu, 0u, 0u, 0u};
        HIPCHK(h, hipMemcpyAsync(h->d_ov_status, init, sizeof(init), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(lk_ov_status_kernel, dim3(std::min((S + 255) / 256, 64)), dim3(256), 0, st, ov, (unsigned int)S, h->d_ov_status);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(stt, h->d_ov_status, sizeof(stt), hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        const bool growable = !h->ov_want_roots && !(stt[0] & ~(LK_E_HASH_FULL | LK_E_NODES_FULL | LK_E_BLOCKS_FULL)) && attempt < 4;
        if (!stt[0] || !growable) break;
        HIPCHK(h, hipMemcpyAsync(h->d_filters, h->d_ov_priors, sizeof(LkFilter) * (size_t)S, hipMemcpyDeviceToDevice, st));
        rc = ov_reserve(h, (uint32_t)S, max_scan_pts, (size_t)biggest, fmap, stt[0]);
        if (rc) return rc;
    }
    if (!stt[0]) h->ov_hw_roots = stt[3], h->ov_hw_nodes = stt[1], h->ov_hw_blocks = stt[2], h->ov_hw_npts = max_scan_pts;
    if (out) {
        std::vector<lk_pose> tmp((size_t)S);
        rc = fetch_poses(h, tmp.data(), S);
        if (rc) return rc;
        memcpy(out, tmp.data(), sizeof(lk_pose) * (size_t)S);
    }
    if (stt[0] & LK_E_KEY_RANGE) {
        char buf[200];
        snprintf(buf, sizeof(buf), "overlay replay: a point of slot %u lies in a voxel whose key is outside the +-2^20 range of the private root tables' packed keys (%.0f km from the origin at this voxel size)",
                 stt[4], 1048576.0 * h->cfg.max_voxel_size / 1000.0);
        return fail(h, LK_ERR_INVALID, buf);
    }
    if (stt[0]) {
        const LkOverlay& ov = h->ov;
        char buf[256];
        snprintf(buf, sizeof(buf), "overlay pool overflow in slot %u (bits 0x%x: 1 private root table, 2 nodes, 4 point blocks, 8 work lists); largest use over the slots: %u nodes, %u blocks, %u roots; per-scan pools: %u root entries, %u child nodes, %u blocks (lk_overlay_reserve)",
                 stt[4], stt[0], stt[1], stt[2], stt[3], ov.hash_cap, ov.nodes_cap - ov.hash_cap, ov.blocks_cap);
        return fail(h, LK_ERR_CAPACITY, buf);
    }
    return LK_OK;
}
extern "C" {

int lk_overlay_stats(lk_handle* h, uint32_t* max_roots, uint32_t* max_nodes, uint32_t* max_blocks) {
    CHECK_H(h);
    if (!h->ov_last_slots || !h->ov.counters) return fail(h, LK_ERR_STATE, "no overlay replay's pools are held by this handle (none has run, or lk_overlay_reserve released them)");
    const unsigned int init[8] = {0u, 0u, 0u, 0u, 0xffffffffu, 0u, 0u, 0u};
    unsigned int stt[8];
    HIPCHK(h, hipMemcpyAsync(h->d_ov_status, init, sizeof(init), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lk_ov_status_kernel, dim3(std::min((h->ov_last_slots + 255u) / 256u, 64u)), dim3(256), 0, h->stream, h->ov, h->ov_last_slots, h->d_ov_status);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(stt, h->d_ov_status, sizeof(stt), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (max_nodes) *max_nodes = stt[1];
    if (max_blocks) *max_blocks = stt[2];
    if (max_roots) *max_roots = stt[3];
    return LK_OK;
}

int lk_overlay_resident_rounds(lk_handle* h, uint32_t* rounds) {
    CHECK_H(h);
    if (rounds) *rounds = h->ov_res_rounds;
    return LK_OK;
}

int lk_overlay_pool_bytes(lk_handle* h, uint64_t* bytes, uint32_t* root_entries, uint32_t* child_nodes, uint32_t* blocks) {
    CHECK_H(h);
    if (bytes) *bytes = (uint64_t)h->ov_pool_bytes;
    if (root_entries) *root_entries = h->ov.hash_cap;
    if (child_nodes) *child_nodes = h->ov.nodes_cap - h->ov.hash_cap;
    if (blocks) *blocks = h->ov.blocks_cap;
    return LK_OK;
}


}  // extern "C"
