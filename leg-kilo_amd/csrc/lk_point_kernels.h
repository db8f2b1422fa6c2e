// lk_point_kernels.h — the data-parallel per-point kernels of one time bucket.
//
//   lk_residual_kernel   KILO.cc:122-210 fused: transform + covariance (K1), voxel-hash probe and
//                        octree plane matching (K2, voxel_map.cc:363-427), ObsShared row, and the
//                        per-workgroup partial sums of A = h^T R^-1 h, b = h^T R^-1 z (K3).
//                        The N x 6 / N x N matrices of the reference are never materialised.
//   lk_reproject_kernel  KILO.cc:216-230 + the hashing half of UpdateVoxelMap (voxel_map.cc:343-358):
//                        re-derive point_w / var with the post-update state, find-or-create the root
//                        voxel, push the point on that root's bucket-local list.
//
// HBM-bound by design (arithmetic intensity ~2 flop/B): one thread per point, lk_point loaded as one
// 16-B float4 per lane (coalesced 1 KiB per wave), plane records are 256-B aligned and read with
// 16-B vector loads, the gate data (first 64 B) before the 6x6 plane covariance (next 168 B).
// grid = (ceil(n/256), n_slots): blockIdx.y selects the filter slot (batch replay).
#pragma once
#include "lk_device.h"

#define LK_PB 256
#ifndef LK_PTS_NT
#define LK_PTS_NT 1    // the 16-B scan point with a non-temporal load (round 6: 1.560 -> 1.537 ms per step, same bits; A/B: -DLK_PTS_NT=0)
#endif
#ifndef LK_ROWS_NT
#define LK_ROWS_NT 1   // config 2's materialised rows leave with non-temporal stores (A/B: -DLK_ROWS_NT=0)
#endif
#if LK_ROWS_NT
#define LK_ROWS_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define LK_ROWS_STORE(v, p) (*(p) = (v))
#endif
#ifndef LK_ROWS_AT_TAKE
#define LK_ROWS_AT_TAKE 1   // a matched lane writes its final LDS row [h z | h/R | R 1] when it takes its plane: no LDS read-back before K3
                            // (0 = park the raw row and finish it after the matching; A/B: 1.798 -> 1.775 ms per step, same bits)
#endif
#ifndef LK_MFMA_RED
#define LK_MFMA_RED 0   // 1: the per-tile normal-equation sums on v_mfma_f64_4x4x4 (A/B build; see residual_tile)
#endif
// perf-attribution switches (tools/ab_env.sh builds with -DLK_X_...=1): each removes one piece of the residual pass so that its
// marginal cost can be measured; results are wrong with any of them set, the product build sets none
#ifndef LK_X_NORETRY
#define LK_X_NORETRY 0
#endif
#ifndef LK_X_NOCHILD
#define LK_X_NOCHILD 0
#endif
#ifndef LK_X_NOEVAL
#define LK_X_NOEVAL 0
#endif
#ifndef LK_X_NORED
#define LK_X_NORED 0
#endif
#ifndef LK_X_NOPRED
#define LK_X_NOPRED 0
#endif
#ifndef LK_X_NOUPD
#define LK_X_NOUPD 0
#endif
#ifndef LK_X_NORES
#define LK_X_NORES 0
#endif
// The residual kernel runs ONE wave per workgroup: with no block barrier in it there is nothing to share, and the
// scheduler can refill a SIMD slot the moment a wave retires instead of waiting for a 4-wave workgroup's worth of
// slots and LDS (measured on the 1024-scan batch: 642 us per bucket at 256 threads, 582 at 128, 578 at 64).
#define LK_RB 64

// The winning candidate's row is parked in the lane's own LDS row the moment it is taken (row[0..2] = w = p_i x (R^T n),
// row[3..5] = n, row[6] = z = -dis, row[13] = sig_r) instead of being carried in ~15 VGPRs through the octree walk: the
// kernel's occupancy is set by registers, and the row has to go to LDS anyway.
struct Match {
    double* row;       // this lane's LDS row (LK_ROW2 doubles)
    float lazy_d;      // |d| and sigma_l of the first passing candidate (deferred probability)
    double lazy_sig;
};

// voxel_map.cc:371-413 for one plane node.  q0..q2 / tail = the first 64 B of the node's match record (center,
// normal, d, radius, flags), already in registers; the remaining 80 B (S11, w, s22) are requested BEFORE the float
// range gate is evaluated so that the whole record costs one memory round trip.
struct RecTail {   // bytes 64..143 of a match record: S11 (6), w (3), s22
    double2 v0, v1, v2, v3, v4;
};
template <bool XID, bool PRE = false>
__device__ __forceinline__ void eval_plane(const lk_match_rec* __restrict__ mr, double2 q0, double2 q1, double2 q2,
                                           float pd, float pradius, int node, int layer, const PointLite& g,
                                           const BucketConst& bc, const LkParams& pr, bool& success, double& prob,
                                           Match& best, const RecTail* pre = nullptr) {
    const double2* sv = reinterpret_cast<const double2*>(mr->s11);  // byte offset 64, 16-B aligned
    double2 v0, v1, v2, v3, v4;
    if (PRE) v0 = pre->v0, v1 = pre->v1, v2 = pre->v2, v3 = pre->v3, v4 = pre->v4;
    else v0 = sv[0], v1 = sv[1], v2 = sv[2], v3 = sv[3], v4 = sv[4];
    V3 c = V3{q0.x, q0.y, q1.x}, n = V3{q1.y, q2.x, q2.y};
    double sd = dot3(n.x, g.p_w.x, n.y, g.p_w.y, n.z, g.p_w.z) + (double)pd;
    float dis_to_plane = (float)fabs(sd);
    float dis_to_center = (float)((c.x - g.p_w.x) * (c.x - g.p_w.x) + (c.y - g.p_w.y) * (c.y - g.p_w.y) +
                                  (c.z - g.p_w.z) * (c.z - g.p_w.z));
    float range_dis = sqrtf(dis_to_center - dis_to_plane * dis_to_plane);
    if (!((double)range_dis <= 3.0 * (double)pradius)) return;  // radius_k = 3
    const V3 q = V3{g.p_w.x - c.x, g.p_w.y - c.y, g.p_w.z - c.z};
    const S3 s11 = S3{v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
    // J plane_var J^T = q^T S11 q - 2 q.w + s22   (w = S12 n, s22 = n^T S22 n precomputed per plane)
    const double sig_pl = quad3(s11, q) - 2.0 * dot3(q.x, v3.x, q.y, v3.y, q.z, v4.x) + v4.y;
    const PlaneTerms t = plane_terms<XID>(g, bc, pr, n);
    const double sig_r = sig_pl + t.ta;
    const double sigma_l = sig_r + (quad3(bc.Prr, t.w) + quad3(bc.Ppp, n));
    // 3-sigma gate  |d| < sigma_num sqrt(sigma_l)  (voxel_map.cc:388) in squared form: both sides are >= 0, and a
    // negative / NaN sigma_l fails either way
    const double d2 = (double)dis_to_plane * (double)dis_to_plane;
    if (!(d2 < (pr.sigma_num * pr.sigma_num) * sigma_l)) return;
    // prob = exp(-d^2 / 2 sigma) / sqrt(sigma) only ranks candidates (voxel_map.cc:389-391).  The first passing
    // candidate always wins against prob = 0 (the 3-sigma gate bounds the exponent by -4.5), so its value is
    // only computed if a second candidate passes; `prob` < 0 encodes "first candidate, value not yet computed".
    bool take;
    if (!success) {
        take = true;
        prob = -1.0;
        best.lazy_d = dis_to_plane;
        best.lazy_sig = sigma_l;
    } else {
        if (prob < 0.0)
            prob = 1.0 / (sqrt(best.lazy_sig)) * exp(-0.5 * (double)best.lazy_d * (double)best.lazy_d / best.lazy_sig);
        const double this_prob = 1.0 / sqrt(sigma_l) * exp(-0.5 * d2 / sigma_l);
        take = this_prob > prob;
        if (take) prob = this_prob;
    }
    success = true;
    if (take) {
        double* r = best.row;
        r[0] = t.w.x, r[1] = t.w.y, r[2] = t.w.z;
        r[3] = n.x, r[4] = n.y, r[5] = n.z;
        r[6] = -(double)(float)sd;  // z = -dis_to_plane_, the signed distance stored as float (voxel_map.h:92, .cc:401-402)
#if LK_ROWS_AT_TAKE
        // the row in its final form [h z | h/R | R 1] right here: nothing is read back from LDS before the reduction
        const double Rv = pr.lidar_ratio * sig_r;   // KILO.cc:205-206
        const double ri = lk_inv_nodecision(Rv);   // scales the row for the sums A = sum h h^T / R, b = sum h z / R: no gate reads it
        r[7] = t.w.x * ri, r[8] = t.w.y * ri, r[9] = t.w.z * ri;
        r[10] = n.x * ri, r[11] = n.y * ri, r[12] = n.z * ri;
        r[13] = Rv;
        r[14] = 1.0;
#else
        r[13] = sig_r;              // J_nq plane_var J_nq^T + n^T (R ext_R) body_cov (R ext_R)^T n (KILO.cc:205-206, before lidar_ratio)
#endif
    }
}

// build_single_residual (voxel_map.cc:363-427): pre-order DFS, children in index order, written as ONE flat loop
// (a single copy of the plane evaluation in the instruction stream; the <=5-deep path lives in scalar registers
// selected with compares, so no dynamically indexed private array / scratch is needed).
// `root` indexes m.match[]: a node id (hash hit), or - grid_cell - a cell of the frozen-map grid, which may be empty and
// otherwise carries its node id in pad_.  Returns whether a root voxel exists there (the lookup of KILO.cc:149 succeeded).
template <bool XID>
__device__ __forceinline__ bool match_root(const LkMap& m, int root, const bool grid_cell, const PointLite& g,
                                           const BucketConst& bc, const LkParams& pr, bool& success, double& prob, Match& best) {
    const int max_layer = pr.max_layer;
    int n0 = root, n1 = -1, n2 = -1, n3 = -1, n4 = -1;
    unsigned int cis = 0;  // next child index of each level, 4 bits per level
    int level = 0;
    bool fresh = true;
    while (level >= 0) {
        int node = (level == 0) ? n0 : (level == 1) ? n1 : (level == 2) ? n2 : (level == 3) ? n3 : n4;
        if (fresh) {
            const lk_match_rec* pl = &m.match[node];
            const double2* q = reinterpret_cast<const double2*>(pl);
            double2 q0 = q[0], q1 = q[1], q2 = q[2];
            float4 tail = *reinterpret_cast<const float4*>(&pl->d);  // d, radius, flags, pad (grid cells: node id)
#if LK_PIN_RECORD
            RecTail rt;   // the whole record in one round trip (see pin_chunk)
            rt.v0 = q[4], rt.v1 = q[5], rt.v2 = q[6], rt.v3 = q[7], rt.v4 = q[8];
            pin_chunk(q0), pin_chunk(q1), pin_chunk(q2), pin_chunk(tail);
            pin_chunk(rt.v0), pin_chunk(rt.v1), pin_chunk(rt.v2), pin_chunk(rt.v3), pin_chunk(rt.v4);
#endif
            if (grid_cell && level == 0) {
                const unsigned int id = __float_as_uint(tail.w);
                if (id == LK_GRID_EMPTY) return false;  // no root voxel at this key
                n0 = node = (int)id;
            }
            if (!LK_X_NOEVAL && (__float_as_uint(tail.z) & LK_PLANE_IS_PLANE)) {
#if LK_PIN_RECORD
                eval_plane<XID, true>(pl, q0, q1, q2, tail.x, tail.y, node, level, g, bc, pr, success, prob, best, &rt);
#else
                eval_plane<XID>(pl, q0, q1, q2, tail.x, tail.y, node, level, g, bc, pr, success, prob, best);
#endif
                --level;
                fresh = false;
                continue;
            }
            if (LK_X_NOCHILD || level >= max_layer || level >= LK_MAX_LAYER) {
                --level;
                fresh = false;
                continue;
            }
            cis &= ~(15u << (4 * level));
        }
        const int4* ch = reinterpret_cast<const int4*>(m.nodes[node].child);
        int4 ca = ch[0], cb = ch[1];
#if LK_PIN_RECORD
        pin_chunk(ca), pin_chunk(cb);
#endif
        unsigned int ci = (cis >> (4 * level)) & 15u;
        int child = -1;
        while (ci < 8u && child < 0) {
            child = (ci == 0) ? ca.x : (ci == 1) ? ca.y : (ci == 2) ? ca.z : (ci == 3) ? ca.w
                  : (ci == 4) ? cb.x : (ci == 5) ? cb.y : (ci == 6) ? cb.z : cb.w;
            ++ci;
        }
        cis = (cis & ~(15u << (4 * level))) | (ci << (4 * level));
        if (child >= 0) {
            ++level;
            if (level == 1) n1 = child;
            else if (level == 2) n2 = child;
            else if (level == 3) n3 = child;
            else n4 = child;
            fresh = true;
        } else {
            --level;
            fresh = false;
        }
    }
    return true;
}

// Matcher of the FROZEN map (grid cells, LkMap::grid_base): the root's cell is a plane record, or the header of the flattened
// list of its subtree's planes (LK_GRID_LIST) - a counted loop over consecutive records, candidates in the reference's
// pre-order, one copy of the plane evaluation.  Returns whether a root voxel exists at the cell.
template <bool XID>
__device__ __forceinline__ bool match_flat(const LkMap& m, int cell, const PointLite& g, const BucketConst& bc, const LkParams& pr,
                                           bool& success, double& prob, Match& best) {
    int idx = cell, remaining = 1;
    bool header = true;   // the record at idx is the cell itself
    while (remaining > 0) {
        const lk_match_rec* pl = &m.match[idx];
        const double2* q = reinterpret_cast<const double2*>(pl);
        double2 q0 = q[0], q1 = q[1], q2 = q[2];
        float4 tail = *reinterpret_cast<const float4*>(&pl->d);  // d, radius, flags, node id
#if LK_PIN_RECORD
        RecTail rt;
        rt.v0 = q[4], rt.v1 = q[5], rt.v2 = q[6], rt.v3 = q[7], rt.v4 = q[8];
        pin_chunk(q0), pin_chunk(q1), pin_chunk(q2), pin_chunk(tail);
        pin_chunk(rt.v0), pin_chunk(rt.v1), pin_chunk(rt.v2), pin_chunk(rt.v3), pin_chunk(rt.v4);
#endif
        const unsigned int fl = __float_as_uint(tail.z);
        if (header) {
            header = false;
            if (__float_as_uint(tail.w) == LK_GRID_EMPTY) return false;  // no root voxel at this key
            if (!(fl & LK_PLANE_IS_PLANE)) {
                if (LK_X_NOCHILD) return true;
                idx = (int)(unsigned int)__double_as_longlong(q0.x);            // {first, count} in the header's first 8 bytes
                remaining = (int)(unsigned int)(__double_as_longlong(q0.x) >> 32);
                continue;
            }
        }
#if LK_PIN_RECORD
        if (!LK_X_NOEVAL) eval_plane<XID, true>(pl, q0, q1, q2, tail.x, tail.y, 0, 0, g, bc, pr, success, prob, best, &rt);
#else
        if (!LK_X_NOEVAL) eval_plane<XID>(pl, q0, q1, q2, tail.x, tail.y, 0, 0, g, bc, pr, success, prob, best);
#endif
        ++idx;
        --remaining;
    }
    return true;
}

// Root voxel of a key for the matcher: index into m.match[] (a node id from the hash table, or a grid cell), -1 = none.
// GRID: 0 = hash table (compile-time), 1 = frozen-map grid (compile-time), 2 = decided by m.grid_on at run time
// (3 = frozen-map grid + one slot's insert overlay: find_root_ov below)
template <int GRID>
__device__ __forceinline__ int find_root(const LkMap& m, int kx, int ky, int kz) {
    if (GRID == 1 || (GRID == 2 && m.grid_on)) {
        const unsigned int ux = (unsigned int)(kx - m.gmin[0]), uy = (unsigned int)(ky - m.gmin[1]), uz = (unsigned int)(kz - m.gmin[2]);
        if (ux >= (unsigned int)m.gdim[0] || uy >= (unsigned int)m.gdim[1] || uz >= (unsigned int)m.gdim[2]) return -1;
        return (int)(m.grid_base + (uz * (unsigned int)m.gdim[1] + uy) * (unsigned int)m.gdim[0] + ux);
    }
    return hash_find(m, kx, ky, kz);  // KILO.cc:149
}

// Batch replay with a per-scan insert overlay (lk_overlay_kernels.h): what the residual pass needs of ONE slot's private map.
struct LkOvView {
    const unsigned long long* keys;   // the slot's private root table: packed keys, entry index = the root's node id
    unsigned int hash_mask;
    const lk_match_rec* match;        // the slot's private match / node pools (ids are slot-local)
    const lk_node_rec* nodes;
    const unsigned int* bits;         // one bit per base grid cell: the slot has a private root at that key
};
// Root of a key for the overlay matcher: >= 0 a cell of the base map's frozen grid (match_flat), <= -2 the private root
// -2 - code of the slot (match_root on the private pools), -1 none.  Only a set bit (the scan has inserted into that voxel) costs
// the trip to the private table.
__device__ __forceinline__ int find_root_ov(const LkMap& base, const LkOvView& ov, int kx, int ky, int kz) {
    const unsigned int ux = (unsigned int)(kx - base.gmin[0]), uy = (unsigned int)(ky - base.gmin[1]), uz = (unsigned int)(kz - base.gmin[2]);
    const bool in = ux < (unsigned int)base.gdim[0] && uy < (unsigned int)base.gdim[1] && uz < (unsigned int)base.gdim[2];
    unsigned int cell = 0;
    if (in) {
        cell = (uz * (unsigned int)base.gdim[1] + uy) * (unsigned int)base.gdim[0] + ux;
        if (!((ov.bits[cell >> 5] >> (cell & 31u)) & 1u)) return (int)(base.grid_base + cell);
    }
    const int lim = 1 << 20;
    if (!(kx < -lim || kx >= lim || ky < -lim || ky >= lim || kz < -lim || kz >= lim)) {
        const unsigned long long pk = ((unsigned long long)((unsigned int)kx & 0x1fffffu)) | ((unsigned long long)((unsigned int)ky & 0x1fffffu) << 21) |
                                      ((unsigned long long)((unsigned int)kz & 0x1fffffu) << 42);
        unsigned int s = lk_hash3(kx, ky, kz) & ov.hash_mask;
        for (unsigned int probe = 0; probe <= ov.hash_mask; probe += 2) {
            const unsigned long long e0 = ov.keys[s], e1 = ov.keys[(s + 1) & ov.hash_mask];
            if (e0 == pk) return -2 - (int)s;
            if (e0 == 0x8000000000000000ull) break;
            if (e1 == pk) return -2 - (int)((s + 1) & ov.hash_mask);
            if (e1 == 0x8000000000000000ull) break;
            s = (s + 2) & ov.hash_mask;
        }
    }
    return in ? (int)(base.grid_base + cell) : -1;
}

// KILO.cc:156-172: key of the ONE neighbour voxel that is tried when the home voxel gave no match.  loc is in
// voxel units, voxel_center +- quater_length in metres — the unit mismatch of the reference is kept as is.  For a
// root voxel, voxel_center = (0.5 + key) * voxel_size and quater_length = voxel_size / 4 (voxel_map.cc:354-357, both
// with the float voxel_size) are functions of the key alone, so the root's node record need not be fetched.
__device__ __forceinline__ void neighbour_key(const LkParams& pr, const float* loc, const int* key, int* near) {
    const double vs = (double)pr.voxel_size_f;
    const double ql = (double)(pr.voxel_size_f / 4);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double vc = (0.5 + key[j]) * vs;
        near[j] = key[j];
        if ((double)loc[j] > (vc + ql))
            near[j] += 1;
        else if ((double)loc[j] < (vc - ql))
            near[j] -= 1;
    }
}

struct ResidualOut {       // optional per-point outputs (config 2 / lk_residuals); any may be null
    double* h6;            // n x 6 row-major
    double* rows8 = nullptr;   // instead of h6 / z / R: n packed records [h(6) z R] of 64 B (lk_batch_residuals_dev)
    double* z;
    double* R;
    unsigned char* valid;
    float* world;          // n x 4 (x y z intensity), cloud_down_world
    int2* ids;             // SPEC instantiations only: per point {home, neighbour} root code of the lookups (spec_code)
    int2* ids_lane = nullptr;   // SPEC: if set, THIS lane's codes go here (a register of the caller) instead of ids[] - a one-tile bucket's checker is the same wave
    // lk_residual_kernel only.  > 0: XCD-aware launch of a BATCH - a 1-D grid of 8 x ceil(tiles / 8) x xmap_slots workgroups.  Workgroups go to the
    // eight XCDs round robin by their linear id (MI355X_MICROARCH.md), so workgroup L runs on XCD L % 8 and is the (L / 8)-th workgroup there: it takes
    // tile (L % 8) * c + (L / 8) % c of scan (L / 8) / c, c = ceil(tiles / 8).  Every XCD then sees the SAME eighth of every scan's bucket - and a
    // bucket comes in voxel order (pcl::VoxelGrid's output order, or lk_batch_sort_by_voxel_dev's), so that eighth is one slab of the map: the
    // plane records an XCD's 4 MB L2 has to hold are an eighth of the map's instead of all of them.  0: the plain 2-D grid (tile, slot).
    int xmap_slots = 0;
    // lk_residual_kernel only: the library's voxel-ordered copy of the caller's batch (lk_batch_order, legkilo_hip.hip).  alt_use[1] != 0 - decided on the
    // device by lk_batch_stamp_kernel just ahead of this launch: the caller's buffer still holds what the copy was made from - reads the copy.
    const lk_point* alt_pts = nullptr;
    const unsigned long long* alt_use = nullptr;
};
// What a speculative residual pass (the pipelined stream path) remembers of a point's two root lookups, so that the verify pass
// can tell whether an insert that ran beside it may have changed the point's result: a root id (>= 0), or - the lookup found no
// root - the complement of the key's slot in LkMap::newroot, or LK_SPEC_NONE when no lookup was made.
#define LK_SPEC_NONE ((int)0x80000000)
__device__ __forceinline__ int spec_code(int root, const int* key) {
    return root >= 0 ? root : ~(int)(lk_hash3(key[0], key[1], key[2]) & LK_NEWROOT_MASK);
}
__device__ __forceinline__ bool spec_suspect(const LkMap& m, int code, unsigned int from) {
    if (code == LK_SPEC_NONE) return false;
    const unsigned int stamp = code >= 0 ? m.dirty[code] : m.newroot[~code];
    return stamp >= from;
}

// LDS row record of one point: h(6), z, 1/R, R   (9 doubles; stride 9 keeps 64-bit LDS reads conflict-free
// for the access pattern of the reduction: lanes of one half-wave read the SAME row, i.e. broadcasts)
#define LK_ROW2 15  // doubles per row: h(6) z | h(6)/R | R valid   (odd stride)

#if LK_OPT_WAVES
#define LK_RES_BOUNDS __launch_bounds__(LK_RB, LK_OPT_WAVES)
#else
#define LK_RES_BOUNDS __launch_bounds__(LK_RB)
#endif
// One 64-point tile of the residual pass, executed by one wave: K1 (transform + covariance terms), K2 (home voxel, one
// neighbour retry), the observation row, and K3 for the tile.  `rows` is the wave's private LDS region (64 x LK_ROW2
// doubles).  Returns, in lane (q, half) = (lane & 31, lane >> 5), component q of [A(21) b(6) sumR count] summed over the
// tile's 64 rows (both halves hold the same value).  Shared by lk_residual_kernel (one tile per single-wave workgroup)
// and lk_small_bucket_kernel (legkilo_hip.hip: a small bucket's tiles inside one workgroup).
// SHORT (the small-bucket kernels, where a tile usually holds a handful of points): the sums run over the rows that hold points,
// rounded up to eight - the rows behind them are zero rows, and fma(0, 0, acc) == acc for every acc this loop can hold (it starts
// at +0 and can never become -0), so the bits are those of the full loop.
template <bool EMIT_ROWS, int GRID = 2, bool XID = false, bool SHORT = false, bool SPEC = false>
__device__ __forceinline__ double residual_tile(const LkMap& map, const LkParams& pr, const BucketConst& bc,
                                                const float4* __restrict__ spts, int i, int n, double* rows, int lane,
                                                const ResidualOut& out, size_t out_base, const LkOvView* ovv = nullptr) {
    bool ok = false;
    double h[6] = {0, 0, 0, 0, 0, 0}, z = 0, R = 0;
    {
        PointLite g;
        int root = -1, nroot = -1;
        if (i < n) {
#if LK_PTS_NT
            typedef float lk_f4v __attribute__((ext_vector_type(4)));
            const lk_f4v p = __builtin_nontemporal_load(reinterpret_cast<const lk_f4v*>(spts) + i);   // read once: keep it from displacing plane records in L2
#else
            const float4 p = spts[i];
#endif
            g = point_lite<XID>(p.x, p.y, p.z, bc, pr);
            if (out.world) {
                float4 w = make_float4((float)g.p_w.x, (float)g.p_w.y, (float)g.p_w.z, 0.f);
                reinterpret_cast<float4*>(out.world + out_base * 4)[i] = w;
            }
            float loc[3];
            int key[3];
            key_trunc(g.p_w, pr, loc, key);
            root = GRID == 3 ? find_root_ov(map, *ovv, key[0], key[1], key[2]) : find_root<GRID>(map, key[0], key[1], key[2]);  // KILO.cc:149
            // stored at once (not carried through the match): registers set this kernel's occupancy
            if (SPEC) {
                if (out.ids_lane) *out.ids_lane = make_int2(spec_code(root, key), LK_SPEC_NONE);
                else out.ids[out_base + i] = make_int2(spec_code(root, key), LK_SPEC_NONE);
            }
        }
        // K2: home voxel first (the root's 144-B record is fetched in one round trip inside match_root)
        bool success = false;
        double prob = 0;
        Match best;
        best.row = rows + lane * LK_ROW2;
        const bool grid_cell = GRID == 1 || GRID == 3 || (GRID == 2 && map.grid_on != 0);
        bool home = false;
        LkMap pmv = {};   // GRID == 3: the slot's private pools (match_root reads match[] and nodes[].child only)
        if (GRID == 3) pmv.match = const_cast<lk_match_rec*>(ovv->match), pmv.nodes = const_cast<lk_node_rec*>(ovv->nodes);
        if (root >= 0) home = grid_cell ? match_flat<XID>(map, root, g, bc, pr, success, prob, best) : match_root<XID>(map, root, false, g, bc, pr, success, prob, best);
        else if (GRID == 3 && root <= -2) home = match_root<XID>(pmv, -2 - root, false, g, bc, pr, success, prob, best);
        // the one-neighbour retry (KILO.cc:156-178): only when the home voxel EXISTS (the lookup at KILO.cc:149 found a tree)
        if (home && !success && !LK_X_NORETRY) {
            float loc[3];     // re-derived here rather than kept alive across the home voxel's walk
            int key[3], near[3];
            key_trunc(g.p_w, pr, loc, key);
            neighbour_key(pr, loc, key, near);
            // the "neighbour" can be the home voxel itself; evaluating it again reproduces the same failure
            if (near[0] != key[0] || near[1] != key[1] || near[2] != key[2]) {
                nroot = GRID == 3 ? find_root_ov(map, *ovv, near[0], near[1], near[2]) : find_root<GRID>(map, near[0], near[1], near[2]);
                if (SPEC) {
                    if (out.ids_lane) out.ids_lane->y = spec_code(nroot, near);
                    else out.ids[out_base + i].y = spec_code(nroot, near);
                }
            }
            if (nroot >= 0) {
                if (grid_cell) match_flat<XID>(map, nroot, g, bc, pr, success, prob, best);
                else match_root<XID>(map, nroot, false, g, bc, pr, success, prob, best);
            } else if (GRID == 3 && nroot <= -2) {
                match_root<XID>(pmv, -2 - nroot, false, g, bc, pr, success, prob, best);
            }
        }
        ok = success;
#if LK_ROWS_AT_TAKE
        // (EMIT_ROWS: the rows go to HBM from the wave's LDS region behind the wave barrier below - whole 16-B pieces, consecutive lanes)
#else
        if (ok) {  // KILO.cc:195-209: h (1x6), z, R for the matched point, from the parked row
            const double* r = best.row;
#pragma unroll
            for (int a = 0; a < 6; ++a) h[a] = r[a];
            z = r[6];
            R = pr.lidar_ratio * r[13];  // (R ext_R) body_cov (R ext_R)^T only, no state covariance (KILO.cc:205-206)
        }
#endif
    }
    // K3, per wave and without any block barrier.  Every lane stores its row [h(6) z | h(6)/R | R valid] (zeros when
    // it did not match) in the wave's LDS region; lane (q = lane & 31, half = lane >> 5) then accumulates component q
    // of [A(21) b(6) sumR count] over the 32 rows of its half, in row order, branch-free and with ONE fma per row
    // (ds_read_b64 broadcasts, all loads independent of the arithmetic); the two halves are combined with one
    // cross-lane read.
#if LK_ROWS_AT_TAKE
    if (!ok) {   // matched lanes wrote their final row when they took their plane
        double* r = rows + lane * LK_ROW2;
#pragma unroll
        for (int a = 0; a < LK_ROW2; ++a) r[a] = 0.0;
    }
#else
    {
        double* r = rows + lane * LK_ROW2;
        const double ri = ok ? 1.0 / R : 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) r[a] = h[a];   // h, z are zero for unmatched lanes
        r[6] = z;
#pragma unroll
        for (int a = 0; a < 6; ++a) r[7 + a] = h[a] * ri;
        r[13] = ok ? R : 0.0;
        r[14] = ok ? 1.0 : 0.0;
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (EMIT_ROWS) {
        // config 2: the tile's 64 observation rows to HBM; rows of unmatched points are zero.  Whole 16-B pieces, consecutive lanes: every
        // store instruction of the wave covers 1 024 contiguous bytes (8 whole cache lines) instead of 64 eight-byte words a row apart
        // (2.4 -> 0.5 L2 write requests per point).  Non-temporal: the rows stream out and are never read again by this launch - kept out of
        // the L2's way they do not evict the plane records every other wave is matching against (HBM reads 38.6 -> 27.7 B per point).
        typedef double lk_d2v __attribute__((ext_vector_type(2)));
        const int tile0 = i - lane;                                   // wave-uniform
        const int nv = n - tile0 < 64 ? n - tile0 : 64;               // points of this tile (<= 0: a tile behind the scan's end)
        const size_t o0 = out_base + (size_t)tile0;
        if (out.rows8) {
            // packed records [h(6) z R] of 64 B (lk_batch_residuals_dev): the tile is ONE contiguous 4 096-B block, lane l stores pieces l, 64 + l, ...
            lk_d2v* dst = reinterpret_cast<lk_d2v*>(out.rows8 + o0 * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = k * 64 + lane, row = e >> 2, c = (e & 3) * 2;
                if (row < nv) {
                    const double* r = rows + row * LK_ROW2;
                    lk_d2v v;
                    v.x = r[c], v.y = c == 6 ? r[13] : r[c + 1];   // doubles 0..6 of the LDS row are h, z; R sits at 13
                    LK_ROWS_STORE(v, dst + e);
                }
            }
        } else {
            // lk_residuals' separate arrays: the tile's h6 block is 64 x 48 B = 192 contiguous pieces
            lk_d2v* dst = reinterpret_cast<lk_d2v*>(out.h6 + o0 * 6);   // (o0 * 48) B: 16-B aligned with the buffer
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = k * 64 + lane, row = e / 3, c = (e - row * 3) * 2;
                if (row < nv) {
                    lk_d2v v;
                    v.x = rows[row * LK_ROW2 + c], v.y = rows[row * LK_ROW2 + c + 1];
                    LK_ROWS_STORE(v, dst + e);
                }
            }
            if (lane < nv) {
                const double* r = rows + lane * LK_ROW2;
                LK_ROWS_STORE(r[6], out.z + o0 + lane);
                LK_ROWS_STORE(r[13], out.R + o0 + lane);
            }
        }
        if (lane < nv) LK_ROWS_STORE((unsigned char)(rows[lane * LK_ROW2 + 14] != 0.0 ? 1 : 0), out.valid + o0 + lane);
        return 0.0;   // config 2 ends here ("residuals only"): no caller of a row-emitting launch reads the tile's sums A, b (K3)
    }
    if (LK_X_NORED) return (lane == 28) ? (ok ? 1.0 : 0.0) : 0.0;
#if LK_MFMA_RED
    // K3 on the matrix cores: G = X^T Y over the tile's 64 rows with X = [h/R (6), R, valid] (row doubles 7..14) and
    // Y = [h (6), z, valid] (row doubles 0..6, 14): G(i,j), i <= j < 6, is A; G(i,6) is b; G(6,7) = sum R; G(7,7) = count.
    // v_mfma_f64_4x4x4_4b: four independent 4 x 4 blocks, K = 4 per instruction -> the four blocks are the 2 x 2 tiling of the
    // 8 x 8 result and 16 instructions walk the 64 rows.  Operand layout measured on gfx950 (tools/probes/mfma_f64_4x4x4_layout.hip):
    // lane = 16 k + 4 b + m supplies A[b][m][k] and B[b][k][m]; D[b][i][j] lands in lane 16 i + 4 b + j.
    {
        const int k = lane >> 4, b = (lane >> 2) & 3, m = lane & 3;
        const int ca = 7 + 4 * (b >> 1) + m;
        const int cb0 = 4 * (b & 1) + m, cb = cb0 < 7 ? cb0 : 14;
        const double* pa = rows + k * LK_ROW2 + ca;
        const double* pb = rows + k * LK_ROW2 + cb;
        double d0 = 0.0, d1 = 0.0;   // two accumulators: consecutive MFMAs do not wait for each other
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
            d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[(4 * t) * LK_ROW2], pb[(4 * t) * LK_ROW2], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(pa[(4 * t + 4) * LK_ROW2], pb[(4 * t + 4) * LK_ROW2], d1, 0, 0, 0);
        }
        const double g = d0 + d1;
        // lane q (both halves) wants component q of [A(21) b(6) sumR count]
        const int q = lane & 31;
        // (gi, gj) of component q, packed gi | gj << 4, eight per 64-bit word: the upper triangle of A row by row, then (i, 6), (6, 7), (7, 7)
        const unsigned long long tw = (q < 8) ? 0x2111504030201000ull : (q < 16) ? 0x3352423222514131ull : (q < 24) ? 0x6261605554445343ull : 0x0000007776656463ull;
        const unsigned int ij = (unsigned int)(tw >> ((q & 7) * 8)) & 0xffu;
        const int gi = (int)(ij & 15u), gj = (int)(ij >> 4);
        const int src = 16 * (gi & 3) + 4 * (2 * (gi >> 2) + (gj >> 2)) + (gj & 3);
        return __shfl(g, src, LK_WAVE);
    }
#endif
    const int q = lane & 31, half = lane >> 5;
    // component q = sum over rows of r[a] * r[b]:  A(i,j) = sum (h_i / R) h_j (upper triangle, row-major: q < 21,
    // a = 7 + i, b = j), b_i = sum (h_i / R) z (q = 21 + i: a = 7 + i, b = 6), sum R (q = 27: 13, 14), count (14, 14).
    // Packed as a | b << 4, eight entries per 64-bit word.
    const unsigned long long tw = (q < 8) ? 0x2818574737271707ull : (q < 16) ? 0x3a59493929584838ull : (q < 24) ? 0x6968675c5b4b5a4aull : 0xeeeeeeeeed6c6b6aull;
    const unsigned int ab = (unsigned int)(tw >> ((q & 7) * 8)) & 0xffu;
    const int a = (int)(ab & 15u), b = (int)(ab >> 4);
    const double* base = rows + (half * 32) * LK_ROW2;
    double acc = 0.0;
    if (SHORT) {
        const int nv = __builtin_amdgcn_readfirstlane(n - (i - lane));   // points in this tile (wave-uniform)
        const int jm = nv >= 32 ? 32 : ((nv + 7) & ~7);
        for (int j0 = 0; j0 < jm; j0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double* r = base + (j0 + j) * LK_ROW2;
                acc = __builtin_fma(r[a], r[b], acc);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const double* r = base + j * LK_ROW2;
            acc = __builtin_fma(r[a], r[b], acc);
        }
    }
    acc += __shfl_xor(acc, 32, LK_WAVE);
    return acc;
}

// One partial record per WAVE -> lk_update_kernel adds them in a fixed order (deterministic).
template <bool EMIT_ROWS, int GRID = 0, bool XID = false, bool SPEC = false>
__global__ void LK_RES_BOUNDS
    lk_residual_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                       size_t pts_slot_stride, int n, double* __restrict__ partials, size_t part_slot_stride,
                       ResidualOut out, size_t out_slot_stride) {
    // per-wave LDS region holding the wave's 64 observation rows
    __shared__ double stage[LK_RB / LK_WAVE][64 * LK_ROW2];
    int slot = blockIdx.y, bx = blockIdx.x;
    if (out.xmap_slots > 0) {   // XCD-aware batch launch (ResidualOut::xmap_slots); uniform per workgroup
        const unsigned int T = (unsigned int)((n + LK_RB - 1) / LK_RB), c = (T + 7u) >> 3;
        const unsigned int x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const unsigned int t = x * c + j % c;
        slot = (int)(j / c);
        if (t >= T || slot >= out.xmap_slots) return;
        bx = (int)t;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    if (out.alt_use && out.alt_use[1] != 0ull) pts = out.alt_pts;   // wave-uniform (scalar load): the voxel-ordered copy of the same scans
    const double acc = residual_tile<EMIT_ROWS, GRID, XID, false, SPEC>(map, pr, bc, reinterpret_cast<const float4*>(pts + (size_t)slot * pts_slot_stride),
                                                bx * LK_RB + tid, n, &stage[wv][0], lane, out, (size_t)slot * out_slot_stride);
    if (!EMIT_ROWS && lane < LK_NPART) {
        const size_t wave_id = (size_t)bx * (LK_RB / LK_WAVE) + wv;
        partials[(size_t)slot * part_slot_stride + wave_id * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
    }
}

// ---------------------------------------------------------------- round 5 experiment: TWO tiles per wave (-> profiles/EXPERIMENTS.md)
// The batch kernel waits 46 % of a wave's life: one trip for the point, one for the candidate's record.  Here a wave takes two 64-point
// tiles and requests BOTH tiles' points, then both tiles' cell records (18 pinned 16-B pieces per lane), before it evaluates either: the
// second tile's trips overlap the first tile's arithmetic.  The price is the second record and point in registers.  Frozen-map grid only
// (match_flat), no row emission; arithmetic and order of sums per tile exactly those of residual_tile (the same bits per partial record).
struct TilePre {
    PointLite g;
    int root;
    double2 q0, q1, q2;
    float4 tail;
    RecTail rt;
};
template <bool XID>
__device__ __forceinline__ void tile_request(const LkMap& map, const LkParams& pr, const BucketConst& bc, const float4& p, bool in_range, TilePre& t) {
    t.root = -1;
    if (in_range) {
        t.g = point_lite<XID>(p.x, p.y, p.z, bc, pr);
        float loc[3];
        int key[3];
        key_trunc(t.g.p_w, pr, loc, key);
        t.root = find_root<1>(map, key[0], key[1], key[2]);
    }
    const lk_match_rec* pl = &map.match[t.root >= 0 ? t.root : (int)map.grid_base];   // (clamped: every lane requests, nothing is merged in)
    const double2* q = reinterpret_cast<const double2*>(pl);
    t.q0 = q[0], t.q1 = q[1], t.q2 = q[2];
    t.tail = *reinterpret_cast<const float4*>(&pl->d);
    t.rt.v0 = q[4], t.rt.v1 = q[5], t.rt.v2 = q[6], t.rt.v3 = q[7], t.rt.v4 = q[8];
}
// match_flat with the cell's own record already requested (t): the first iteration uses it, list records are loaded as before
template <bool XID>
__device__ __forceinline__ bool match_flat_pre(const LkMap& m, TilePre& t, const BucketConst& bc, const LkParams& pr, bool& success, double& prob, Match& best) {
    pin_chunk(t.q0), pin_chunk(t.q1), pin_chunk(t.q2), pin_chunk(t.tail);
    pin_chunk(t.rt.v0), pin_chunk(t.rt.v1), pin_chunk(t.rt.v2), pin_chunk(t.rt.v3), pin_chunk(t.rt.v4);
    const unsigned int fl0 = __float_as_uint(t.tail.z);
    if (__float_as_uint(t.tail.w) == LK_GRID_EMPTY) return false;
    if (fl0 & LK_PLANE_IS_PLANE) {
        eval_plane<XID, true>(&m.match[t.root], t.q0, t.q1, t.q2, t.tail.x, t.tail.y, 0, 0, t.g, bc, pr, success, prob, best, &t.rt);
        return true;
    }
    int idx = (int)(unsigned int)__double_as_longlong(t.q0.x), remaining = (int)(unsigned int)(__double_as_longlong(t.q0.x) >> 32);
    while (remaining > 0) {
        const lk_match_rec* pl = &m.match[idx];
        const double2* q = reinterpret_cast<const double2*>(pl);
        double2 q0 = q[0], q1 = q[1], q2 = q[2];
        float4 tail = *reinterpret_cast<const float4*>(&pl->d);
        RecTail rt;
        rt.v0 = q[4], rt.v1 = q[5], rt.v2 = q[6], rt.v3 = q[7], rt.v4 = q[8];
        pin_chunk(q0), pin_chunk(q1), pin_chunk(q2), pin_chunk(tail);
        pin_chunk(rt.v0), pin_chunk(rt.v1), pin_chunk(rt.v2), pin_chunk(rt.v3), pin_chunk(rt.v4);
        eval_plane<XID, true>(pl, q0, q1, q2, tail.x, tail.y, 0, 0, t.g, bc, pr, success, prob, best, &rt);
        ++idx;
        --remaining;
    }
    return true;
}
// the rest of a tile: K2 (home voxel from the requested record, one neighbour retry), row, K3
template <bool XID>
__device__ __forceinline__ double tile_finish(const LkMap& map, const LkParams& pr, const BucketConst& bc, TilePre& t, double* rows, int lane) {
    bool success = false;
    double prob = 0;
    Match best;
    best.row = rows + lane * LK_ROW2;
    bool home = false;
    if (t.root >= 0) home = match_flat_pre<XID>(map, t, bc, pr, success, prob, best);
    if (home && !success) {
        float loc[3];
        int key[3], near[3];
        key_trunc(t.g.p_w, pr, loc, key);
        neighbour_key(pr, loc, key, near);
        int nroot = -1;
        if (near[0] != key[0] || near[1] != key[1] || near[2] != key[2]) nroot = find_root<1>(map, near[0], near[1], near[2]);
        if (nroot >= 0) match_flat<XID>(map, nroot, t.g, bc, pr, success, prob, best);
    }
    const bool ok = success;
    if (!ok) {
        double* r = rows + lane * LK_ROW2;
#pragma unroll
        for (int a = 0; a < LK_ROW2; ++a) r[a] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int q = lane & 31, half = lane >> 5;
    const unsigned long long tw = (q < 8) ? 0x2818574737271707ull : (q < 16) ? 0x3a59493929584838ull : (q < 24) ? 0x6968675c5b4b5a4aull : 0xeeeeeeeeed6c6b6aull;
    const unsigned int ab = (unsigned int)(tw >> ((q & 7) * 8)) & 0xffu;
    const int a = (int)(ab & 15u), b = (int)(ab >> 4);
    const double* base = rows + (half * 32) * LK_ROW2;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const double* r = base + j * LK_ROW2;
        acc = __builtin_fma(r[a], r[b], acc);
    }
    acc += __shfl_xor(acc, 32, LK_WAVE);
    return acc;
}
#ifndef LK_PAIR_WAVES
#define LK_PAIR_WAVES 3
#endif
template <bool XID>
__global__ void __launch_bounds__(LK_RB, LK_PAIR_WAVES)
    lk_residual_pair_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts, size_t pts_slot_stride, int n,
                            double* __restrict__ partials, size_t part_slot_stride, ResidualOut out, size_t out_slot_stride) {
    static_assert(LK_RB == LK_WAVE, "one wave per workgroup");
    __shared__ double stage[64 * LK_ROW2];
    if (blockIdx.x & 1u) return;   // the launch keeps lk_residual_kernel's grid: even workgroups take tiles b and b + 1
    const int slot = blockIdx.y, lane = threadIdx.x;
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    const float4* spts = reinterpret_cast<const float4*>(pts + (size_t)slot * pts_slot_stride);
    const int i0 = blockIdx.x * LK_RB + lane, i1 = i0 + LK_RB;
    const float4 p0 = spts[min(i0, n - 1)], p1 = spts[min(i1, n - 1)];   // both points first, clamped: no lane test in front of a load
    TilePre t0, t1;
    tile_request<XID>(map, pr, bc, p0, i0 < n, t0);
    tile_request<XID>(map, pr, bc, p1, i1 < n, t1);
    const double acc0 = tile_finish<XID>(map, pr, bc, t0, stage, lane);
    if (lane < LK_NPART) partials[(size_t)slot * part_slot_stride + (size_t)blockIdx.x * LK_NPART + lane] = (lane < 29) ? acc0 : 0.0;
    if ((int)((blockIdx.x + 1) * LK_RB) >= n) return;
    __builtin_amdgcn_wave_barrier();   // tile 0's reads of the rows are complete
    const double acc1 = tile_finish<XID>(map, pr, bc, t1, stage, lane);
    if (lane < LK_NPART) partials[(size_t)slot * part_slot_stride + (size_t)(blockIdx.x + 1) * LK_NPART + lane] = (lane < 29) ? acc1 : 0.0;
}

// ---------------------------------------------------------------- pipelined stream path: verify pass
// The stream path runs the insert of bucket k on its own HIP stream while bucket k+1's predict + residual pass (SPEC
// instantiation: it also stores each point's two root codes) run on the main stream.  An insert changes what a point's match can
// see only inside the subtrees of the roots it stamps (LkMap::dirty / newroot).  This pass runs when those stamps are final: a tile
// none of whose points looked at a stamped root keeps its speculative partial record - it was computed from data no insert touched
// - and any other tile waits until the insert has completed (LK_SPEC_DONE, bounded spin) and is evaluated again, from scratch.
// The result is what the sequential order predict -> residual would have produced after the insert: same code, same bits.
__device__ __forceinline__ void spec_wait(const LkMap& m, int word, unsigned int need) {
    if (need != 0 && (threadIdx.x & 63) == 0) {
        const unsigned long long t0 = wall_clock64();   // 100 MHz constant clock
        while (__hip_atomic_load(&m.spec[word], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > 20000000ull) {   // 0.2 s: the insert stream is not making progress - fail the call, never hang
                atomicOr(&m.counters[LK_CTR_ERR], LK_E_SPEC_TIMEOUT);
                break;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
template <bool XID>
__global__ void LK_RES_BOUNDS
    lk_verify_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts, int n,
                     double* __restrict__ partials, ResidualOut out, unsigned int dirty_from, unsigned int need_done) {
    __shared__ double stage[64 * LK_ROW2];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * LK_RB + lane;
    bool susp = false;
    if (i < n) {
        const int2 c = out.ids[i];
        susp = spec_suspect(map, c.x, dirty_from) || spec_suspect(map, c.y, dirty_from);
    }
    if (__ballot(susp) == 0ull) return;
    if (lane == 0) atomicAdd(&map.counters[LK_CTR_SPEC_REDO], 1u);
    spec_wait(map, LK_SPEC_DONE, need_done);
    BucketConst bc;
    load_bucket_const<false>(&filters[0], pr, bc);
    const double acc = residual_tile<false, 0, XID>(map, pr, bc, reinterpret_cast<const float4*>(pts), i, n, stage, lane, out, (size_t)0);
    if (lane < LK_NPART) partials[(size_t)blockIdx.x * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
}

// Ragged batch: bucket b of every scan that has one.  grid = (waves of the LARGEST bucket b, scans); a workgroup beyond
// its scan's bucket leaves at once (two scalar loads).
template <int GRID>
__global__ void LK_RES_BOUNDS
    lk_residual_ragged_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                              LkRagged rg, int b, double* __restrict__ partials, size_t part_slot_stride) {
    __shared__ double stage[LK_RB / LK_WAVE][64 * LK_ROW2];
    const int slot = blockIdx.y;
    if (b >= rag_nb(rg, slot)) return;
    const unsigned long long* po = rag_pt_off(rg, slot);
    const unsigned long long base = po[b];
    const int n = (int)(po[b + 1] - base);
    if ((int)(blockIdx.x * LK_RB) >= n) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    BucketConst bc;
    load_bucket_const<false>(&filters[slot], pr, bc);
    ResidualOut out;
    out.h6 = nullptr, out.z = nullptr, out.R = nullptr, out.valid = nullptr, out.world = nullptr;
    const double acc = residual_tile<false, GRID>(map, pr, bc, reinterpret_cast<const float4*>(pts + base), blockIdx.x * LK_RB + tid, n,
                                            &stage[wv][0], lane, out, (size_t)0);
    if (lane < LK_NPART) {
        const size_t wave_id = (size_t)blockIdx.x * (LK_RB / LK_WAVE) + wv;
        partials[(size_t)slot * part_slot_stride + wave_id * LK_NPART + lane] = (lane < 29) ? acc : 0.0;
    }
}

// ---------------------------------------------------------------- find-or-create a root voxel
// voxel_map.cc:345-357.  Lock-free for readers; a creator claims the slot (EMPTY -> LOCKED), writes
// the key and the root node, then publishes the node id.  A thread that meets a LOCKED slot retries
// the same slot on its next loop trip (the claimer finishes inside one trip, so there is no
// intra-wave deadlock).
__device__ __forceinline__ int root_find_or_create(const LkMap& m, const LkParams& pr, const int* key) {
    // fast path: roots that existed before this launch are found with plain loads (no atomics).
    // Unpublished slots carry the key sentinel INT_MIN, so a half-written entry can never match.
    {
        int r = hash_find(m, key[0], key[1], key[2]);
        if (r >= 0) return r;
    }
    unsigned int s = lk_hash3(key[0], key[1], key[2]) & m.hash_mask;
    int* slotw = reinterpret_cast<int*>(m.hash);
    for (unsigned int trips = 0; trips < 64u * (m.hash_mask + 1u); ++trips) {
        int w = __hip_atomic_load(&slotw[4 * s + 3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (w == LK_EMPTY) {
            int expected = LK_EMPTY;
            if (__hip_atomic_compare_exchange_strong(&slotw[4 * s + 3], &expected, LK_LOCKED, __ATOMIC_ACQ_REL,
                                                     __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) {
                unsigned int id = atomicAdd(&m.counters[LK_CTR_NODES], 1u);
                if (id >= m.max_nodes) {
                    atomicOr(&m.counters[LK_CTR_ERR], LK_E_NODES_FULL);
                    __hip_atomic_store(&slotw[4 * s + 3], LK_EMPTY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    return -1;
                }
                atomicAdd(&m.counters[LK_CTR_ROOTS], 1u);
                lk_node_rec* nd = &m.nodes[id];
                double vs = (double)pr.voxel_size_f;
#pragma unroll
                for (int c = 0; c < 8; ++c) nd->child[c] = -1;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    nd->voxel_center[c] = (0.5 + key[c]) * vs;  // voxel_map.cc:355-357
                    nd->key[c] = key[c];
                }
                nd->quater_length = pr.voxel_size_f / 4;          // voxel_map.cc:354
                nd->layer = 0;
                nd->npts = 0;
                nd->new_points = 0;
                nd->state = LK_NODE_UPDATE_ENABLE;
                nd->block = -1;
                nd->list_head = -1;
                nd->pad_[0] = 0;  // list count
                m.planes[id].flags = 0;
                m.match[id].flags = 0;
                // pipelined stream path: a residual pass that runs beside this insert may see the key absent or present - both the
                // key (for lookups that found nothing) and the new root id are stamped with the insert's epoch (LkMap::dirty)
                m.dirty[id] = m.epoch;
                m.newroot[lk_hash3(key[0], key[1], key[2]) & LK_NEWROOT_MASK] = m.epoch;
                slotw[4 * s + 0] = key[0];
                slotw[4 * s + 1] = key[1];
                slotw[4 * s + 2] = key[2];
                __hip_atomic_store(&slotw[4 * s + 3], (int)id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return (int)id;
            }
            continue;  // lost the race: re-read this slot
        }
        if (w == LK_LOCKED) {
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        int kx = __hip_atomic_load(&slotw[4 * s + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ky = __hip_atomic_load(&slotw[4 * s + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int kz = __hip_atomic_load(&slotw[4 * s + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kx == key[0] && ky == key[1] && kz == key[2]) return w;
        s = (s + 1) & m.hash_mask;
    }
    atomicOr(&m.counters[LK_CTR_ERR], LK_E_HASH_FULL);
    return -1;
}

// Queue bucket-local point i on its root voxel: the first LK_SLOTS points of a root go to its inline slot line (so
// the per-root kernels read all indices with one coalesced load), later ones to a linked list; the first arrival
// registers the root in the touched list.  Arrival order is arbitrary: the consumers sort by index.
__device__ __forceinline__ void queue_point_on_root(const LkMap& map, int root, int i) {
    const unsigned int k = atomicAdd(&map.nodes[root].pad_[0], 1u);
    if (k < (unsigned int)LK_SLOTS) {
        map.slots[(size_t)root * LK_SLOTS + k] = i;
    } else {
        map.next[i] = atomicExch(&map.nodes[root].list_head, i);
    }
    if (k == 0) {
        unsigned int t = atomicAdd(&map.counters[LK_CTR_TOUCHED], 1u);
        map.touched[t] = root;
    }
}

// KILO.cc:216-230 + voxel_map.cc:343-358 (hash half).  pts are the bucket's points (bucket-local i).
// bc: R, p of the posterior (nothing else of it is used); updated: the bucket's update took place (KILO.cc:213-215)
__device__ __forceinline__ void dev_reproject_point_bc(const LkMap& map, const LkParams& pr, const BucketConst& bc, const bool updated,
                                                       const lk_point* __restrict__ pts, float* __restrict__ world /* n x 4 or null */,
                                                       const int do_insert, const int i) {
    const float4 p = reinterpret_cast<const float4*>(pts)[i];
    struct { V3 p_w; } g;
    g.p_w = point_world(p.x, p.y, p.z, bc, pr);
    if (world && updated) {
        reinterpret_cast<float4*>(world)[i] = make_float4((float)g.p_w.x, (float)g.p_w.y, (float)g.p_w.z, 255.f);
    }
    if (!do_insert) return;
    int key[3];
    key_floor(g.p_w, pr.voxel_size_f, key);
    int root = root_find_or_create(map, pr, key);
    if (root < 0) return;
    // Drop points that UpdateOctoTree would ignore (voxel_map.cc:185-241), per point and in parallel: walk down from
    // the root while the node is an initialised NON-plane below max_layer (such nodes are never refitted, so they
    // stay what they are), and skip the point if the node it ends in is frozen — a plane, or a max-layer leaf, whose
    // update_enable_ is already false (frozen is permanent).  Anything else (un-initialised node, live leaf, a child
    // that does not exist yet) is queued for the ordered per-root replay.
    {
        int node = root;
        bool ignore = false;
        for (int depth = 0; depth <= LK_MAX_LAYER; ++depth) {
            const lk_node_rec* nr = &map.nodes[node];
#if LK_PIN_RECORD
            // the node's record (children, centre, layer, state: bytes 0..79) and its plane flags in ONE round trip
            int4 n0 = reinterpret_cast<const int4*>(nr)[0], n1 = reinterpret_cast<const int4*>(nr)[1], n2 = reinterpret_cast<const int4*>(nr)[2],
                 n3 = reinterpret_cast<const int4*>(nr)[3], n4 = reinterpret_cast<const int4*>(nr)[4];
            unsigned int pf = map.planes[node].flags;
            pin_chunk(n0), pin_chunk(n1), pin_chunk(n2), pin_chunk(n3), pin_chunk(n4);
            asm volatile("" : "+v"(pf));
            const unsigned int st = (unsigned int)n4.z;
            if (!(st & LK_NODE_INIT_OCTO)) break;
            const bool is_plane = (pf & LK_PLANE_IS_PLANE) != 0;
            const int layer = n3.w;
            if (is_plane || layer >= pr.max_layer) {
                ignore = !(st & LK_NODE_UPDATE_ENABLE);
                break;
            }
            const double cx = __hiloint2double(n2.y, n2.x), cy = __hiloint2double(n2.w, n2.z), cz = __hiloint2double(n3.y, n3.x);
            const int oct = ((g.p_w.x > cx) ? 4 : 0) + ((g.p_w.y > cy) ? 2 : 0) + ((g.p_w.z > cz) ? 1 : 0);
            const int child = oct == 0 ? n0.x : oct == 1 ? n0.y : oct == 2 ? n0.z : oct == 3 ? n0.w : oct == 4 ? n1.x : oct == 5 ? n1.y : oct == 6 ? n1.z : n1.w;
            if (child < 0) break;
            node = child;
#else
            const unsigned int st = nr->state;
            const unsigned int pf = map.planes[node].flags;
            if (!(st & LK_NODE_INIT_OCTO)) break;
            const bool is_plane = (pf & LK_PLANE_IS_PLANE) != 0;
            const int layer = nr->layer;
            if (is_plane || layer >= pr.max_layer) {
                ignore = !(st & LK_NODE_UPDATE_ENABLE);
                break;
            }
            const int oct = ((g.p_w.x > nr->voxel_center[0]) ? 4 : 0) + ((g.p_w.y > nr->voxel_center[1]) ? 2 : 0) +
                            ((g.p_w.z > nr->voxel_center[2]) ? 1 : 0);
            const int child = nr->child[oct];
            if (child < 0) break;
            node = child;
#endif
        }
        if (ignore) return;
    }
    queue_point_on_root(map, root, i);
}
__device__ __forceinline__ void dev_reproject_point(const LkMap& map, const LkParams& pr, const LkFilter* __restrict__ filters,
                                                    const lk_point* __restrict__ pts, float* __restrict__ world /* n x 4 or null */,
                                                    const int do_insert, const int i) {
    const LkFilter* f = &filters[0];
    BucketConst bc;
    load_bucket_const<false>(f, pr, bc);   // R, p only matter here (no R * ext_R product, no covariance blocks used)
    dev_reproject_point_bc(map, pr, bc, world ? f->updated != 0 : false, pts, world, do_insert, i);
}
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_PB)
    lk_reproject_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                        int n, float* __restrict__ world /* n x 4 or null */, int do_insert);
#else
__global__ void __launch_bounds__(LK_PB)
    lk_reproject_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                        int n, float* __restrict__ world /* n x 4 or null */, int do_insert) {
    const int i = blockIdx.x * LK_PB + threadIdx.x;
    if (i >= n) return;
    dev_reproject_point(map, pr, filters, pts, world, do_insert, i);
}
#endif
