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

struct Match {
    int node;
    int layer;
    V3 n;          // plane normal
    V3 c;          // plane center
    float dis;     // signed distance stored as float (voxel_map.h:92, .cc:401-402)
    double sig_pl; // J_nq * plane_var * J_nq^T
};

// voxel_map.cc:371-413 for one plane node.  Returns nothing; updates success / prob / best.
__device__ __forceinline__ void eval_plane(const lk_plane_rec* __restrict__ pl, int node, int layer, const PointGeom& g,
                                           double sigma_num, bool& success, double& prob, Match& best) {
    const double2* q = reinterpret_cast<const double2*>(pl);
    double2 q0 = q[0], q1 = q[1], q2 = q[2];   // center xyz, normal xyz
    float2 dr = *reinterpret_cast<const float2*>(&pl->d);
    V3 c = V3{q0.x, q0.y, q1.x}, n = V3{q1.y, q2.x, q2.y};
    double sd = n.x * g.p_w.x + n.y * g.p_w.y + n.z * g.p_w.z + (double)dr.x;
    float dis_to_plane = (float)fabs(sd);
    float dis_to_center = (float)((c.x - g.p_w.x) * (c.x - g.p_w.x) + (c.y - g.p_w.y) * (c.y - g.p_w.y) +
                                  (c.z - g.p_w.z) * (c.z - g.p_w.z));
    float range_dis = sqrtf(dis_to_center - dis_to_plane * dis_to_plane);
    if (!((double)range_dis <= 3.0 * (double)dr.y)) return;  // radius_k = 3
    double J[6] = {g.p_w.x - c.x, g.p_w.y - c.y, g.p_w.z - c.z, -n.x, -n.y, -n.z};
    // sigma = J * plane_var * J^T evaluated as (J * PV) * J^T with the symmetric upper triangle
    double t[6] = {0, 0, 0, 0, 0, 0};
    const double* pv = pl->plane_var;
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int cc = r; cc < 6; ++cc) {
            double v = pv[k++];
            t[cc] += J[r] * v;
            if (cc != r) t[r] += J[cc] * v;
        }
    double sig_pl = t[0] * J[0] + t[1] * J[1] + t[2] * J[2] + t[3] * J[3] + t[4] * J[4] + t[5] * J[5];
    double sigma_l = sig_pl + quad3(g.var, n);
    if (!((double)dis_to_plane < sigma_num * sqrt(sigma_l))) return;
    success = true;
    double this_prob = 1.0 / (sqrt(sigma_l)) * exp(-0.5 * (double)dis_to_plane * (double)dis_to_plane / sigma_l);
    if (this_prob > prob) {
        prob = this_prob;
        best.node = node;
        best.layer = layer;
        best.n = n;
        best.c = c;
        best.dis = (float)sd;
        best.sig_pl = sig_pl;
    }
}

// build_single_residual (voxel_map.cc:363-427): pre-order DFS, children in index order.  The depth is a
// template parameter (<= LK_MAX_LAYER) so that no dynamically indexed private array (scratch) is needed.
template <int L>
__device__ __forceinline__ void match_node(const LkMap& m, int node, int max_layer, const PointGeom& g,
                                           double sigma_num, bool& success, double& prob, Match& best) {
    unsigned int flags = m.planes[node].flags;
    if (flags & LK_PLANE_IS_PLANE) {
        eval_plane(&m.planes[node], node, L, g, sigma_num, success, prob, best);
        return;
    }
    if constexpr (L < LK_MAX_LAYER) {
        if (L < max_layer) {
            for (int ci = 0; ci < 8; ++ci) {
                int child = m.nodes[node].child[ci];
                if (child >= 0) match_node<L + 1>(m, child, max_layer, g, sigma_num, success, prob, best);
            }
        }
    }
}
__device__ __forceinline__ void match_root(const LkMap& m, int root, int max_layer, const PointGeom& g, double sigma_num,
                                           bool& success, double& prob, Match& best) {
    match_node<0>(m, root, max_layer, g, sigma_num, success, prob, best);
}

// KILO.cc:142-183: root lookup, match, one-neighbour retry (unit-mismatch comparison kept)
__device__ __forceinline__ bool match_point(const LkMap& m, const LkParams& pr, const PointGeom& g, Match& best) {
    float loc[3];
    int key[3];
    key_trunc(g.p_w, pr.voxel_size_d, loc, key);
    int root = hash_find(m, key[0], key[1], key[2]);
    if (root < 0) return false;
    bool success = false;
    double prob = 0;
    match_root(m, root, pr.max_layer, g, pr.sigma_num, success, prob, best);
    if (!success) {
        const lk_node_rec* nr = &m.nodes[root];
        double ql = (double)nr->quater_length;
        int near[3] = {key[0], key[1], key[2]};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double vc = nr->voxel_center[j];
            if ((double)loc[j] > (vc + ql))
                near[j] += 1;
            else if ((double)loc[j] < (vc - ql))
                near[j] -= 1;
        }
        int nroot = hash_find(m, near[0], near[1], near[2]);
        if (nroot >= 0) match_root(m, nroot, pr.max_layer, g, pr.sigma_num, success, prob, best);
    }
    return success;
}

// KILO.cc:195-209: h (1x6), z, R for a matched point
__device__ __forceinline__ void obs_row(const Match& b, const PointGeom& g, const BucketConst& bc, double ratio,
                                        double* h, double& z, double& R) {
    V3 u = mat3T_mul_v(bc.R, b.n);  // R^T n
    // crossmat(p_i) * u
    h[0] = -g.p_i.z * u.y + g.p_i.y * u.z;
    h[1] = g.p_i.z * u.x - g.p_i.x * u.z;
    h[2] = -g.p_i.y * u.x + g.p_i.x * u.y;
    h[3] = b.n.x, h[4] = b.n.y, h[5] = b.n.z;
    z = -(double)b.dis;
    S3 vb = congruence(bc.RE, g.body);  // (R ext_R) body_cov (R ext_R)^T, no state covariance (KILO.cc:205-206)
    R = ratio * (b.sig_pl + quad3(vb, b.n));
}

struct ResidualOut {       // optional per-point outputs (config 2 / lk_residuals); any may be null
    double* h6;            // n x 6 row-major
    double* z;
    double* R;
    unsigned char* valid;
    float* world;          // n x 4 (x y z intensity), cloud_down_world
};

template <bool EMIT_ROWS>
__global__ void __launch_bounds__(LK_PB)
    lk_residual_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                       size_t pts_slot_stride, int n, double* __restrict__ partials, size_t part_slot_stride,
                       ResidualOut out, size_t out_slot_stride) {
    __shared__ double red[LK_PB / LK_WAVE][LK_NPART];
    const int slot = blockIdx.y;
    const int i = blockIdx.x * LK_PB + threadIdx.x;
    BucketConst bc;
    load_bucket_const(&filters[slot], pr, bc);
    double acc[29];
#pragma unroll
    for (int q = 0; q < 29; ++q) acc[q] = 0.0;
    if (i < n) {
        const float4 p = reinterpret_cast<const float4*>(pts + (size_t)slot * pts_slot_stride)[i];
        PointGeom g = point_geom(p.x, p.y, p.z, bc, pr);
        if (out.world) {
            float4 w = make_float4((float)g.p_w.x, (float)g.p_w.y, (float)g.p_w.z, 0.f);
            reinterpret_cast<float4*>(out.world + (size_t)slot * out_slot_stride * 4)[i] = w;
        }
        Match best;
        bool ok = match_point(map, pr, g, best);
        double h[6] = {0, 0, 0, 0, 0, 0}, z = 0, R = 0;
        if (ok) {
            obs_row(best, g, bc, pr.lidar_ratio, h, z, R);
            double ri = 1.0 / R;
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double ha = h[a] * ri;
#pragma unroll
                for (int b = a; b < 6; ++b) acc[q++] = ha * h[b];
                acc[21 + a] = ha * z;
            }
            acc[27] = R;
            acc[28] = 1.0;
        }
        if (EMIT_ROWS) {
            size_t o = (size_t)slot * out_slot_stride + i;
            out.valid[o] = ok ? 1 : 0;
            out.z[o] = z;
            out.R[o] = R;
#pragma unroll
            for (int a = 0; a < 6; ++a) out.h6[o * 6 + a] = h[a];
        }
    }
    // K3: wave shuffle reduction, then 4 waves through LDS, one partial record per workgroup
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 29; ++q) {
        double v = wave_sum(acc[q]);
        if (lane == 0) red[wv][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < LK_NPART) {
        double s = 0.0;
        if (threadIdx.x < 29)
            for (int w = 0; w < LK_PB / LK_WAVE; ++w) s += red[w][threadIdx.x];
        partials[(size_t)slot * part_slot_stride + (size_t)blockIdx.x * LK_NPART + threadIdx.x] = s;
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

// KILO.cc:216-230 + voxel_map.cc:343-358 (hash half).  pts are the bucket's points (bucket-local i).
__global__ void __launch_bounds__(LK_PB)
    lk_reproject_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const lk_point* __restrict__ pts,
                        int n, float* __restrict__ world /* n x 4 or null */, int do_insert) {
    const int i = blockIdx.x * LK_PB + threadIdx.x;
    if (i >= n) return;
    const LkFilter* f = &filters[0];
    BucketConst bc;
    load_bucket_const(f, pr, bc);
    const float4 p = reinterpret_cast<const float4*>(pts)[i];
    PointGeom g = point_geom(p.x, p.y, p.z, bc, pr);
    if (world && f->updated) {
        reinterpret_cast<float4*>(world)[i] = make_float4((float)g.p_w.x, (float)g.p_w.y, (float)g.p_w.z, 255.f);
    }
    if (!do_insert) return;
    int key[3];
    key_floor(g.p_w, pr.voxel_size_f, key);
    int root = root_find_or_create(map, pr, key);
    if (root < 0) return;
    // a frozen plane root ignores every further point (voxel_map.cc:191-204): skip the list entirely
    unsigned int st = map.nodes[root].state;
    unsigned int pf = map.planes[root].flags;
    if ((st & LK_NODE_INIT_OCTO) && (pf & LK_PLANE_IS_PLANE) && !(st & LK_NODE_UPDATE_ENABLE)) return;
    int old = atomicExch(&map.nodes[root].list_head, i);
    map.next[i] = old;
    atomicAdd(&map.nodes[root].pad_[0], 1u);
    if (old == -1) {
        unsigned int t = atomicAdd(&map.counters[LK_CTR_TOUCHED], 1u);
        map.touched[t] = root;
    }
}
