// lk_device.h — device-side data model and fp64 helpers for the gfx950 kernels.
//
// Data layout in HBM (one lk_handle):
//   hash     : int4 {kx,ky,kz,node}[cap]  open addressing, linear probing, node<0 = empty     16 B/slot
//   planes   : lk_plane_rec[max_nodes]     256 B, 256-aligned: what the residual kernel streams
//   nodes    : lk_node_rec[max_nodes]      128 B: octree topology + insert state machine fields
//   blocks   : lk_block_rec[max_blocks]    52 x 72 B points (point_w + sym var) of a live leaf
//   filters  : LkFilter[n_slots]           state (36) + P (900) + per-bucket constants
//   scan     : lk_point[max_scan_points]   x,y,z,curvature f32 (16 B, float4 loads)
// Everything is fp64 except where the reference itself computes in float
// (voxel_map.cc:24-25, :374-379, stored d_/radius_/eigenvalues).  Built with
// -ffp-contract=off so that the same expression gives the same bits at every call site
// (the insert kernel re-derives point_w / var and must route a point exactly like the
// kernel that hashed it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/legkilo_hip.h"

// occupancy target of the residual kernel (tools/ab_libs.sh A/B: DESIGN.md section 6)
#ifndef LK_OPT_WAVES
#define LK_OPT_WAVES 5   // residual kernel: 5 waves per SIMD (<= 96 VGPRs, 28 B of spills); measured 562 -> 520 us per 20.5 M points
#endif
#define LK_WAVE 64
#define LK_EMPTY (-1)
#define LK_LOCKED (-2)
#define LK_MAX_LAYER 4
#define LK_SLOTS 32  // inline index slots per root voxel (one 128-B line); more points overflow to a linked list
#define LK_NPART 32  // doubles per block partial: A(21) b(6) sumR(1) count(1) pad

// error bits in LkMap.counters[LK_CTR_ERR]
#define LK_E_HASH_FULL 1u
#define LK_E_NODES_FULL 2u
#define LK_E_BLOCKS_FULL 4u
#define LK_E_SCRATCH_FULL 8u
#define LK_E_BAD_BLOB 16u
#define LK_E_KEY_RANGE 64u      // overlay replay: a point's voxel key does not fit the packed 3 x 21-bit key of the private root tables (not a capacity problem: no pool growth helps)
#define LK_E_SPEC_TIMEOUT 32u   // a verify wave of the pipelined stream path gave up waiting for the insert stream (bounded spin)

enum { LK_CTR_NODES = 0, LK_CTR_BLOCKS = 1, LK_CTR_ROOTS = 2, LK_CTR_ERR = 3, LK_CTR_TOUCHED = 4, LK_CTR_SCRATCH = 5,
       LK_CTR_HEAVY = 6, LK_CTR_FREE = 7 /* signed: blocks poppable this bucket */, LK_CTR_FREED = 8 /* blocks retired
       during this bucket */, LK_CTR_GROUPS = 9 /* leaf groups of this bucket */, LK_CTR_GIDX = 10 /* their indices */, LK_CTR_FALLBACK = 11 /* groups handed to the generic code */,
       LK_CTR_SPEC_REDO = 12 /* cumulative: tiles the verify pass of the pipelined stream path evaluated again */,
       /* 13: LK_CTR_NODES0 of a private overlay map (lk_overlay_kernels.h) */
       LK_CTR_RES_REDO = 14 /* cumulative: buckets the scan-resident stream kernel evaluated again after a conflicting insert */,
       LK_CTR_COUNT = 16 };

struct LkFilter {
    double x[LK_STATE_DOUBLES];  // rot(9) pos vel ba bw grav imu_a imu_w bv contact
    double P[900];               // row-major
    double last_predict_t, last_update_t;
    unsigned long long n_effect;
    unsigned int n_updates, n_buckets;
    int updated;                 // did the last point bucket update (KILO.cc:188)
    int last_N;
    double pad_[4];
};

struct LkParams {                // immutable per handle, passed by value
    double ext_R[9], ext_T[3];
    double voxel_size_d;         // config max_voxel_size_ (double use at KILO.cc:145)
    double sigma_num;
    double lidar_ratio;
    double dir_var;              // pow(sin(DEG2RAD(beam_err)),2), voxel_map.cc:27
    double inv_vs_exact;         // 1/voxel_size when that is exact in fp64 (power of two), else 0
    float voxel_size_f;          // float use at voxel_map.cc:289,337
    float range_var;             // dept_err^2 in float, voxel_map.cc:25
    float planer_threshold;      // voxel_map.h:140
    int max_layer, max_points_num;
    int layer_init_num[5];
    int ext_identity;            // ext_R is exactly the identity (leg_fusion.yaml / diter.yaml): ext_R v == v bit for bit, the two 3x3 products per point / candidate are skipped
};

// Compact, derived copy of a plane for the residual kernel (device-only, never exported): 144 B = 9 x 16-B
// loads instead of the 15 of lk_plane_rec.  With J = [q, -n], q = p - center, and plane_var = [[S11 S12],[S21 S22]]
//   J plane_var J^T = q^T S11 q - 2 q^T (S12 n) + n^T S22 n
// so only S11 (6 unique), w = S12 n and s22 = n^T S22 n are needed per plane; they are written whenever the
// plane is (re)fitted or imported.  Same value as the reference's (J PV) J^T up to fp64 rounding.
struct lk_match_rec {
    double center[3];
    double normal[3];
    float d, radius;
    unsigned int flags, pad_;
    double s11[6];               // xx xy xz yy yz zz
    double w[3];
    double s22;
};
static_assert(sizeof(lk_match_rec) == 144, "match record must be 144 B");

__host__ __device__ inline void lk_derive_match(const lk_plane_rec* pl, lk_match_rec* mr) {
    for (int k = 0; k < 3; ++k) mr->center[k] = pl->center[k], mr->normal[k] = pl->normal[k];
    mr->d = pl->d, mr->radius = pl->radius, mr->flags = pl->flags, mr->pad_ = 0;
    const double* v = pl->plane_var;  // upper triangle row-major of the 6x6: row r starts at r*6 - r*(r-1)/2
    // S11 = rows/cols 0..2
    mr->s11[0] = v[0], mr->s11[1] = v[1], mr->s11[2] = v[2], mr->s11[3] = v[6], mr->s11[4] = v[7], mr->s11[5] = v[11];
    // S12 (3x3, rows 0..2, cols 3..5): row0 = v[3..5], row1 = v[8..10], row2 = v[12..14]
    const double* n = pl->normal;
    mr->w[0] = v[3] * n[0] + v[4] * n[1] + v[5] * n[2];
    mr->w[1] = v[8] * n[0] + v[9] * n[1] + v[10] * n[2];
    mr->w[2] = v[12] * n[0] + v[13] * n[1] + v[14] * n[2];
    // S22 = rows/cols 3..5: (3,3)=v[15] (3,4)=v[16] (3,5)=v[17] (4,4)=v[18] (4,5)=v[19] (5,5)=v[20]
    const double t0 = v[15] * n[0] + v[16] * n[1] + v[17] * n[2];
    const double t1 = v[16] * n[0] + v[18] * n[1] + v[19] * n[2];
    const double t2 = v[17] * n[0] + v[19] * n[1] + v[20] * n[2];
    mr->s22 = t0 * n[0] + t1 * n[1] + t2 * n[2];
}

struct LkMap {                   // device pointers of one voxel map, passed by value
    int4* hash;
    lk_plane_rec* planes;
    lk_match_rec* match;         // derived, same index as planes[]
    lk_node_rec* nodes;
    lk_block_rec* blocks;
    unsigned int* counters;      // LK_CTR_*
    int* touched;                // roots touched by the current bucket
    int* heavy;                  // subset of touched that needs the wave-per-root state machine
    int* next;                   // per-point list links (bucket-local index): overflow beyond LK_SLOTS
    int* slots;                  // [max_nodes][LK_SLOTS] bucket-local point indices queued on a root
    int* scratch;                // per-root gathered indices
    int* groups;                 // 2 x [max_scan] x 8 ints: leaf-group descriptors of the current bucket (LkGroup), then the fallback items
    int* gidx;                   // [max_scan] bucket-local point indices of the groups, input order inside a group
    int* free_list;              // point blocks that may be re-allocated during this bucket
    int* freed_next;             // point blocks retired during this bucket (allocatable from the next bucket on)
    unsigned int hash_mask, max_nodes, max_blocks, max_scan;
    // Pipelined stream path (legkilo_hip.hip, "spec"): the insert of bucket k runs on its own HIP stream while bucket k+1's
    // predict + residual pass run; whatever the insert may change under the residual pass is stamped with the bucket's epoch:
    //   dirty[node]   root voxels (layer-0 node ids) whose subtree's planes may change: new roots, roots on the heavy list
    //   newroot[h]    h = lk_hash3(key) & LK_NEWROOT_MASK of a root CREATED by the insert (a lookup that found nothing has no id)
    //   spec[LK_SPEC_DONE]  epoch of the last insert that has completed (written by its last workgroup)
    unsigned int* dirty;
    unsigned int* newroot;
    unsigned int* spec;
    unsigned int epoch;          // epoch of the bucket whose insert this launch belongs to (0 outside the stream path)
    // Frozen-map acceleration structure of batch replay (frozen_map(), legkilo_hip.hip): the root voxels' match records in a
    // DENSE 3-D array over the bounding box of the root keys, so that the per-point chain scan point -> hash slot -> record
    // loses its middle trip (key -> cell index is arithmetic).  The cells live in the SAME array as the nodes' match records,
    // behind them: cell c = match[grid_base + c] - one base address and 32-bit record indices for both, which is what keeps
    // the matcher's addressing scalar-base + lane-offset.  cell.pad_ = node id, LK_GRID_EMPTY where no root exists.
    // Derived data: rebuilt after any change of the map, never exported; grid_on = 0 selects the hash table.
    unsigned int grid_base;
    int gmin[3], gdim[3];
    int grid_on;
};
#define LK_NEWROOT_MASK 16383u
enum { LK_SPEC_DONE = 0, LK_SPEC_TICKET = 1, LK_SPEC_DECIDED = 2, LK_SPEC_TICKET2 = 3, LK_SPEC_WORDS = 16 };
#define LK_GRID_EMPTY 0xffffffffu
// A grid cell of a root that is NOT a plane is a list header: flags carries LK_GRID_LIST, the first 8 bytes (center[0]) hold
// {first, count}: the plane nodes of the root's subtree in the pre-order of build_single_residual (voxel_map.cc:415-421),
// flattened into consecutive match records behind the grid (count == 0: nothing to evaluate).  The matcher of the frozen map
// is then a counted loop over consecutive records - no child pointers, no walk state.
#define LK_GRID_LIST 0x100u

// Device tables of a RAGGED batch (lk_batch_replay_ragged_dev): every scan has its own number of points, its own
// time buckets and its own start time.  Passed by value.
struct LkRagged {
    const unsigned long long* pt_off;  // [S][ldb + 1]: index into the point array of the first point of bucket b of scan s
    const double* t;                   // [S][ldb]: absolute time of bucket b of scan s (t_begin + curvature, KILO.cc:376)
    const unsigned int* nb;            // [S]: number of buckets of scan s
    int ldb;                           // row pitch of t; pt_off rows have ldb + 1 entries
    const unsigned int* bstart;        // optional [S + 1]: CSR form (tables built on the device, lk_batch_replay_scans_dev) - scan s owns the
                                       // buckets bstart[s] .. bstart[s+1) of the FLAT pt_off / t arrays; null: the padded [S][ldb] form above
    const unsigned int* imu_off;       // optional [S + 1]: messages of scan s = imu[imu_off[s] .. imu_off[s+1]), time-sorted
    const double* imu;                 // [n][msg_stride]: lk_imu (7 doubles: stamp, acc, gyr) or lk_kin_imu (33 doubles, stamp first)
    int msg_stride;                    // 7 (only_imu_use) or 33 (leg fusion: kinematic + IMU messages)
    double kin_noise;                  // kin_meas_noise (KILO.cc:305)
    int q_diag;                        // the process noise Q is diagonal (initProcessCovQ, eskf.cc:47-62): P += dt^2 Q touches 30 entries
    double acc_scale;                  // gravity / acc_norm (KILO.cc:246)
    double Rn[6];                      // IMU measurement noise: acc, acc, acc_z, gyr, gyr, gyr
};

__device__ __forceinline__ int rag_nb(const LkRagged& rg, int slot) {
    return rg.bstart ? (int)(rg.bstart[slot + 1] - rg.bstart[slot]) : (int)rg.nb[slot];
}
__device__ __forceinline__ const unsigned long long* rag_pt_off(const LkRagged& rg, int slot) {
    return rg.bstart ? rg.pt_off + rg.bstart[slot] : rg.pt_off + (size_t)slot * (rg.ldb + 1);
}
__device__ __forceinline__ const double* rag_t(const LkRagged& rg, int slot) {
    return rg.bstart ? rg.t + rg.bstart[slot] : rg.t + (size_t)slot * rg.ldb;
}

// ---------------------------------------------------------------- small fp64 helpers
struct V3 {
    double x, y, z;
};
struct S3 {  // symmetric 3x3: xx xy xz yy yz zz
    double xx, xy, xz, yy, yz, zz;
};

// a*b + c*d + e*f with explicit FMAs.  The translation unit is built with -ffp-contract=off so that the
// compiler never decides contraction per call site; where fusing pays it is spelled out here, and every kernel
// that shares these helpers therefore computes identical bits (see the note at the top of this file).
__device__ __forceinline__ double dot3(double a, double b, double c, double d, double e, double f) {
    return __builtin_fma(e, f, __builtin_fma(c, d, a * b));
}
__device__ __forceinline__ V3 mat3_mul_v(const double* M, V3 v) {  // row-major 3x3
    return V3{dot3(M[0], v.x, M[1], v.y, M[2], v.z), dot3(M[3], v.x, M[4], v.y, M[5], v.z),
              dot3(M[6], v.x, M[7], v.y, M[8], v.z)};
}
__device__ __forceinline__ V3 mat3T_mul_v(const double* M, V3 v) {
    return V3{dot3(M[0], v.x, M[3], v.y, M[6], v.z), dot3(M[1], v.x, M[4], v.y, M[7], v.z),
              dot3(M[2], v.x, M[5], v.y, M[8], v.z)};
}
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = dot3(A[3 * i], B[j], A[3 * i + 1], B[3 + j], A[3 * i + 2], B[6 + j]);
}
// C = A * S * A^T for symmetric S, result symmetric (upper triangle evaluated once)
__device__ __forceinline__ S3 congruence(const double* A, S3 s) {
    double T[9];  // T = A * S
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a0 = A[3 * i], a1 = A[3 * i + 1], a2 = A[3 * i + 2];
        T[3 * i + 0] = dot3(a0, s.xx, a1, s.xy, a2, s.xz);
        T[3 * i + 1] = dot3(a0, s.xy, a1, s.yy, a2, s.yz);
        T[3 * i + 2] = dot3(a0, s.xz, a1, s.yz, a2, s.zz);
    }
    S3 r;
    r.xx = dot3(T[0], A[0], T[1], A[1], T[2], A[2]);
    r.xy = dot3(T[0], A[3], T[1], A[4], T[2], A[5]);
    r.xz = dot3(T[0], A[6], T[1], A[7], T[2], A[8]);
    r.yy = dot3(T[3], A[3], T[4], A[4], T[5], A[5]);
    r.yz = dot3(T[3], A[6], T[4], A[7], T[5], A[8]);
    r.zz = dot3(T[6], A[6], T[7], A[7], T[8], A[8]);
    return r;
}
__device__ __forceinline__ double quad3(S3 s, V3 n) {  // n^T S n, evaluated as (n^T S) n
    double t0 = dot3(n.x, s.xx, n.y, s.xy, n.z, s.xz);
    double t1 = dot3(n.x, s.xy, n.y, s.yy, n.z, s.yz);
    double t2 = dot3(n.x, s.xz, n.y, s.yz, n.z, s.zz);
    return dot3(t0, n.x, t1, n.y, t2, n.z);
}
__device__ __forceinline__ void skew3(V3 v, double* K) {
    K[0] = 0.0, K[1] = -v.z, K[2] = v.y, K[3] = v.z, K[4] = 0.0, K[5] = -v.x, K[6] = -v.y, K[7] = v.x, K[8] = 0.0;
}
// Rodrigues: I + sin(a) K + (1-cos(a)) K K  (math_utils.hpp:19-68)
__device__ __forceinline__ void rodrigues3(V3 axis, double ang, double* R) {
    double K[9], KK[9];
    skew3(axis, K);
    mat3_mul(K, K, KK);
    double s, c;
    sincos(ang, &s, &c);   // one argument reduction for both (same values as sin() and cos())
    const double c1 = 1.0 - c;
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + c1 * KK[i];
}
// Both Exp overloads of math_utils.hpp as one instruction stream (they differ in the threshold only): lets two lanes of a wave
// evaluate one each AT THE SAME TIME (wave_predict_core) instead of one after the other.
__device__ __forceinline__ void exp_so3_thr(double v1, double v2, double v3, double thr, double* R) {
    double n = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (n > thr) {
        rodrigues3(V3{v1 / n, v2 / n, v3 / n}, n, R);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
}
// Exp(v1,v2,v3), threshold 1e-5 (math_utils.hpp:54-68)
__device__ __forceinline__ void exp3_1e5(double v1, double v2, double v3, double* R) {
    double n = sqrt(v1 * v1 + v2 * v2 + v3 * v3);
    if (n > 0.00001) {
        rodrigues3(V3{v1 / n, v2 / n, v3 / n}, n, R);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
}
// Exp(vec&&), threshold 1e-7 (math_utils.hpp:19-32)
__device__ __forceinline__ void expv_1e7(V3 a, double* R) {
    double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (n > 0.0000001) {
        rodrigues3(V3{a.x / n, a.y / n, a.z / n}, n, R);
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
}

// ---------------------------------------------------------------- per-point geometry (KILO.cc:126-140)
struct PointGeom {
    V3 p_i;   // point in the IMU frame
    V3 p_w;   // point in the world frame
    S3 body;  // calcBodyCov
    S3 var;   // world covariance incl. state covariance
};

// calcBodyCov, voxel_map.cc:22-40 (range / range_var in float as in the reference).
// The reference builds two tangent vectors b1, b2 (3 normalisations) and A = range [d]x [b1 b2], then
//   cov = d rv d^T + A dv A^T.
// Because [b1 b2] is an orthonormal basis of the plane normal to the unit ray d,  N N^T = I - d d^T  and
// [d]x d = 0, so  A dv A^T = range^2 dv [d]x (I - d d^T) [d]x^T = range^2 dv (I - d d^T)  identically.
// The closed form below is that expression: same value to fp64 rounding (the reference's own result carries
// the rounding of its three normalisations), ~20 VALU instructions instead of ~250 (3 sqrt + 10 fp64 divides).
__device__ __forceinline__ S3 calc_body_cov(V3 pb, const LkParams& pr) {
    if (pb.z == 0) pb.z = 0.0001;
    const double n2 = dot3(pb.x, pb.x, pb.y, pb.y, pb.z, pb.z);
    const double nrm = sqrt(n2);
    const double r = (double)(float)nrm;                 // float range, voxel_map.cc:24
    const double inrm = 1.0 / nrm;
    const V3 d = V3{pb.x * inrm, pb.y * inrm, pb.z * inrm};
    const double beta = (r * r) * pr.dir_var;            // range^2 * sin^2(beam_err)
    const double alpha = (double)pr.range_var - beta;    // dept_err^2 - range^2 sin^2(beam_err)
    S3 c;
    c.xx = __builtin_fma(alpha * d.x, d.x, beta);
    c.xy = (alpha * d.x) * d.y;
    c.xz = (alpha * d.x) * d.z;
    c.yy = __builtin_fma(alpha * d.y, d.y, beta);
    c.yz = (alpha * d.y) * d.z;
    c.zz = __builtin_fma(alpha * d.z, d.z, beta);
    return c;
}

// state-dependent constants of one bucket, derived from LkFilter by each thread (uniform)
struct BucketConst {
    double R[9], p[3], RE[9];
    S3 Prr, Ppp;
};
template <bool WITH_RE = true>
__device__ __forceinline__ void load_bucket_const(const LkFilter* __restrict__ f, const LkParams& pr, BucketConst& bc) {
#pragma unroll
    for (int i = 0; i < 9; ++i) bc.R[i] = f->x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bc.p[i] = f->x[9 + i];
    if (WITH_RE) mat3_mul(bc.R, pr.ext_R, bc.RE);
    const double* P = f->P;
    bc.Prr = S3{P[0], P[1], P[2], P[31], P[32], P[62]};
    bc.Ppp = S3{P[3 * 30 + 3], P[3 * 30 + 4], P[3 * 30 + 5], P[4 * 30 + 4], P[4 * 30 + 5], P[5 * 30 + 5]};
}

// KILO.cc:126-140 (and :219-228): p_i, p_w, body_var, var
__device__ __forceinline__ PointGeom point_geom(float bx, float by, float bz, const BucketConst& bc, const LkParams& pr) {
    PointGeom g;
    V3 pb = V3{(double)bx, (double)by, (double)bz};
    V3 e = mat3_mul_v(pr.ext_R, pb);
    g.p_i = V3{e.x + pr.ext_T[0], e.y + pr.ext_T[1], e.z + pr.ext_T[2]};
    V3 w = mat3_mul_v(bc.R, g.p_i);
    g.p_w = V3{w.x + bc.p[0], w.y + bc.p[1], w.z + bc.p[2]};
    g.body = calc_body_cov(pb, pr);
    double K[9], RK[9];
    skew3(g.p_i, K);
    mat3_mul(bc.R, K, RK);
    S3 a = congruence(bc.RE, g.body);
    S3 b = congruence(RK, bc.Prr);
    g.var = S3{a.xx + b.xx + bc.Ppp.xx, a.xy + b.xy + bc.Ppp.xy, a.xz + b.xz + bc.Ppp.xz,
               a.yy + b.yy + bc.Ppp.yy, a.yz + b.yz + bc.Ppp.yz, a.zz + b.zz + bc.Ppp.zz};
    return g;
}

// The world point alone (KILO.cc:219-222), by the very expressions of point_geom / point_lite: the re-projection only hashes
// the point and writes it out - the covariance is needed by the insert passes, which derive it themselves.
__device__ __forceinline__ V3 point_world(float bx, float by, float bz, const BucketConst& bc, const LkParams& pr) {
    V3 pb = V3{(double)bx, (double)by, (double)bz};
    V3 e = mat3_mul_v(pr.ext_R, pb);
    V3 p_i = V3{e.x + pr.ext_T[0], e.y + pr.ext_T[1], e.z + pr.ext_T[2]};
    V3 w = mat3_mul_v(bc.R, p_i);
    return V3{w.x + bc.p[0], w.y + bc.p[1], w.z + bc.p[2]};
}

// Residual-side point geometry: what lk_residual_kernel needs of KILO.cc:126-140 WITHOUT materialising the 3x3
// covariances.  The matcher and the observation row only ever consume  n^T var n  for a candidate normal n
// (voxel_map.cc:384-388, KILO.cc:205-206), and with body_cov = alpha d d^T + beta I (calc_body_cov above),
// K = [p_i]x, u = R^T n, m = (R ext_R)^T n = ext_R^T u:
//   n^T (R ext_R) body_cov (R ext_R)^T n = alpha (d.m)^2 + beta (m.m),      d.m = (pb.m) / |pb|
//   n^T (R K) P_rr (R K)^T n            = w^T P_rr w,  w = p_i x u  (= the rotation part of the H row, KILO.cc:197-199)
//   n^T P_pp n
// These are identities in exact arithmetic for ANY R / ext_R (no orthonormality is assumed); in fp64 they differ
// from the reference's matrix route by rounding only (~1e-16 relative), ~60 flops per candidate instead of ~230 per
// point + 12 per candidate, and ~40 fewer live VGPRs.
// 1 / x for the residual kernel's two divisions per point (finite x well inside the normal range): v_rcp_f64 + two Newton steps
// instead of the IEEE division sequence (two v_div_scale, v_rcp, five fma, v_div_fmas, v_div_fixup) - the same value to the last
// bit or one ulp beside it, ~20 instructions fewer per point.  LK_FAST_RCP=0: the full division (A/B).
#ifndef LK_FAST_RCP
#define LK_FAST_RCP 0   // measured +1.5 % (662 k vs 652 k scans/s, same box) - and one match of the 1024-scan batch flips (a gate within an ulp of its threshold): the oracle divides, so does the shipped build
#endif
__device__ __forceinline__ double lk_inv(double x) {
#if LK_FAST_RCP
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}
// 1 / x for a value that feeds NO decision: v_rcp_f64 + two Newton steps (5 instructions instead of the ~12 of an IEEE division; the last bit may differ)
#ifndef LK_FAST_RCP_ROW
#define LK_FAST_RCP_ROW 1   // round 6: 1.548 -> 1.538 ms per step, parity sample unchanged (A/B: -DLK_FAST_RCP_ROW=0)
#endif
__device__ __forceinline__ double lk_inv_nodecision(double x) {
#if LK_FAST_RCP_ROW
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}
struct PointLite {
    V3 p_i, p_w;
    double alpha_n, beta;  // alpha / |pb|^2, beta
    float gz;              // 1e-4 when calcBodyCov's z == 0 guard fired (voxel_map.cc:23), else 0
};
template <bool XID = false>
__device__ __forceinline__ PointLite point_lite(float bx, float by, float bz, const BucketConst& bc, const LkParams& pr) {
    PointLite g;
    V3 pb = V3{(double)bx, (double)by, (double)bz};
    V3 e = XID ? pb : mat3_mul_v(pr.ext_R, pb);   // XID: ext_R is exactly I, and 1 * x + 0 * y + 0 * z == x bit for bit
    g.p_i = V3{e.x + pr.ext_T[0], e.y + pr.ext_T[1], e.z + pr.ext_T[2]};
    V3 w = mat3_mul_v(bc.R, g.p_i);
    g.p_w = V3{w.x + bc.p[0], w.y + bc.p[1], w.z + bc.p[2]};
    g.gz = 0.f;
    if (pb.z == 0) pb.z = 0.0001, g.gz = 0.0001f;
    const double n2 = dot3(pb.x, pb.x, pb.y, pb.y, pb.z, pb.z);
    const double r = (double)(float)sqrt(n2);            // float range, voxel_map.cc:24
    g.beta = (r * r) * pr.dir_var;
    g.alpha_n = ((double)pr.range_var - g.beta) * lk_inv(n2);
    return g;
}
struct PlaneTerms {  // per (point, candidate normal)
    V3 w;            // p_i x (R^T n)
    double ta;       // n^T (R ext_R) body_cov (R ext_R)^T n
};
template <bool XID = false>
__device__ __forceinline__ PlaneTerms plane_terms(const PointLite& g, const BucketConst& bc, const LkParams& pr, V3 n) {
    PlaneTerms t;
    const V3 u = mat3T_mul_v(bc.R, n);
    const V3 m = XID ? u : mat3T_mul_v(pr.ext_R, u);
    // pb.m = pb^T ext_R^T u = (ext_R pb).u = (p_i - ext_T).u  (+ the guard's 1e-4 along ext_R's third column = 1e-4 m.z):
    // the body point itself need not stay in registers
    const double dm = dot3(g.p_i.x - pr.ext_T[0], u.x, g.p_i.y - pr.ext_T[1], u.y, g.p_i.z - pr.ext_T[2], u.z) + (double)g.gz * m.z;
    const double mm = dot3(m.x, m.x, m.y, m.y, m.z, m.z);
    t.ta = __builtin_fma(g.alpha_n * dm, dm, g.beta * mm);
    t.w = V3{-g.p_i.z * u.y + g.p_i.y * u.z, g.p_i.z * u.x - g.p_i.x * u.z, -g.p_i.y * u.x + g.p_i.x * u.y};
    return t;
}

// ---------------------------------------------------------------- voxel keys and hash
// residual-side key, KILO.cc:143-148: float cast, -1.0 for negatives, (int) truncation
__device__ __forceinline__ void key_trunc(V3 pw, const LkParams& pr, float* loc, int* key) {
    // x / voxel_size; when 1/voxel_size is exact (voxel_size a power of two, e.g. 0.5) the product is the
    // same correctly rounded quotient and saves three fp64 divides per point
    double q[3];
    if (pr.inv_vs_exact != 0.0) {
        q[0] = pw.x * pr.inv_vs_exact, q[1] = pw.y * pr.inv_vs_exact, q[2] = pw.z * pr.inv_vs_exact;
    } else {
        q[0] = pw.x / pr.voxel_size_d, q[1] = pw.y / pr.voxel_size_d, q[2] = pw.z / pr.voxel_size_d;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float l = (float)q[j];
        if (l < 0) l = (float)((double)l - 1.0);
        loc[j] = l;
        key[j] = (int)l;
    }
}
// insert-side key, eigen_types.hpp:89-95 with the float voxel_size of voxel_map.cc:337
__device__ __forceinline__ void key_floor(V3 pw, float vs_f, int* key) {
    double vs = (double)vs_f;
    key[0] = (int)floor(pw.x / vs);
    key[1] = (int)floor(pw.y / vs);
    key[2] = (int)floor(pw.z / vs);
}
// slot hash: Teschner products (eigen_types.hpp:79-82) followed by an avalanche so that linear
// probing in a power-of-two table does not cluster; std::unordered_map semantics need no more.
__host__ __device__ __forceinline__ unsigned int lk_hash3(int x, int y, int z) {
    unsigned int h = ((unsigned int)x * 73856093u) ^ ((unsigned int)y * 471943u) ^ ((unsigned int)z * 83492791u);
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    return h;
}
// Keeps a loaded 16-B chunk from being sunk below this point: the compiler moves a load next to its first use, and the uses of a
// match record sit behind three data-dependent tests (cell empty? plane? range gate) - the kernel's ISA had THREE dependent L2
// round trips per candidate (pad_ word; header + flags, then d / radius; the 80-B tail) where the layout was designed for one.
__device__ __forceinline__ void pin_chunk(double2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void pin_chunk(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pin_chunk(int4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
#ifndef LK_PIN_RECORD
#define LK_PIN_RECORD 1   // records and table entries are requested whole and PINNED ahead of the first test of any part of them (0: loads
                          // where the source has them - the compiler sinks them behind the tests; A/B)
#endif
// Linear probing, two consecutive slots per round trip: a wave waits for its slowest lane, and with ~14 distinct keys
// per wave at load factor 0.3 some lane almost always needs a second probe; both slots are requested together.
__device__ __forceinline__ int hash_find(const LkMap& m, int kx, int ky, int kz) {
    unsigned int s = lk_hash3(kx, ky, kz) & m.hash_mask;
    for (unsigned int probe = 0; probe <= m.hash_mask; probe += 2) {
        int4 e0 = m.hash[s];
        int4 e1 = m.hash[(s + 1) & m.hash_mask];
#if LK_PIN_RECORD
        pin_chunk(e0), pin_chunk(e1);   // really ONE round trip: left alone, the compiler fetches e0.w, then e0's key, then e1
#endif
        if (e0.w == LK_EMPTY) return -1;
        if (e0.w >= 0 && e0.x == kx && e0.y == ky && e0.z == kz) return e0.w;
        if (e1.w == LK_EMPTY) return -1;
        if (e1.w >= 0 && e1.x == kx && e1.y == ky && e1.z == kz) return e1.w;
        s = (s + 2) & m.hash_mask;
    }
    return -1;
}

// Broadcast of one lane's value when the SOURCE LANE IS A COMPILE-TIME CONSTANT: two v_readlane_b32 into an SGPR pair instead of
// the two ds_bpermute_b32 that __shfl compiles to - no trip through the LDS pipe (which the one-wave filter kernels saturate) and
// the result is a scalar operand.  Same bits as __shfl(v, lane).
#ifndef LK_READLANE
#define LK_READLANE 1
#endif
template <int LANE>
__device__ __forceinline__ double lane_bcast(double v) {
#if LK_READLANE
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE), hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
    return __hiloint2double(hi, lo);
#else
    return __shfl(v, LANE, LK_WAVE);
#endif
}
template <int LANE>
__device__ __forceinline__ int lane_bcast(int v) {
#if LK_READLANE
    return __builtin_amdgcn_readlane(v, LANE);
#else
    return __shfl(v, LANE, LK_WAVE);
#endif
}
// the same for a lane index that is constant after unrolling (i in a `#pragma unroll` loop): the builtin wants a uniform value
__device__ __forceinline__ double lane_bcast_u(double v, int lane) {
#if LK_READLANE
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
#else
    return __shfl(v, lane, LK_WAVE);
#endif
}
__device__ __forceinline__ int lane_bcast_u(int v, int lane) {
#if LK_READLANE
    return __builtin_amdgcn_readlane(v, lane);
#else
    return __shfl(v, lane, LK_WAVE);
#endif
}

// wave-wide sum, result in every lane: the xor butterfly 32, 16, 8, 4, 2, 1.  Built from gfx950's v_permlane32_swap / v_permlane16_swap
// (a swap of a register with a copy of itself leaves the two halves / row pairs side by side, and a + b == b + a) and DPP moves
// (row_ror:8 == lane ^ 8 inside a row of 16; row_half_mirror then quad_perm [3,2,1,0] == lane ^ 4; quad_perm for ^ 2 and ^ 1)
// instead of six pairs of ds_bpermute_b32: the same partners, the same additions, the same bits in every lane
// (tools/probes/wave_sum_dpp_butterfly.hip), no trip through the LDS pipe.  LK_DPP_SUM=0: the __shfl_xor form.
#ifndef LK_DPP_SUM
#define LK_DPP_SUM 1
#endif
typedef unsigned int lk_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
#if LK_DPP_SUM
    {
        const unsigned int lo = (unsigned int)__double2loint(v), hi = (unsigned int)__double2hiint(v);
        const lk_u2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const lk_u2 h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    {
        const unsigned int lo = (unsigned int)__double2loint(v), hi = (unsigned int)__double2hiint(v);
        const lk_u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const lk_u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    v += dpp_mov_f64<0x128>(v);
    v += dpp_mov_f64<0x1B>(dpp_mov_f64<0x141>(v));
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0xB1>(v);
    return v;
#else
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, LK_WAVE);
    return v;
#endif
}
