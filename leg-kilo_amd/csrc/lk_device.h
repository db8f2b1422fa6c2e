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

#define LK_WAVE 64
#define LK_EMPTY (-1)
#define LK_LOCKED (-2)
#define LK_MAX_LAYER 4
#define LK_NPART 32  // doubles per block partial: A(21) b(6) sumR(1) count(1) pad

// error bits in LkMap.counters[LK_CTR_ERR]
#define LK_E_HASH_FULL 1u
#define LK_E_NODES_FULL 2u
#define LK_E_BLOCKS_FULL 4u
#define LK_E_SCRATCH_FULL 8u

enum { LK_CTR_NODES = 0, LK_CTR_BLOCKS = 1, LK_CTR_ROOTS = 2, LK_CTR_ERR = 3, LK_CTR_TOUCHED = 4, LK_CTR_SCRATCH = 5,
       LK_CTR_COUNT = 8 };

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
    float voxel_size_f;          // float use at voxel_map.cc:289,337
    float range_var;             // dept_err^2 in float, voxel_map.cc:25
    float planer_threshold;      // voxel_map.h:140
    int max_layer, max_points_num;
    int layer_init_num[5];
};

struct LkMap {                   // device pointers of one voxel map, passed by value
    int4* hash;
    lk_plane_rec* planes;
    lk_node_rec* nodes;
    lk_block_rec* blocks;
    unsigned int* counters;      // LK_CTR_*
    int* touched;                // roots touched by the current bucket
    int* next;                   // per-point list links (bucket-local index)
    int* scratch;                // per-root gathered indices
    unsigned int hash_mask, max_nodes, max_blocks, max_scan;
};

// ---------------------------------------------------------------- small fp64 helpers
struct V3 {
    double x, y, z;
};
struct S3 {  // symmetric 3x3: xx xy xz yy yz zz
    double xx, xy, xz, yy, yz, zz;
};

__device__ __forceinline__ V3 mat3_mul_v(const double* M, V3 v) {  // row-major 3x3
    return V3{M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z,
              M[6] * v.x + M[7] * v.y + M[8] * v.z};
}
__device__ __forceinline__ V3 mat3T_mul_v(const double* M, V3 v) {
    return V3{M[0] * v.x + M[3] * v.y + M[6] * v.z, M[1] * v.x + M[4] * v.y + M[7] * v.z,
              M[2] * v.x + M[5] * v.y + M[8] * v.z};
}
__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
// C = A * S * A^T for symmetric S, result symmetric (upper triangle evaluated once)
__device__ __forceinline__ S3 congruence(const double* A, S3 s) {
    double T[9];  // T = A * S
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a0 = A[3 * i], a1 = A[3 * i + 1], a2 = A[3 * i + 2];
        T[3 * i + 0] = a0 * s.xx + a1 * s.xy + a2 * s.xz;
        T[3 * i + 1] = a0 * s.xy + a1 * s.yy + a2 * s.yz;
        T[3 * i + 2] = a0 * s.xz + a1 * s.yz + a2 * s.zz;
    }
    S3 r;
    r.xx = T[0] * A[0] + T[1] * A[1] + T[2] * A[2];
    r.xy = T[0] * A[3] + T[1] * A[4] + T[2] * A[5];
    r.xz = T[0] * A[6] + T[1] * A[7] + T[2] * A[8];
    r.yy = T[3] * A[3] + T[4] * A[4] + T[5] * A[5];
    r.yz = T[3] * A[6] + T[4] * A[7] + T[5] * A[8];
    r.zz = T[6] * A[6] + T[7] * A[7] + T[8] * A[8];
    return r;
}
__device__ __forceinline__ double quad3(S3 s, V3 n) {  // n^T S n, evaluated as (n^T S) n
    double t0 = n.x * s.xx + n.y * s.xy + n.z * s.xz;
    double t1 = n.x * s.xy + n.y * s.yy + n.z * s.yz;
    double t2 = n.x * s.xz + n.y * s.yz + n.z * s.zz;
    return t0 * n.x + t1 * n.y + t2 * n.z;
}
__device__ __forceinline__ void skew3(V3 v, double* K) {
    K[0] = 0.0, K[1] = -v.z, K[2] = v.y, K[3] = v.z, K[4] = 0.0, K[5] = -v.x, K[6] = -v.y, K[7] = v.x, K[8] = 0.0;
}
// Rodrigues: I + sin(a) K + (1-cos(a)) K K  (math_utils.hpp:19-68)
__device__ __forceinline__ void rodrigues3(V3 axis, double ang, double* R) {
    double K[9], KK[9];
    skew3(axis, K);
    mat3_mul(K, K, KK);
    double s = sin(ang), c1 = 1.0 - cos(ang);
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + s * K[i] + c1 * KK[i];
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

// calcBodyCov, voxel_map.cc:22-40 (range / range_var in float as in the reference)
__device__ __forceinline__ S3 calc_body_cov(V3 pb, const LkParams& pr) {
    if (pb.z == 0) pb.z = 0.0001;
    float range = (float)sqrt(pb.x * pb.x + pb.y * pb.y + pb.z * pb.z);
    double nrm = sqrt(pb.x * pb.x + pb.y * pb.y + pb.z * pb.z);
    V3 d = V3{pb.x / nrm, pb.y / nrm, pb.z / nrm};
    V3 b1 = V3{1.0, 1.0, -(d.x + d.y) / d.z};
    double n1 = sqrt(b1.x * b1.x + b1.y * b1.y + b1.z * b1.z);
    b1 = V3{b1.x / n1, b1.y / n1, b1.z / n1};
    V3 b2 = V3{b1.y * d.z - b1.z * d.y, b1.z * d.x - b1.x * d.z, b1.x * d.y - b1.y * d.x};
    double n2sq = b2.x * b2.x + b2.y * b2.y + b2.z * b2.z;
    if (n2sq > 0) {
        double n2 = sqrt(n2sq);
        b2 = V3{b2.x / n2, b2.y / n2, b2.z / n2};
    }
    // A = range * hat(d) * [b1 b2]   (3x2)
    double r = (double)range;
    double H[9];
    skew3(d, H);
#pragma unroll
    for (int i = 0; i < 9; ++i) H[i] = r * H[i];
    V3 a1 = mat3_mul_v(H, b1), a2 = mat3_mul_v(H, b2);
    double rv = (double)pr.range_var, dv = pr.dir_var;
    V3 dr = V3{d.x * rv, d.y * rv, d.z * rv};
    V3 a1v = V3{a1.x * dv, a1.y * dv, a1.z * dv}, a2v = V3{a2.x * dv, a2.y * dv, a2.z * dv};
    S3 c;
    c.xx = dr.x * d.x + (a1v.x * a1.x + a2v.x * a2.x);
    c.xy = dr.x * d.y + (a1v.x * a1.y + a2v.x * a2.y);
    c.xz = dr.x * d.z + (a1v.x * a1.z + a2v.x * a2.z);
    c.yy = dr.y * d.y + (a1v.y * a1.y + a2v.y * a2.y);
    c.yz = dr.y * d.z + (a1v.y * a1.z + a2v.y * a2.z);
    c.zz = dr.z * d.z + (a1v.z * a1.z + a2v.z * a2.z);
    return c;
}

// state-dependent constants of one bucket, derived from LkFilter by each thread (uniform)
struct BucketConst {
    double R[9], p[3], RE[9];
    S3 Prr, Ppp;
};
__device__ __forceinline__ void load_bucket_const(const LkFilter* __restrict__ f, const LkParams& pr, BucketConst& bc) {
#pragma unroll
    for (int i = 0; i < 9; ++i) bc.R[i] = f->x[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) bc.p[i] = f->x[9 + i];
    mat3_mul(bc.R, pr.ext_R, bc.RE);
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

// ---------------------------------------------------------------- voxel keys and hash
// residual-side key, KILO.cc:143-148: float cast, -1.0 for negatives, (int) truncation
__device__ __forceinline__ void key_trunc(V3 pw, double vs, float* loc, int* key) {
    double q[3] = {pw.x / vs, pw.y / vs, pw.z / vs};
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
__device__ __forceinline__ int hash_find(const LkMap& m, int kx, int ky, int kz) {
    unsigned int s = lk_hash3(kx, ky, kz) & m.hash_mask;
    for (unsigned int probe = 0; probe <= m.hash_mask; ++probe) {
        int4 e = m.hash[s];
        if (e.w == LK_EMPTY) return -1;
        if (e.w >= 0 && e.x == kx && e.y == ky && e.z == kz) return e.w;
        s = (s + 1) & m.hash_mask;
    }
    return -1;
}

// wave-wide sum, result in every lane
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, LK_WAVE);
    return v;
}
