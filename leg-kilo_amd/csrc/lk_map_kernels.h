// lk_map_kernels.h — voxel-map mutation: one WAVE (64 lanes) owns one root voxel.
//
//   lk_insert_root_kernel + lk_insert_apply_kernel
//                         UpdateVoxelMap / UpdateOctoTree (voxel_map.cc:336-361, :185-241) for the points
//                         the re-projection kernel queued on each touched root.  Points of one root are
//                         replayed in input order (successive-minimum selection over the root's list);
//                         different roots are independent, so the order ACROSS roots is free.
//   dev_init_plane        init_plane (voxel_map.cc:42-117): lanes stride the node's points, wave
//                         shuffle reductions give the centroid / scatter sums and the 21 unique terms
//                         of plane_var = sum_i J_i var_i J_i^T; every lane runs the same 3x3 Jacobi.
//   lk_build_* kernels    BuildVoxelMap (voxel_map.cc:287-334): first-frame variance formula (:306-307),
//                         points grouped per root by a stable radix sort, then init_octo_tree /
//                         cut_octo_tree (:119-183) per root with ping-pong index lists.
//
// Wave-synchronous programming: control flow is wave-uniform, lane 0 performs the scalar record
// writes, wave_fence() orders them before the other lanes' reads (same CU, workgroup scope).
#pragma once
#include "lk_device.h"
#include "lk_eig3.h"

#define LK_MB 256  // threads per block in the per-root kernels (4 waves = 4 roots in flight)
#ifndef LK_X_CHILD_CONST
#ifndef LK_X_ATTR
#define LK_X_ATTR 0   // attribution builds only (results WRONG by construction): 1 no plane_var, 2 no eigen-decomposition in the full fit, 4 no
                      // point_geom for the new points, 8 no refit-event tests, 16 no index sort - tools/gpu_ov_attr.sh times the overlay root pass on each
#endif
#define LK_X_CHILD_CONST 1   // apply_leaf: a child it has just created is known without reading it back (0: node_load, A/B)
#endif

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ int bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int t = __shfl_xor(v, o, LK_WAVE);
        v = t < v ? t : v;
    }
    return v;
}

// Sums of N per-lane values over the wave, all N results in every lane - like N calls of wave_sum, but as a TRANSPOSE-REDUCE: at
// the xor step m a lane keeps one half of its values and adds the partner's copy of that half (low lanes the first half, high lanes
// the second), so the count halves with every step: ceil(N/2) + ceil(N/4) + ... exchanges instead of 6 N (N = 21: 24 instead of
// 126; N = 9: 13 instead of 54), after which lane L holds the complete sum of ONE component, and N more shuffles gather them.
// The summation order differs from wave_sum's butterfly (a different, equally valid rounding of the same sum).
#ifndef LK_WSN_READLANE
#define LK_WSN_READLANE 0
#endif
template <int N>
__device__ __forceinline__ void wave_sum_n(double* v) {
    const int lane = threadIdx.x & 63;
    constexpr int H5 = (N + 1) / 2, H4 = (H5 + 1) / 2, H3 = (H4 + 1) / 2, H2 = (H3 + 1) / 2, H1 = (H2 + 1) / 2, H0 = (H1 + 1) / 2;
    auto step = [&](const int n, const int half, const int mask) {
        const bool hi = (lane & mask) != 0;
#pragma unroll
        for (int j = 0; j < half; ++j) {
            const double a = v[j];                              // first half  [0, half)
            const double b = (half + j < n) ? v[half + j] : 0.0;   // second half [half, n), zero-padded
            const double send = hi ? a : b, keep = hi ? b : a;
            v[j] = keep + __shfl_xor(send, mask, LK_WAVE);
        }
    };
    step(N, H5, 32);
    step(H5, H4, 16);
    step(H4, H3, 8);
    step(H3, H2, 4);
    step(H2, H1, 2);
    step(H1, H0, 1);
    // component c sits in v[0] of the lane whose bits select it: bit 5 adds H5, bit 4 adds H4, ... (compile-time per c)
    double r[N];
#pragma unroll
    for (int c = 0; c < N; ++c) {
        int rem = c, src = 0;
        if (rem >= H5) rem -= H5, src |= 32;
        if (rem >= H4) rem -= H4, src |= 16;
        if (rem >= H3) rem -= H3, src |= 8;
        if (rem >= H2) rem -= H2, src |= 4;
        if (rem >= H1) rem -= H1, src |= 2;
        if (rem >= H0) rem -= H0, src |= 1;
        // gathered with ds_bpermute, not v_readlane: 2 N scalar results at once (N = 21: 42 SGPRs on top of a kernel that already
        // spills scalars) gave wrong values in lk_insert_root_kernel - varying with unrelated code changes - while the vector form
        // is stable (round 3, tools A/B: -DLK_READLANE=0 fixed every failing test)
#if LK_WSN_READLANE
        r[c] = lane_bcast_u(v[0], src);    // A/B build only (tools/probes/readlane_gather): the v_readlane form of the gather
#else
        r[c] = __shfl(v[0], src, LK_WAVE);
#endif
    }
#pragma unroll
    for (int c = 0; c < N; ++c) v[c] = r[c];
}

struct PtU {  // one point, uniform across the wave
    double pw[3];
    double var[6];
};

__device__ __forceinline__ void load_pt(const lk_pt_rec* base, const int* idx, int j, double* pw, double* var) {
    const lk_pt_rec* r = &base[idx ? idx[j] : j];
#pragma unroll
    for (int c = 0; c < 3; ++c) pw[c] = r->pw[c];
#pragma unroll
    for (int c = 0; c < 6; ++c) var[c] = r->var[c];
}

// symmetric 3x3 eigen-solver; evecs columns = eigenvectors (stand-in for EigenSolver, voxel_map.cc:55).  Default: the closed form
// of lk_eig3.h (one plane fit is the serial work of one wave: ~0.5 us instead of the 3.8 us of 6-7 Jacobi sweeps); -DLK_EIG_JACOBI=1
// keeps the cyclic Jacobi iteration the oracle runs (A/B).
#ifndef LK_EIG_JACOBI
#define LK_EIG_JACOBI 0
#endif
// eigenvectors as three separate vectors (see lk_eig_sym3_cols: a 3 x 3 array picked by a run-time column index ends up in LDS / scratch)
__device__ __forceinline__ void eig_sym3_dev(const double* Ain, double* ev, double* V);
__device__ __forceinline__ void eig_sym3_cols_dev(const double* Ain, double* ev, double* v0, double* v1, double* v2) {
#if !LK_EIG_JACOBI
    lk_eig_sym3_cols(Ain, ev, v0, v1, v2);
#else
    double V[9];
    eig_sym3_dev(Ain, ev, V);
#pragma unroll
    for (int k = 0; k < 3; ++k) v0[k] = V[3 * k + 0], v1[k] = V[3 * k + 1], v2[k] = V[3 * k + 2];
#endif
}
__device__ __forceinline__ void eig_sym3_dev(const double* Ain, double* ev, double* V) {
#if !LK_EIG_JACOBI
    lk_eig_sym3(Ain, ev, V);
    return;
#endif
    double a[3][3] = {{Ain[0], Ain[1], Ain[2]}, {Ain[1], Ain[3], Ain[4]}, {Ain[2], Ain[4], Ain[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-300 || off <= 1e-34 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] != 0.0) {
                    double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                    double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double akp = a[k][p], akq = a[k][q];
                        a[k][p] = c * akp - s * akq;
                        a[k][q] = s * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double apk = a[p][k], aqk = a[q][k];
                        a[p][k] = c * apk - s * aqk;
                        a[q][k] = s * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        double vkp = v[k][p], vkq = v[k][q];
                        v[k][p] = c * vkp - s * vkq;
                        v[k][q] = s * vkp + c * vkq;
                    }
                }
            }
    }
    ev[0] = a[0][0], ev[1] = a[1][1], ev[2] = a[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[3 * i + j] = v[i][j];
}

// init_plane, voxel_map.cc:42-117.  Whole wave; returns is_plane (uniform).
__device__ __noinline__ bool dev_init_plane(lk_plane_rec* pl, lk_match_rec* mr, float planer_threshold, const lk_pt_rec* base,
                                            const int* idx, int count) {
    const int lane = threadIdx.x & 63;
    double s[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] = 0.0;
    for (int j = lane; j < count; j += LK_WAVE) {
        double pw[3], var[6];
        load_pt(base, idx, j, pw, var);
        s[0] += pw[0], s[1] += pw[1], s[2] += pw[2];
        s[3] += pw[0] * pw[0], s[4] += pw[0] * pw[1], s[5] += pw[0] * pw[2];
        s[6] += pw[1] * pw[1], s[7] += pw[1] * pw[2], s[8] += pw[2] * pw[2];
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) s[q] = wave_sum(s[q]);
    const double n = (double)count;
    double c[3] = {s[0] / n, s[1] / n, s[2] / n};
    double cov[6] = {s[3] / n - c[0] * c[0], s[4] / n - c[0] * c[1], s[5] / n - c[0] * c[2],
                     s[6] / n - c[1] * c[1], s[7] / n - c[1] * c[2], s[8] / n - c[2] * c[2]};
    double ev[3], V0[3], V1[3], V2[3];   // eigenvectors of ev[0], ev[1], ev[2]
    eig_sym3_cols_dev(cov, ev, V0, V1, V2);
    int imin = 0, imax = 0;   // first minimum / first maximum, tracked as scalars (ev[imin] with a run-time index keeps ev[] in memory)
    {
        double lo = ev[0], hi = ev[0];
        if (ev[1] < lo) imin = 1, lo = ev[1];
        if (ev[1] > hi) imax = 1, hi = ev[1];
        if (ev[2] < lo) imin = 2, lo = ev[2];
        if (ev[2] > hi) imax = 2, hi = ev[2];
    }
    int imid = 3 - imin - imax;
    if (imid > 2) imid = imin;
    const bool is_plane = ev[imin] < (double)planer_threshold;
    double acc[21];
#pragma unroll
    for (int q = 0; q < 21; ++q) acc[q] = 0.0;
    // select eigen-columns without dynamic register indexing
    auto col = [&](int k, double* o) {
        o[0] = (k == 0) ? V0[0] : (k == 1) ? V1[0] : V2[0];
        o[1] = (k == 0) ? V0[1] : (k == 1) ? V1[1] : V2[1];
        o[2] = (k == 0) ? V0[2] : (k == 1) ? V1[2] : V2[2];
    };
    auto evk = [&](int k) { return (k == 0) ? ev[0] : (k == 1) ? ev[1] : ev[2]; };
    double vmin[3], vmid[3], vmax[3];
    col(imin, vmin), col(imid, vmid), col(imax, vmax);
    const double emin = evk(imin), emid = evk(imid), emax = evk(imax);
    if (is_plane) {
        // rhs_m = v_m v_min^T + v_min v_m^T and den_m = n (lambda_min - lambda_m), for the two m != min
        double rhsA[9], rhsB[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                rhsA[3 * r + cc] = vmid[r] * vmin[cc] + vmin[r] * vmid[cc];
                rhsB[3 * r + cc] = vmax[r] * vmin[cc] + vmin[r] * vmax[cc];
            }
        const double denA = count * (emin - emid), denB = count * (emin - emax);
        const double invn = 1.0 / count;
        for (int j = lane; j < count; j += LK_WAVE) {
            double pw[3], var[6];
            load_pt(base, idx, j, pw, var);
            double q[3] = {pw[0] - c[0], pw[1] - c[1], pw[2] - c[2]};
            double la[3] = {q[0] / denA, q[1] / denA, q[2] / denA};
            double lb[3] = {q[0] / denB, q[1] / denB, q[2] / denB};
            double FA[3], FB[3];  // F rows of eigen-columns mid / max; the row of min is zero
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                FA[cc] = la[0] * rhsA[cc] + la[1] * rhsA[3 + cc] + la[2] * rhsA[6 + cc];
                FB[cc] = lb[0] * rhsB[cc] + lb[1] * rhsB[3 + cc] + lb[2] * rhsB[6 + cc];
            }
            double J[6][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    J[r][cc] = vmid[r] * FA[cc] + vmax[r] * FB[cc];  // evecs * F
                    J[3 + r][cc] = (r == cc) ? invn : 0.0;
                }
            double Sv[3][3] = {{var[0], var[1], var[2]}, {var[1], var[3], var[4]}, {var[2], var[4], var[5]}};
            double JV[6][3];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) JV[r][cc] = J[r][0] * Sv[0][cc] + J[r][1] * Sv[1][cc] + J[r][2] * Sv[2][cc];
            int k = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = r; cc < 6; ++cc) acc[k++] += JV[r][0] * J[cc][0] + JV[r][1] * J[cc][1] + JV[r][2] * J[cc][2];
        }
#pragma unroll
        for (int q = 0; q < 21; ++q) acc[q] = wave_sum(acc[q]);
    }
    if (lane == 0) {
        pl->points_size = count;
        if (is_plane) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pl->center[k] = c[k], pl->normal[k] = vmin[k];
#pragma unroll
            for (int q = 0; q < 21; ++q) pl->plane_var[q] = acc[q];
            pl->min_eigen_value = (float)emin;
            pl->mid_eigen_value = (float)emid;
            pl->max_eigen_value = (float)emax;
            pl->radius = (float)sqrt(emax);
            pl->d = (float)(-(vmin[0] * c[0] + vmin[1] * c[1] + vmin[2] * c[2]));
            pl->flags = LK_PLANE_IS_PLANE | LK_PLANE_IS_INIT;
            lk_derive_match(pl, mr);  // compact copy for the residual kernel
        } else {
            pl->flags = pl->flags & ~LK_PLANE_IS_PLANE;
            mr->flags = pl->flags;
        }
    }
    wave_fence();
    return is_plane;
}

// ---- register-resident plane fit (one point per lane, count <= 64): the same formulas, in the same order, as
// dev_init_plane, split into the cheap test (centroid, scatter, eigen-decomposition -> is_plane) and the expensive
// plane_var accumulation, so that a chain of refit events inside one bucket costs one full fit, not one per event.
struct PlaneFit {
    bool is_plane;
    double s9[9];   // the moment sums (sum p, sum p p^T) the fit was made from: the full fit of a leaf reuses those of its last event
    double c[3];
    double emin, emid, emax;
    double vmin[3], vmid[3], vmax[3];
};
// decide_only: just is_plane = (lambda_min < threshold), decided WITHOUT an eigen-decomposition: lambda_min >= t  <=>
// cov - t I is positive semi-definite, and it is positive definite iff its three leading principal minors are > 0
// (Sylvester).  Used for the refit events in the middle of a bucket, whose eigenvectors nobody ever reads; differs
// from the Jacobi decision only when lambda_min equals the threshold to rounding.
// REUSE: the sums come from `prev` (a compile-time choice: with a run-time pointer that may be null the nine doubles became a private
// ARRAY - promoted to LDS, 72 B x 256 threads, or spilled to scratch - and the insert kernels ran 30-40 % longer for it)
template <bool decide_only = false, bool REUSE = false>
__device__ __forceinline__ PlaneFit plane_test_regs(const double* pw, bool active, int count, float planer_threshold,
                                                    const PlaneFit* prev = nullptr) {
    double s[9];
    if (REUSE) {   // same points, same count as the event these sums come from: the nine wave reductions are not repeated
#pragma unroll
        for (int q = 0; q < 9; ++q) s[q] = prev->s9[q];
    } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) s[q] = 0.0;
        if (active) {
            s[0] += pw[0], s[1] += pw[1], s[2] += pw[2];
            s[3] += pw[0] * pw[0], s[4] += pw[0] * pw[1], s[5] += pw[0] * pw[2];
            s[6] += pw[1] * pw[1], s[7] += pw[1] * pw[2], s[8] += pw[2] * pw[2];
        }
        wave_sum_n<9>(s);
    }
    const double n = (double)count;
    PlaneFit f;
#pragma unroll
    for (int q = 0; q < 9; ++q) f.s9[q] = s[q];
    f.c[0] = s[0] / n, f.c[1] = s[1] / n, f.c[2] = s[2] / n;
    double cov[6] = {s[3] / n - f.c[0] * f.c[0], s[4] / n - f.c[0] * f.c[1], s[5] / n - f.c[0] * f.c[2],
                     s[6] / n - f.c[1] * f.c[1], s[7] / n - f.c[1] * f.c[2], s[8] / n - f.c[2] * f.c[2]};
    if (decide_only) {
        const double t = (double)planer_threshold;
        const double b11 = cov[0] - t, b22 = cov[3] - t, b33 = cov[5] - t, bxy = cov[1], bxz = cov[2], byz = cov[4];
        const double m2 = b11 * b22 - bxy * bxy;
        const double m3 = b11 * (b22 * b33 - byz * byz) - bxy * (bxy * b33 - byz * bxz) + bxz * (bxy * byz - b22 * bxz);
        f.is_plane = !(b11 > 0.0 && m2 > 0.0 && m3 > 0.0);
        f.emin = f.emid = f.emax = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) f.vmin[k] = f.vmid[k] = f.vmax[k] = 0.0;
        return f;
    }
    double ev[3], V0[3], V1[3], V2[3];   // eigenvectors of ev[0], ev[1], ev[2]
    eig_sym3_cols_dev(cov, ev, V0, V1, V2);
    int imin = 0, imax = 0;   // first minimum / first maximum, tracked as scalars (ev[imin] with a run-time index keeps ev[] in memory)
    {
        double lo = ev[0], hi = ev[0];
        if (ev[1] < lo) imin = 1, lo = ev[1];
        if (ev[1] > hi) imax = 1, hi = ev[1];
        if (ev[2] < lo) imin = 2, lo = ev[2];
        if (ev[2] > hi) imax = 2, hi = ev[2];
    }
    int imid = 3 - imin - imax;
    if (imid > 2) imid = imin;
    auto col = [&](int k, double* o) {
        o[0] = (k == 0) ? V0[0] : (k == 1) ? V1[0] : V2[0];
        o[1] = (k == 0) ? V0[1] : (k == 1) ? V1[1] : V2[1];
        o[2] = (k == 0) ? V0[2] : (k == 1) ? V1[2] : V2[2];
    };
    auto evk = [&](int k) { return (k == 0) ? ev[0] : (k == 1) ? ev[1] : ev[2]; };
    col(imin, f.vmin), col(imid, f.vmid), col(imax, f.vmax);
    f.emin = evk(imin), f.emid = evk(imid), f.emax = evk(imax);
    f.is_plane = f.emin < (double)planer_threshold;
    return f;
}
// plane_var = sum_i J_i var_i J_i^T over the active lanes (voxel_map.cc:76-95), 21 unique terms.
// Round 5: J_i = [A_i ; I / n] with A_i = v_mid FA_i^T + v_max FB_i^T (rank 2: the row of v_min is zero),
//   FA_i = ((q.v_mid) v_min + (q.v_min) v_mid) / den_A,  FB_i likewise with v_max,  q = p_i - centre.
// With a = var FA, b = var FB:  A var A^T = v_mid v_mid^T (FA.a) + (v_mid v_max^T + v_max v_mid^T) (FA.b) + v_max v_max^T (FB.b)  and  A var = v_mid a^T + v_max b^T,
// so a lane contributes 15 numbers (a, b, the three dot products, var) instead of the 21 entries of a 6 x 3 x 3 x 6 product, the wave reduces
// 15 sums instead of 21, and the 21 entries are composed once from the sums.  The same value (a different, equally valid rounding of the
// same sum); a third of the arithmetic and ~50 registers fewer at the root pass's peak.
__device__ __forceinline__ void plane_var_regs(const PlaneFit& f, const double* pw, const double* var, bool active, int count,
                                               double* acc) {
    const double invA = 1.0 / (count * (f.emin - f.emid)), invB = 1.0 / (count * (f.emin - f.emax));
    const double invn = 1.0 / count;
    double s15[15];   // a (3), b (3), FA.a, FA.b, FB.b, var (6)
#pragma unroll
    for (int q = 0; q < 15; ++q) s15[q] = 0.0;
    if (active) {
        const double q0 = pw[0] - f.c[0], q1 = pw[1] - f.c[1], q2 = pw[2] - f.c[2];
        const double dmin = q0 * f.vmin[0] + q1 * f.vmin[1] + q2 * f.vmin[2];
        const double dmid = (q0 * f.vmid[0] + q1 * f.vmid[1] + q2 * f.vmid[2]) * invA;
        const double dmax = (q0 * f.vmax[0] + q1 * f.vmax[1] + q2 * f.vmax[2]) * invB;
        const double dmA = dmin * invA, dmB = dmin * invB;
        double FA[3], FB[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) FA[c] = dmid * f.vmin[c] + dmA * f.vmid[c], FB[c] = dmax * f.vmin[c] + dmB * f.vmax[c];
        s15[0] = var[0] * FA[0] + var[1] * FA[1] + var[2] * FA[2], s15[1] = var[1] * FA[0] + var[3] * FA[1] + var[4] * FA[2],
        s15[2] = var[2] * FA[0] + var[4] * FA[1] + var[5] * FA[2];
        s15[3] = var[0] * FB[0] + var[1] * FB[1] + var[2] * FB[2], s15[4] = var[1] * FB[0] + var[3] * FB[1] + var[4] * FB[2],
        s15[5] = var[2] * FB[0] + var[4] * FB[1] + var[5] * FB[2];
        s15[6] = FA[0] * s15[0] + FA[1] * s15[1] + FA[2] * s15[2];
        s15[7] = FA[0] * s15[3] + FA[1] * s15[4] + FA[2] * s15[5];
        s15[8] = FB[0] * s15[3] + FB[1] * s15[4] + FB[2] * s15[5];
#pragma unroll
        for (int c = 0; c < 6; ++c) s15[9 + c] = var[c];
    }
    wave_sum_n<15>(s15);
    const double saa = s15[6], sab = s15[7], sbb = s15[8];
    int kk = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int cc = r; cc < 3; ++cc)
            acc[kk++] = f.vmid[r] * f.vmid[cc] * saa + (f.vmid[r] * f.vmax[cc] + f.vmax[r] * f.vmid[cc]) * sab + f.vmax[r] * f.vmax[cc] * sbb;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) acc[kk++] = (f.vmid[r] * s15[cc] + f.vmax[r] * s15[3 + cc]) * invn;
    }
    const double invn2 = invn * invn;
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[15 + c] = s15[9 + c] * invn2;
}
// lane 0 writes the plane and its compact match copy, both from registers (the match record is derived from the values, not read
// back from the plane record just stored).  No fence: nothing in the apply pass reads a plane it has committed.
template <bool ALL_LANES = false>   // ALL_LANES: every lane commits ITS OWN plane (lk_ov_insert_lane_kernel: one root voxel per lane)
__device__ __forceinline__ void plane_commit(lk_plane_rec* pl, lk_match_rec* mr, const PlaneFit& f, const double* acc, int count) {
    if (ALL_LANES || (threadIdx.x & 63) == 0) {
        pl->points_size = count;
        if (f.is_plane) {
            lk_plane_rec t;   // in registers
#pragma unroll
            for (int k = 0; k < 3; ++k) t.center[k] = f.c[k], t.normal[k] = f.vmin[k];
#pragma unroll
            for (int q = 0; q < 21; ++q) t.plane_var[q] = acc[q];
            t.radius = (float)sqrt(f.emax);
            t.d = (float)(-(f.vmin[0] * f.c[0] + f.vmin[1] * f.c[1] + f.vmin[2] * f.c[2]));
            t.flags = LK_PLANE_IS_PLANE | LK_PLANE_IS_INIT;
            lk_match_rec m;
            lk_derive_match(&t, &m);
#pragma unroll
            for (int k = 0; k < 3; ++k) pl->center[k] = t.center[k], pl->normal[k] = t.normal[k];
#pragma unroll
            for (int q = 0; q < 21; ++q) pl->plane_var[q] = t.plane_var[q];
            pl->min_eigen_value = (float)f.emin;
            pl->mid_eigen_value = (float)f.emid;
            pl->max_eigen_value = (float)f.emax;
            pl->radius = t.radius;
            pl->d = t.d;
            pl->flags = t.flags;
#pragma unroll
            for (int k = 0; k < 3; ++k) mr->center[k] = m.center[k], mr->normal[k] = m.normal[k], mr->w[k] = m.w[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) mr->s11[k] = m.s11[k];
            mr->d = m.d, mr->radius = m.radius, mr->flags = m.flags, mr->pad_ = 0, mr->s22 = m.s22;
        } else {
            const unsigned int fl = pl->flags & ~LK_PLANE_IS_PLANE;
            pl->flags = fl;
            mr->flags = fl;
        }
    }
}

// ---- node helpers (wave-uniform; lane 0 writes)
// Point-block pool.  Blocks retired by a freeze / cut during bucket k go to `freed_next`; lk_bucket_begin_kernel
// moves them to `free_list` before bucket k+1, so within one launch the free list is only ever popped (a signed
// counter; a failed pop restores it) and the retired list only ever pushed: no ABA, no lock.
__device__ __forceinline__ int pop_or_bump_block(const LkMap& m) {
    int* fc = reinterpret_cast<int*>(&m.counters[LK_CTR_FREE]);
    const int k = atomicSub(fc, 1) - 1;
    if (k >= 0) return m.free_list[k];
    atomicAdd(fc, 1);
    unsigned int b = atomicAdd(&m.counters[LK_CTR_BLOCKS], 1u);
    if (b >= m.max_blocks) {
        atomicOr(&m.counters[LK_CTR_ERR], LK_E_BLOCKS_FULL);
        b = m.max_blocks - 1;  // keep memory safe; the error flag fails the call
    }
    return (int)b;
}
__device__ __forceinline__ void retire_block(const LkMap& m, int block) {  // one lane
    if (block < 0) return;
    const unsigned int p = atomicAdd(&m.counters[LK_CTR_FREED], 1u);
    if (p < m.max_blocks) m.freed_next[p] = block;
}
__device__ __forceinline__ int alloc_block(const LkMap& m) {
    int id = -1;
    if ((threadIdx.x & 63) == 0) id = pop_or_bump_block(m);
    return bcast0(id);
}
__device__ __forceinline__ int create_child(const LkMap& m, int parent, int oct, const double* pcenter, float pquater,
                                            int player) {
    int id = -1;
    if ((threadIdx.x & 63) == 0) {
        unsigned int n = atomicAdd(&m.counters[LK_CTR_NODES], 1u);
        if (n >= m.max_nodes) {
            atomicOr(&m.counters[LK_CTR_ERR], LK_E_NODES_FULL);
            n = m.max_nodes - 1;
        }
        id = (int)n;
        lk_node_rec* nd = &m.nodes[id];
        int xyz[3] = {(oct >> 2) & 1, (oct >> 1) & 1, oct & 1};
#pragma unroll
        for (int c = 0; c < 8; ++c) nd->child[c] = -1;
#pragma unroll
        for (int c = 0; c < 3; ++c) nd->voxel_center[c] = pcenter[c] + (double)((float)(2 * xyz[c] - 1) * pquater);
        nd->quater_length = pquater / 2;
        nd->layer = player + 1;
        nd->npts = 0;
        nd->new_points = 0;
        nd->state = LK_NODE_UPDATE_ENABLE;
        nd->block = -1;
        nd->list_head = -1;
        nd->pad_[0] = 0;
        m.planes[id].flags = 0;
        m.match[id].flags = 0;
        m.nodes[parent].child[oct] = id;
    }
    id = bcast0(id);
    wave_fence();
    return id;
}
__device__ __forceinline__ int octant_of(const double* pw, const double* center) {
    return ((pw[0] > center[0]) ? 4 : 0) + ((pw[1] > center[1]) ? 2 : 0) + ((pw[2] > center[2]) ? 1 : 0);
}

// registers holding the mutable scalars of the node being processed
struct NodeRegs {
    int npts, new_points, block, layer;
    unsigned int state;
};
__device__ __forceinline__ NodeRegs node_load(const lk_node_rec* nd) {
    NodeRegs r;
    r.npts = bcast0(nd->npts), r.new_points = bcast0(nd->new_points), r.block = bcast0(nd->block);
    r.layer = bcast0(nd->layer), r.state = (unsigned int)bcast0((int)nd->state);
    return r;
}
__device__ __forceinline__ void node_store(lk_node_rec* nd, const NodeRegs& r, const bool fence = true) {
    if ((threadIdx.x & 63) == 0) {
        nd->npts = r.npts, nd->new_points = r.new_points, nd->block = r.block, nd->state = r.state;
    }
    if (fence) wave_fence();   // fence = false: nothing in this wave reads the record again (end of a group of the apply pass)
}
// temp_points_.push_back(pv)
__device__ __forceinline__ void node_push(const LkMap& m, NodeRegs& r, const PtU& pt) {
    if (r.block < 0) r.block = alloc_block(m);
    if ((threadIdx.x & 63) == 0 && r.npts < LK_BLOCK_PTS) {
        lk_pt_rec* dst = &m.blocks[r.block].pts[r.npts];
#pragma unroll
        for (int c = 0; c < 3; ++c) dst->pw[c] = pt.pw[c];
#pragma unroll
        for (int c = 0; c < 6; ++c) dst->var[c] = pt.var[c];
    }
    r.npts += 1;
    wave_fence();
}
__device__ __forceinline__ void node_drop_block(const LkMap& m, NodeRegs& r) {  // temp_points_ will never be read again
    if ((threadIdx.x & 63) == 0) retire_block(m, r.block);
    r.block = -1;
}
__device__ __forceinline__ void node_freeze(const LkMap& m, NodeRegs& r) {  // update_enable_=false; swap(temp_points_); new_points_=0
    r.state &= ~LK_NODE_UPDATE_ENABLE;
    r.npts = 0;
    r.new_points = 0;
    node_drop_block(m, r);
}

// init_octo_tree + cut_octo_tree for a node whose points live in its block (voxel_map.cc:119-183).
template <int L>
__device__ __noinline__ void dev_init_octo(const LkMap m, const LkParams pr, int node) {
    lk_node_rec* nd = &m.nodes[node];
    NodeRegs r = node_load(nd);
    const int thr = pr.layer_init_num[L];
    if (!(r.npts > thr)) return;
    const bool is_plane = dev_init_plane(&m.planes[node], &m.match[node], pr.planer_threshold, m.blocks[r.block].pts, nullptr, r.npts);
    if (is_plane) {
        r.state &= ~LK_NODE_OCTO_STATE;
        if (r.npts > pr.max_points_num) node_freeze(m, r);
    } else {
        r.state |= LK_NODE_OCTO_STATE;
        if (L >= pr.max_layer) {
            r.state &= ~LK_NODE_OCTO_STATE;  // cut_octo_tree returns at once (voxel_map.cc:140-143)
        } else if constexpr (L < LK_MAX_LAYER) {
            // distribute the points to the octants in order (voxel_map.cc:144-161)
            double center[3] = {nd->voxel_center[0], nd->voxel_center[1], nd->voxel_center[2]};
            const float quater = nd->quater_length;
            const lk_pt_rec* src = m.blocks[r.block].pts;
            for (int j = 0; j < r.npts; ++j) {
                PtU pt;
                load_pt(src, nullptr, j, pt.pw, pt.var);
                int oct = octant_of(pt.pw, center);
                int child = bcast0(nd->child[oct]);
                if (child < 0) child = create_child(m, node, oct, center, quater, L);
                NodeRegs cr = node_load(&m.nodes[child]);
                node_push(m, cr, pt);
                cr.new_points += 1;
                node_store(&m.nodes[child], cr);
            }
            for (int ci = 0; ci < 8; ++ci) {  // voxel_map.cc:162-182
                int child = bcast0(nd->child[ci]);
                if (child < 0) continue;
                int cn = bcast0(m.nodes[child].npts);
                if (cn > pr.layer_init_num[L + 1]) dev_init_octo<L + 1>(m, pr, child);
            }
            node_drop_block(m, r);  // the parent's own points are never read again (dead)
        }
    }
    r.state |= LK_NODE_INIT_OCTO;
    r.new_points = 0;
    node_store(nd, r);
}

// UpdateOctoTree (voxel_map.cc:185-241) for one point; descends iteratively.
__device__ __forceinline__ void dev_update_octo(const LkMap& m, const LkParams& pr, int root, const PtU& pt) {
    int node = root;
    for (int depth = 0; depth <= LK_MAX_LAYER; ++depth) {
        lk_node_rec* nd = &m.nodes[node];
        NodeRegs r = node_load(nd);
        const bool is_plane = (bcast0((int)m.planes[node].flags) & (int)LK_PLANE_IS_PLANE) != 0;
        if (!(r.state & LK_NODE_INIT_OCTO)) {
            r.new_points += 1;
            node_push(m, r, pt);
            node_store(nd, r);
            if (r.npts > pr.layer_init_num[r.layer]) {
                switch (r.layer) {
                    case 0: dev_init_octo<0>(m, pr, node); break;
                    case 1: dev_init_octo<1>(m, pr, node); break;
                    case 2: dev_init_octo<2>(m, pr, node); break;
                    case 3: dev_init_octo<3>(m, pr, node); break;
                    default: dev_init_octo<4>(m, pr, node); break;
                }
            }
            return;
        }
        if (is_plane) {
            if (r.state & LK_NODE_UPDATE_ENABLE) {
                r.new_points += 1;
                node_push(m, r, pt);
                if (r.new_points > 5) {  // update_size_threshold_, voxel_map.h:158
                    const bool still = dev_init_plane(&m.planes[node], &m.match[node], pr.planer_threshold, m.blocks[r.block].pts, nullptr, r.npts);
                    r.new_points = 0;
                    // a refit that turns the node into a non-plane below max_layer makes later points descend
                    // to children (voxel_map.cc:205-223): its own temp_points_ are never read again
                    if (!still && r.layer < pr.max_layer) node_drop_block(m, r);
                }
                if (r.npts >= pr.max_points_num) node_freeze(m, r);
                node_store(nd, r);
            }
            return;
        }
        if (r.layer < pr.max_layer) {
            double center[3] = {nd->voxel_center[0], nd->voxel_center[1], nd->voxel_center[2]};
            int oct = octant_of(pt.pw, center);
            int child = bcast0(nd->child[oct]);
            if (child < 0) child = create_child(m, node, oct, center, nd->quater_length, r.layer);
            node = child;
            continue;
        }
        if (r.state & LK_NODE_UPDATE_ENABLE) {
            r.new_points += 1;
            if (r.state & LK_NODE_PTS_DROPPED) {
                r.npts += 1;  // count only: > max_points_num already, frozen below without a refit
            } else {
                node_push(m, r, pt);
            }
            if (r.new_points > 5 && !(r.state & LK_NODE_PTS_DROPPED)) {
                dev_init_plane(&m.planes[node], &m.match[node], pr.planer_threshold, m.blocks[r.block].pts, nullptr, r.npts);
                r.new_points = 0;
            }
            if (r.npts > pr.max_points_num) {
                node_freeze(m, r);
                r.state &= ~LK_NODE_PTS_DROPPED;
            }
            node_store(nd, r);
        }
        return;
    }
}

// ------------------------------------------------------------------ the ordered insert
// FROM_PV = false: points are re-derived from the scan (lk_point) and the post-update state - the same inlined transform the
//                  re-projection kernel hashed them with (bit-identical);
// FROM_PV = true : points are caller-supplied pointWithVar records (VoxelMapManager::UpdateVoxelMap).
// Work unit = one LEAF GROUP: the points of this bucket that land in the same leaf (or in the same not-yet-existing child) of one
// root voxel, in input order.  Groups of one root touch disjoint subtrees (distinct leaves; distinct child slots of a parent), so
// they are independent; the order INSIDE a group is the input order, which the reference's result depends on.
//   lk_insert_root_kernel    one WAVE per touched root.  A root that only needs its few points appended (an un-initialised root
//                            that stays below layer_init_num, a plane root that reaches neither its 6th new point - refit,
//                            voxel_map.cc:195 - nor max_points_num - freeze, :199) is finished at once, one lane per point.
//                            Otherwise the wave sorts the root's queued indices in registers, walks every point (read-only) to
//                            its target and forms the groups with ballots.  ONE group (98 % of the roots) or up to
//                            LK_INLINE_GROUPS: the wave applies them right away, one after the other - the points, the leaf's
//                            counters and its block id are already in registers, so nothing is handed over through memory.
//                            More groups: one descriptor per group + its indices are emitted.
//   lk_insert_apply_kernel   one wave per EMITTED group: the same register simulation of voxel_map.cc:186-237 (apply_leaf).  A root
//                            with six leaf groups is six work items side by side; replayed one after the other by the root's wave
//                            they set the kernel's duration (80-150 k cycles against a mean of 36 k).
//   lk_insert_fallback_kernel  the generic per-point state machine for what the simulation hands over (cuts, leftovers, long lists).
// Round 3: the root kernel replaces three launches (a thread-per-root light pass, a group pass, an apply pass that re-read
// descriptor, indices, node record and scan points of every group): the insert of a bucket is a chain of dependent memory round
// trips at a few waves per CU, and each hand-over through memory was another two or three of them.
#ifdef LK_DEBUG_INS
// DEBUG BUILD ONLY (-DLK_DEBUG_INS): 100 MHz stamps per phase of apply_leaf, summed over all groups; [15] = groups
__device__ unsigned long long lk_ins_dbg[16];
#define INS_STAMP(k) do { const unsigned long long t1_ = wall_clock64(); ph_[k] += t1_ - t0_; if ((threadIdx.x & 63) == 0) atomicAdd(&lk_ins_dbg[k], t1_ - t0_); t0_ = t1_; } while (0)
__device__ unsigned long long lk_slow_dbg[2 * 16];   // per-phase sums of the groups slower / faster than 20 us; [15] = their number
// per-root durations of the root kernel's waves: 6 rows x 32 bins of 2 us; rows: light / one group, applied inline / several groups,
// emitted / long list / whole wave (all its roots) / the root's first phase (record + slot line loaded, indices sorted, walk done)
__device__ unsigned int lk_root_hist[6 * 32];
__device__ unsigned int lk_ev_hist[16 * 32];   // apply_leaf: [fit events of the group][duration, 2-us bins]
#define ROOT_HIST(row, t_from) do { if ((threadIdx.x & 63) == 0) { unsigned long long d_ = (wall_clock64() - (t_from)) / 200ull; atomicAdd(&lk_root_hist[(row) * 32 + (int)(d_ > 31ull ? 31ull : d_)], 1u); } } while (0)
#else
#define INS_STAMP(k) do { } while (0)
#define ROOT_HIST(row, t_from) do { } while (0)
#endif
#ifdef LK_DEBUG_LI
__device__ unsigned long long lk_li_dbg[64];
#endif
struct LkGroup {       // 64 B
    int leaf;          // target leaf, or -1: child `oct` of `parent` has to be created first
    int parent, oct;
    int off, count;    // indices map.gidx[off .. off+count) in input order; LONG: map.scratch[off .. off+count) unsorted
    int kind;          // 0 = leaf group, 1 = LONG (a root with more than 64 queued points: per-point replay), 2 = fallback item
    int root, pad;     // kind 2: pad = layer of the leaf
    // the leaf as the group pass saw it (leaf >= 0).  Nothing else writes the leaf before its group is applied: groups are disjoint
    int npts, new_points, block, layer;
    unsigned int state;
    int is_plane, pad2[2];
};
static_assert(sizeof(LkGroup) == 64, "group descriptor must be 64 B");
#ifndef LK_INLINE_GROUPS
#define LK_INLINE_GROUPS 3   // roots with at most this many leaf groups are finished by their own wave, group after group (>= 1)
#endif
// Round 6: the four header words {leaf, block, cnt, decided} live in a DENSE int4 array of their own (LkOverlay::jobhdr, same [g][t] index): the fit passes look at
// every touched root's three job slots and nearly all are empty - at a 96-B stride that scan was half of their HBM traffic.  The words below stay for the layout.
struct LkFitJob {   // 96 B: a plane fit the overlay replay's root pass leaves to lk_ov_fit_lane_kernel (apply_leaf<DEFER>)
    int leaf, block, cnt, decided;   // (header, now in jobhdr) cnt = points of the leaf's last refit event (0: no fit), block = where they are, decided = is_plane of that event
    double s9[9];                    // its moment sums (sum p, sum p p^T)
    int base_block, n_base;          // a SPLIT leaf (lk_ov_root_lane_kernel): its first n_base points are in the base map's block base_block; else n_base = 0
};
static_assert(sizeof(LkFitJob) == 96, "fit job must be 96 B");
struct LeafInfo {
    int npts, new_points, block, layer;
    unsigned int state;
    int is_plane;
};

template <bool FROM_PV>
__device__ __forceinline__ void insert_point_pw(const LkParams& pr, const BucketConst& bc, const lk_point* __restrict__ pts,
                                                const lk_pt_rec* __restrict__ pv, int idx, double* pw, float4& p4) {
    if (FROM_PV) {
        pw[0] = pv[idx].pw[0], pw[1] = pv[idx].pw[1], pw[2] = pv[idx].pw[2];
    } else {  // the same inlined transform as everywhere else (identical bits)
        const float4 p = reinterpret_cast<const float4*>(pts)[idx];
        p4 = p;
        V3 pb = V3{(double)p.x, (double)p.y, (double)p.z};
        V3 e = mat3_mul_v(pr.ext_R, pb);
        V3 pi = V3{e.x + pr.ext_T[0], e.y + pr.ext_T[1], e.z + pr.ext_T[2]};
        V3 w = mat3_mul_v(bc.R, pi);
        pw[0] = w.x + bc.p[0], pw[1] = w.y + bc.p[1], pw[2] = w.z + bc.p[2];
    }
}
__device__ __forceinline__ void insert_point_pw_of(const LkParams& pr, const BucketConst& bc, const float4& p, double* pw) {   // insert_point_pw's transform of a point in hand
    V3 pb = V3{(double)p.x, (double)p.y, (double)p.z};
    V3 e = mat3_mul_v(pr.ext_R, pb);
    V3 pi = V3{e.x + pr.ext_T[0], e.y + pr.ext_T[1], e.z + pr.ext_T[2]};
    V3 w = mat3_mul_v(bc.R, pi);
    pw[0] = w.x + bc.p[0], pw[1] = w.y + bc.p[1], pw[2] = w.z + bc.p[2];
}
__device__ __forceinline__ void geom_to_pt(const PointGeom& g, PtU& pt) {
    pt.pw[0] = g.p_w.x, pt.pw[1] = g.p_w.y, pt.pw[2] = g.p_w.z;
    pt.var[0] = g.var.xx, pt.var[1] = g.var.xy, pt.var[2] = g.var.xz;
    pt.var[3] = g.var.yy, pt.var[4] = g.var.yz, pt.var[5] = g.var.zz;
}
// groups that need the generic per-point code are queued behind the descriptors (kind 1: whole long list; kind 2: leaf `leaf`,
// optional init_octo_tree (oct != 0, layer in pad), then gidx[off .. off+count) one by one)
__device__ __forceinline__ void insert_defer(const LkMap& map, int leaf, int do_init, int off, int count, int kind, int root, int layer) {
    if ((threadIdx.x & 63) == 0) {
        const unsigned int q = atomicAdd(&map.counters[LK_CTR_FALLBACK], 1u);
        if (q < map.max_scan) {
            LkGroup d;
            d.leaf = leaf, d.parent = -1, d.oct = do_init, d.off = off, d.count = count, d.kind = kind, d.root = root, d.pad = layer;
            d.npts = d.new_points = d.block = d.layer = 0, d.state = 0, d.is_plane = 0, d.pad2[0] = d.pad2[1] = 0;
            reinterpret_cast<LkGroup*>(map.groups)[map.max_scan + q] = d;
        } else {
            atomicOr(&map.counters[LK_CTR_ERR], LK_E_SCRATCH_FULL);
        }
    }
}

// ONE leaf group: the leaf `Tn` (or child `To` of `Tp`, created here) receives its g points in input order - the register
// simulation of voxel_map.cc:186-237.  Lane L holds node point L: existing points from the leaf's block, then the group's points in
// order, fetched through point_at(rr, valid, pt) (called by ALL lanes: it may shuffle).  Only the first gs points fit in the wave;
// that is always enough to reach the freeze of a leaf (npts <= max_points_num + 1 <= 64), after which the rest of the group is
// ignored anyway; in the other cases the remainder goes to the per-point state machine (lk_insert_fallback_kernel), for which
// store_idx(base) must first put the group's indices into map.gidx[base .. base + g) when `off` < 0 (they are not there yet).
// cow_src (overlay replay only, lk_overlay_kernels.h): the leaf is a THIN private root - its record is private, its li.npts old points
// still sit in the BASE map's block cow_src.  They are read from there and ALL node points (old + new) are stored to the private block;
// the return value says whether that happened (false: the leaf did not take the register path, the caller copies the old points).
// DEFER (overlay replay): the leaf's one full plane fit is not made here - lane 0 writes a job (leaf, block, the last event's count,
// decision and moment sums) and lk_ov_fit_lane_kernel, one lane per job, does the eigen-decomposition, plane_var and the commit.
template <bool DEFER = false, typename PointAt, typename StoreIdx>
__device__ __forceinline__ bool apply_leaf(const LkMap& map, const LkParams& pr, const int Tn, const int Tp, const int To, const int g,
                                           const int root, const LeafInfo& li, int off, PointAt point_at, StoreIdx store_idx,
                                           const lk_pt_rec* cow_src = nullptr, LkFitJob* job = nullptr, int4* job_hdr = nullptr) {
    const int lane = threadIdx.x & 63;
    bool cow_done = false;
#ifdef LK_DEBUG_INS
    unsigned long long t0_ = wall_clock64();
    const unsigned long long ta_ = t0_;
    unsigned long long ph_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int nev_ = 0;
    if (lane == 0) atomicAdd(&lk_ins_dbg[15], 1ull);
#endif
    int leaf = Tn;
    NodeRegs r;
    r.npts = li.npts, r.new_points = li.new_points, r.block = li.block, r.layer = li.layer, r.state = li.state;
    bool lplane = li.is_plane != 0;
    if (leaf < 0) {  // voxel_map.cc:214-222: first point of a new octant creates the child
        const lk_node_rec* pn = &map.nodes[Tp];
        double pc[3] = {pn->voxel_center[0], pn->voxel_center[1], pn->voxel_center[2]};
        const int pl = bcast0(pn->layer);
        leaf = create_child(map, Tp, To, pc, pn->quater_length, pl);
#if LK_X_CHILD_CONST
        r.npts = 0, r.new_points = 0, r.block = -1, r.layer = pl + 1, r.state = LK_NODE_UPDATE_ENABLE;
#else
        r = node_load(&map.nodes[leaf]);
#endif
        lplane = false;
    }
    lk_node_rec* ln = &map.nodes[leaf];
#ifdef LK_DEBUG_LI
    if (Tn >= 0) {   // DEBUG BUILD ONLY: the leaf info handed over against the record in memory
        NodeRegs q = node_load(ln);
        const int qpl = (bcast0((int)map.planes[leaf].flags) & (int)LK_PLANE_IS_PLANE) != 0;
        if (lane == 0 && (q.npts != r.npts || q.new_points != r.new_points || q.block != r.block || q.layer != r.layer || q.state != r.state || qpl != (int)lplane)) {
            const unsigned long long k = atomicAdd(&lk_li_dbg[0], 1ull);
            if (k < 4) {
                unsigned long long* o = &lk_li_dbg[1 + 14 * k];
                o[0] = leaf, o[1] = root, o[2] = (unsigned)r.npts, o[3] = (unsigned)q.npts, o[4] = (unsigned)r.new_points, o[5] = (unsigned)q.new_points, o[6] = (unsigned)r.block, o[7] = (unsigned)q.block;
                o[8] = r.layer, o[9] = q.layer, o[10] = r.state, o[11] = q.state, o[12] = lplane, o[13] = off;
            }
        }
    }
#endif
    const int L = r.layer;
    const bool uninit = !(r.state & LK_NODE_INIT_OCTO);
    const bool live = (r.state & LK_NODE_UPDATE_ENABLE) != 0;
    const bool maxnp = !uninit && !lplane && L >= pr.max_layer;
    INS_STAMP(0);
    if (!uninit && !live && (lplane || maxnp)) return false;  // frozen leaf ignores its points
    int consumed = 0;
    bool need_init = false;
    if ((uninit || ((lplane || maxnp) && live)) && !(r.state & LK_NODE_PTS_DROPPED) && r.npts < LK_WAVE) {
        const int n0 = r.npts;
        const int gs = min(g, LK_WAVE - n0);
        double ppw[3] = {0, 0, 0}, pvar[6] = {0, 0, 0, 0, 0, 0};
        if (lane < n0) load_pt(cow_src ? cow_src : map.blocks[r.block].pts, nullptr, lane, ppw, pvar);
#ifdef LK_DEBUG_INS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        INS_STAMP(8);
#endif
        {
            const int rr = lane - n0;
            const bool valid = rr >= 0 && rr < gs;
            PtU pt;
            point_at(valid ? rr : 0, valid, pt);
            if (valid) {
#pragma unroll
                for (int c = 0; c < 3; ++c) ppw[c] = pt.pw[c];
#pragma unroll
                for (int c = 0; c < 6; ++c) pvar[c] = pt.var[c];
            }
        }
        INS_STAMP(1);
        const int thr = pr.layer_init_num[L];
        int cur = n0, newp = r.new_points;
        int mode = uninit ? 0 : (lplane ? 1 : 2);  // 0 un-initialised, 1 plane, 2 non-planar max-layer leaf
        bool frozen = false, general_init = false, stop = false, fitted = false, flipped_to_tree = false;
        PlaneFit fit;
        fit.is_plane = lplane;
        int fit_count = 0;
        while (consumed < gs && !stop) {
            // one step = the points up to the next refit event of the leaf's current mode (or all that are left); ONE copy of the
            // plane test in the instruction stream for the three modes (the root kernel's code is about as large as the instruction cache)
            const int rem = gs - consumed;
            const int m0 = mode;
            // mode 0: voxel_map.cc:186-189 then init_octo_tree :119-137;  mode 1: :191-204;  mode 2: :224-237
            const int lim = m0 == 0 ? thr + 1 - cur : min(6 - newp, (m0 == 1 ? pr.max_points_num : pr.max_points_num + 1) - cur);
            const int k = max(min(rem, lim), 1);
            cur += k, newp += k, consumed += k;
            if (m0 == 0 ? cur > thr : newp > 5) {
#if LK_X_ATTR & 8
                fit.is_plane = true;
#pragma unroll
                for (int q = 0; q < 9; ++q) fit.s9[q] = ppw[q % 3] + (double)cur;
#else
                fit = plane_test_regs<true>(ppw, lane < cur, cur, pr.planer_threshold);
#endif
                fit_count = cur, fitted = true, newp = 0;
                if (m0 == 0) {
                    if (fit.is_plane) {
                        mode = 1;
                        if (cur > pr.max_points_num) frozen = true, stop = true;
                    } else if (L >= pr.max_layer) {
                        mode = 2;  // cut_octo_tree returns at once at max_layer (:140-143)
                    } else {
                        general_init = true, stop = true;  // the generic code cuts the voxel
                    }
                } else if (m0 == 1) {
                    if (!fit.is_plane) {
                        if (L < pr.max_layer) flipped_to_tree = true, stop = true;
                        else mode = 2;
                    }
                } else if (fit.is_plane) {
                    mode = 1;
                }
            }
            if (m0 == 1 && cur >= pr.max_points_num) frozen = true, stop = true;
            if (m0 == 2 && cur > pr.max_points_num) frozen = true, stop = true;
        }
        INS_STAMP(2);
        // ---- commit points, counters, one full fit
#ifdef LK_DEBUG_INS
        if (Tn < 0) nev_ |= 1;
        if (cur > n0 && r.block < 0) nev_ |= 2;
        if (frozen) nev_ |= 4;
        if (fitted && fit.is_plane) nev_ |= 8;
#endif
        if (cur > n0 && r.block < 0) r.block = alloc_block(map);
        cow_done = cow_src != nullptr;
        if ((cow_src ? true : lane >= n0) && lane < cur) {
            lk_pt_rec* dst = &map.blocks[r.block].pts[lane];
#pragma unroll
            for (int c = 0; c < 3; ++c) dst->pw[c] = ppw[c];
#pragma unroll
            for (int c = 0; c < 6; ++c) dst->var[c] = pvar[c];
        }
        r.npts = cur;
        if (general_init) {
            r.new_points = cur;  // as counted by the pushes; init_octo_tree resets it
            node_store(ln, r, false);
            need_init = true;    // the generic code cuts the voxel: lk_insert_fallback_kernel
        } else {
            r.new_points = newp;
            if (fitted) {
                if (DEFER) {
                    if (lane == 0) {
                        job->base_block = -1, job->n_base = 0;
#pragma unroll
                        for (int q = 0; q < 9; ++q) job->s9[q] = fit.s9[q];
                        *job_hdr = make_int4(leaf, r.block, fit_count, fit.is_plane ? 1 : 0);   // {leaf, block, cnt, decided}: the dense header array (LkFitJob)
                    }
                } else {
                // the one full fit of this leaf in this bucket: the state of its LAST refit event
                const bool decided = fit.is_plane;
                const PlaneFit last = fit;   // the last event tested exactly these fit_count points: its sums are reused
                INS_STAMP(3);
#if LK_X_ATTR & 2
                fit = plane_test_regs<true, true>(ppw, lane < fit_count, fit_count, pr.planer_threshold, &last);
                fit.vmin[2] = fit.vmid[1] = fit.vmax[0] = 1.0, fit.emin = 1e-3, fit.emid = 1.0, fit.emax = 2.0;
#else
                fit = plane_test_regs<false, true>(ppw, lane < fit_count, fit_count, pr.planer_threshold, &last);
#endif
                INS_STAMP(4);
                fit.is_plane = decided;  // control flow above already followed the event's decision
                double acc21[21];
#if LK_X_ATTR & 1
#pragma unroll
                for (int q = 0; q < 21; ++q) acc21[q] = 1e-6;
#else
                if (fit.is_plane) plane_var_regs(fit, ppw, pvar, lane < fit_count, fit_count, acc21);
#endif
                INS_STAMP(5);
                plane_commit(&map.planes[leaf], &map.match[leaf], fit, acc21, fit_count);
                INS_STAMP(6);
#ifdef LK_DEBUG_INS
                if (lane == 0) atomicAdd(&lk_ins_dbg[14], 1ull);
#endif
                }
                r.state = (r.state | LK_NODE_INIT_OCTO) & ~LK_NODE_OCTO_STATE;
                if (flipped_to_tree) node_drop_block(map, r);  // its own points are never read again
            }
            if (frozen) node_freeze(map, r);
            node_store(ln, r, false);
            if (frozen && !flipped_to_tree) consumed = g;  // a frozen leaf ignores the rest of its points
        }
    }
#ifdef LK_DEBUG_LI
    if (lane == 0 && map.nodes[root].key[0] == 11 && map.nodes[root].key[1] == 10 && map.nodes[root].key[2] == 5) {
        const unsigned long long k = atomicAdd(&lk_li_dbg[60], 1ull);
        if (k < 4) {
            unsigned long long* o = &lk_li_dbg[1 + 14 * k];
            o[0] = leaf, o[1] = root, o[2] = (unsigned)Tn, o[3] = (unsigned)Tp, o[4] = (unsigned)To, o[5] = (unsigned)g, o[6] = (unsigned)r.npts, o[7] = (unsigned)r.block;
            o[8] = r.layer, o[9] = r.new_points, o[10] = r.state, o[11] = consumed, o[12] = need_init, o[13] = off;
        }
    }
#endif
    INS_STAMP(7);
#ifdef LK_DEBUG_INS
    if (lane == 0) {
        unsigned long long d_ = (wall_clock64() - ta_) / 200ull;
        atomicAdd(&lk_ev_hist[(nev_ > 15 ? 15 : nev_) * 32 + (int)(d_ > 31ull ? 31ull : d_)], 1u);
        if (nev_ == 8) {
            unsigned long long* o = &lk_slow_dbg[d_ >= 10ull ? 0 : 16];
            for (int k = 0; k < 10; ++k) atomicAdd(&o[k], ph_[k]);
            atomicAdd(&o[15], 1ull);
        }
    }
#endif
    // ---------------- a cut and / or whatever is left of the group: the generic state machine, in its own kernel
    if (need_init || consumed < g) {
        if (off < 0) {
            int base = 0;
            if (lane == 0) base = (int)atomicAdd(&map.counters[LK_CTR_GIDX], (unsigned int)g);
            base = bcast0(base);
            if (base + g > (int)map.max_scan) {
                if (lane == 0) atomicOr(&map.counters[LK_CTR_ERR], LK_E_SCRATCH_FULL);
                return cow_done;
            }
            store_idx(base);
            off = base;
        }
        insert_defer(map, leaf, need_init ? 1 : 0, off + consumed, g - consumed, 2, root, L);
    }
    return cow_done;
}

// Does a touched root only need its m queued points appended (no plane of its subtree can change in this bucket)?  An un-initialised
// root that stays at <= layer_init_num points, or a plane root that reaches neither its 6th new point (refit, voxel_map.cc:195) nor
// max_points_num (freeze, :199).
__device__ __forceinline__ bool root_is_light(const LkParams& pr, int m, unsigned int st, unsigned int pf, int npts, int newp) {
    if (m > 8) return false;
    if (!(st & LK_NODE_INIT_OCTO)) return (npts + m <= pr.layer_init_num[0]) && (npts + m <= LK_BLOCK_PTS);
    if ((pf & LK_PLANE_IS_PLANE) && (st & LK_NODE_UPDATE_ENABLE)) return (newp + m <= 5) && (npts + m < pr.max_points_num);
    return false;
}
// One wave per touched root (see the comment block above).
// (Round 3, measured and not kept - profiles/r03i_insert_chain_experiments.txt: waves taking roots from a queue counter instead of
// striding cost 0.60 against 0.44 ms per 5 x 20k scan - thousands of same-address atomics at launch; larger grids leave the kernel
// at 40 us: its duration is its slowest single root, 8-12 us typically with a tail to 30 us in the memory phases, see the histograms.)
// OV (overlay replay, lk_overlay_kernels.h): `map` is a scan's private map and some touched roots are THIN - childless voxels whose
// records the copy-on-write pass has made private while their old points still sit in the base map's block (pad_[LK_PAD_LIVE] == 2,
// pad_[LK_PAD_COWBLK] = 1 + that block).  This pass reads those points where they are and stores old + new points to the private
// block in one go (light roots and roots that are one in-place leaf group: nearly all); any other path copies them first.
#define LK_PAD_LIVE 3     // lk_node_rec::pad_[3] of a PRIVATE root record: 0 not in the slot's map yet, 1 complete, 2 thin
#define LK_PAD_COWBLK 5   // lk_node_rec::pad_[5] of a thin / split private root: 1 + id of the BASE map's point block that holds its old points
#define LK_PAD_SUMSRC 7   // lk_node_rec::pad_[7] of a private root: where its moment sums (LkLeafSum) are - 0 its own record, 1 the base leaf's, 2 none yet
#define LK_PAD_SPLIT 6    // lk_node_rec::pad_[6] of a private root leaf the FAST root pass (lk_ov_root_lane_kernel) has appended to without copying its old
                          // points: the first pad_[6] points of the leaf still are the base block's (COWBLK - 1), points pad_[6] .. npts-1 sit at their own
                          // index in the private block.  0 = the private block is complete.  The generic passes merge such a root before they touch it
// CPLX (overlay replay): the work list is not the touched list but what the fast root pass (lk_ov_root_lane_kernel) has left over -
// map.heavy = {root, index in the touched list} pairs, counter LK_CTR_HEAVY; the fit jobs stay indexed by the touched-list position.
template <bool FROM_PV, bool OV = false, bool CPLX = false>
__device__ __forceinline__ void dev_insert_root(const LkMap& map, const LkParams& pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                                                const lk_pt_rec* __restrict__ pv, int n, const int wave, const int nwaves, const LkMap* cow_base = nullptr,
                                                LkFitJob* jobs = nullptr, const size_t job_stride = 0, unsigned int* dyn_next = nullptr, int4* jobhdr = nullptr) {
    // dyn_next (not OV): a wave's FIRST root is its own index in the touched list, every further one the next nobody has taken (a ticket
    // counter, zero at the start of the pass) instead of index + nwaves: a wave that drew a plane fit (12-16 us) does not also own the
    // root nwaves further on while its neighbours, done with an append after 2 us, idle at the barrier (grid-resident stream kernel)
    const int lane = threadIdx.x & 63;
    const int n_touched = CPLX ? (int)min(map.counters[LK_CTR_HEAVY], map.max_scan) : (int)map.counters[LK_CTR_TOUCHED];
    if (CPLX && n_touched == 0) return;
    auto item_root = [&](int i) { return CPLX ? map.heavy[2 * i] : map.touched[i]; };
    LkGroup* groups = reinterpret_cast<LkGroup*>(map.groups);
    BucketConst bc;
    if (!FROM_PV) load_bucket_const(&filters[0], pr, bc);
#ifdef LK_DEBUG_INS
    const unsigned long long tw_ = wall_clock64();
    bool any_ = false;
#endif
    // OV: a wave of the batch replay works through ~100 roots one after the other, each a chain of dependent round trips (its id from the
    // touched list; its record, plane flags and slot line; its scan points; its leaf's points).  The first two are requested ahead: the id
    // TWO roots ahead, the record ONE root ahead (lane k holds 16-B piece k of the 128-B record, fields are picked with readlane) - they
    // land while the current root is worked on.  Roots of one touched list are distinct, so nothing requested ahead is changed meanwhile.
    // (OV: a slot-line entry is the queued POINT itself, {x, y, z, index} - lk_ov_reproject_kernel - so the root's scan points arrive
    // with its record instead of being gathered from the scan in 64-B sectors: 690 B per root for 173 B used)
    int pf_root1 = -1, pf_root2 = -1, pf_flags = 0, pf_slot = 0x7fffffff;
    float pf_px = 0.f, pf_py = 0.f, pf_pz = 0.f;
    int4 pf_rec = make_int4(0, 0, 0, 0);
    auto prefetch_record = [&](int r) {
        if (r >= 0) {
            if (lane < 8) pf_rec = reinterpret_cast<const int4*>(&map.nodes[r])[lane];
            pf_flags = (int)map.planes[r].flags;
            pf_slot = 0x7fffffff;
            if (lane < LK_SLOTS) {
                const float4 q = reinterpret_cast<const float4*>(map.slots)[(size_t)r * LK_SLOTS + lane];
                pf_px = q.x, pf_py = q.y, pf_pz = q.z, pf_slot = __float_as_int(q.w);
            }
        }
    };
    if (OV) {
        pf_root1 = wave < n_touched ? bcast0(item_root(wave)) : -1;
        pf_root2 = wave + nwaves < n_touched ? bcast0(item_root(wave + nwaves)) : -1;
        prefetch_record(pf_root1);
    }
    auto next_item = [&](int t) -> int {
        if (!OV && dyn_next) {
            unsigned int k = 0;
            if (lane == 0) k = atomicAdd(dyn_next, 1u);
            return nwaves + bcast0((int)k);
        }
        return t + nwaves;
    };
    for (int t = wave; t < n_touched; t = next_item(t)) {
#ifdef LK_DEBUG_INS
        const unsigned long long tr_ = wall_clock64();
        if (any_) ROOT_HIST(4, tw_);   // a wave with a second root: its time so far
        any_ = true;
#endif
        int root, m, rnpts, rnewp, rblock, rlayer, cur_list, slot_idx, ov_live = 0, ov_cowblk = 0, ov_split = 0;
        unsigned int rst, rpf;
        const float cpx = pf_px, cpy = pf_py, cpz = pf_pz;   // OV: this root's queued points (lane k: the k-th queued), before the next root's are requested
        if (OV) {
            root = pf_root1;
            rlayer = __builtin_amdgcn_readlane(pf_rec.w, 3);
            rnpts = __builtin_amdgcn_readlane(pf_rec.x, 4), rnewp = __builtin_amdgcn_readlane(pf_rec.y, 4);
            rst = (unsigned int)__builtin_amdgcn_readlane(pf_rec.z, 4), rblock = __builtin_amdgcn_readlane(pf_rec.w, 4);
            cur_list = __builtin_amdgcn_readlane(pf_rec.w, 5);
            m = __builtin_amdgcn_readlane(pf_rec.x, 6), ov_live = __builtin_amdgcn_readlane(pf_rec.w, 6);   // pad_[0], pad_[LK_PAD_LIVE]
            ov_cowblk = __builtin_amdgcn_readlane(pf_rec.y, 7);                                              // pad_[LK_PAD_COWBLK]
            ov_split = __builtin_amdgcn_readlane(pf_rec.z, 7);                                               // pad_[LK_PAD_SPLIT]
            rpf = (unsigned int)bcast0(pf_flags);
            slot_idx = pf_slot;
            pf_root1 = pf_root2;
            prefetch_record(pf_root1);
            pf_root2 = t + 2 * nwaves < n_touched ? bcast0(item_root(t + 2 * nwaves)) : -1;
        } else {
            root = bcast0(map.touched[t]);
        }
        const int tix = CPLX ? bcast0(map.heavy[2 * t + 1]) : t;   // the root's position in the touched list = its row of fit jobs
        lk_node_rec* nd = &map.nodes[root];
        if (!OV) {
            // one batch of loads: the root's record, its plane flags, its slot line
            m = bcast0((int)nd->pad_[0]);
            rst = (unsigned int)bcast0((int)nd->state), rpf = (unsigned int)bcast0((int)map.planes[root].flags);
            rnpts = bcast0(nd->npts), rnewp = bcast0(nd->new_points), rblock = bcast0(nd->block), rlayer = bcast0(nd->layer);
            cur_list = bcast0(nd->list_head);
            slot_idx = (lane < LK_SLOTS) ? map.slots[(size_t)root * LK_SLOTS + lane] : 0x7fffffff;
        }
        const lk_pt_rec* cow_src = nullptr;
        // this root's fit jobs (apply_leaf<DEFER>): none yet.  Entry [g][t] = its g-th inline leaf group: nearly every root is ONE group, so
        // plane 0 is dense for lk_ov_fit_lane_kernel's lanes (job_stride = entries per plane)
        if (OV && lane < LK_INLINE_GROUPS) jobhdr[(size_t)lane * job_stride + tix].z = 0;
        int job_i = 0;
        int cow_n = 0;   // old points that still sit in the base block
        if (OV) {
            const unsigned int live = (unsigned int)ov_live;
            const int cow_blk = ov_cowblk - 1;
            if (live == 2u && cow_blk >= 0) cow_src = cow_base->blocks[cow_blk].pts, cow_n = rnpts;
            else if (ov_split > 0 && cow_blk >= 0) cow_src = cow_base->blocks[cow_blk].pts, cow_n = min(ov_split, rnpts);
            if (live == 2u && lane == 0) nd->pad_[LK_PAD_LIVE] = 1;   // complete when this wave is done with it (nothing reads the word before the next bucket)
        }
        // thin / split root leaving the fused paths: its old points go to the private block first, then it is a root like any other
        auto cow_finalise = [&]() {
            if (OV && cow_src) {
                if (lane < cow_n && rblock >= 0) {
                    double qw[3], qv[6];
                    load_pt(cow_src, nullptr, lane, qw, qv);
                    lk_pt_rec* dst = &map.blocks[rblock].pts[lane];
#pragma unroll
                    for (int c = 0; c < 3; ++c) dst->pw[c] = qw[c];
#pragma unroll
                    for (int c = 0; c < 6; ++c) dst->var[c] = qv[c];
                }
                if (lane == 0) nd->pad_[LK_PAD_SPLIT] = 0;
                wave_fence();
                cow_src = nullptr;
            }
        };
        if (OV && ov_split > 0 && ov_live != 2) cow_finalise();   // a leaf the fast pass left split: merged before anything else looks at its block
        if (lane == 0) nd->pad_[0] = 0, nd->list_head = -1;   // the root's bucket-local queue is consumed
        // ---- light root: append only (one lane per point, input order = ascending index)
        const bool light = !FROM_PV && root_is_light(pr, m, rst, rpf, rnpts, rnewp);
        if (light) {
            const int idx = (lane < m) ? slot_idx : 0x7fffffff;
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (j < m && __builtin_amdgcn_readlane(idx, j) < idx) ? 1 : 0;
            int block = rblock;
            if (block < 0) block = alloc_block(map);
            cow_finalise();   // (a thin root owns its private block already)
            if (lane < m) {
                float4 p;
                if (OV) p = make_float4(cpx, cpy, cpz, 0.f);
                else p = reinterpret_cast<const float4*>(pts)[idx];
                const PointGeom gm = point_geom(p.x, p.y, p.z, bc, pr);
                lk_pt_rec* dst = &map.blocks[block].pts[rnpts + rank];
                dst->pw[0] = gm.p_w.x, dst->pw[1] = gm.p_w.y, dst->pw[2] = gm.p_w.z;
                dst->var[0] = gm.var.xx, dst->var[1] = gm.var.xy, dst->var[2] = gm.var.xz;
                dst->var[3] = gm.var.yy, dst->var[4] = gm.var.yz, dst->var[5] = gm.var.zz;
            }
            if (lane == 0) nd->npts = rnpts + m, nd->new_points = rnewp + m, nd->block = block;
            ROOT_HIST(0, tr_);
            continue;
        }
        if (lane == 0) map.dirty[root] = map.epoch;   // pipelined stream path: a plane of this root's subtree may change in this bucket (LkMap::dirty)
        // ---- the root's indices
        int base = 0;
        const bool in_slots = m <= LK_SLOTS;  // the common case: every queued index sits in the root's slot line
        if (!in_slots) {
            if (lane == 0) base = (int)atomicAdd(&map.counters[LK_CTR_SCRATCH], (unsigned int)m);
            base = bcast0(base);
            if (base + m > (int)map.max_scan) {
                if (lane == 0) atomicOr(&map.counters[LK_CTR_ERR], LK_E_SCRATCH_FULL);
                continue;
            }
            // unordered indices -> scratch: the slot line with one coalesced store, the overflow by walking the list
            if (lane < LK_SLOTS) map.scratch[base + lane] = slot_idx;
            for (int k = LK_SLOTS; k < m && cur_list >= 0; ++k) {
                if (lane == 0) map.scratch[base + k] = cur_list;
                cur_list = bcast0(map.next[cur_list]);
            }
            wave_fence();
        }
        if (m > LK_WAVE) {  // very long list: handed over whole
            cow_finalise();
            insert_defer(map, root, 0, base, m, 1, root, 0);
            ROOT_HIST(3, tr_);
            continue;
        }
        // sort the root's indices in registers: rank = number of smaller indices, then a forward permute
        const int myidx = (lane < m) ? (in_slots ? slot_idx : map.scratch[base + lane]) : 0x7fffffff;
        int rank = 0;
#if LK_X_ATTR & 16
        rank = lane;
#else
        for (int j = 0; j < m; ++j) rank += (__builtin_amdgcn_readlane(myidx, j) < myidx) ? 1 : 0;
#endif
        const int sidx = __builtin_amdgcn_ds_permute(((lane < m) ? rank : lane) << 2, myidx);  // lane j: j-th smallest
        const bool mine = lane < m;
        // every lane walks (read-only) to the node its point would be pushed into: down through initialised non-planar nodes
        // below max_layer (voxel_map.cc:205-223); these never change again, so the walk is exact for the whole bucket.
        // tnode < 0: child `toct` of `tparent` does not exist.  The target's counters come along (its record is being read anyway).
        int tnode = -1, tparent = -1, toct = 0;
        int t_npts = 0, t_newp = 0, t_block = -1, t_layer = 0, t_plane = 0;
        unsigned int t_state = 0;
        float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool pts_in_hand = OV && in_slots;
        if (pts_in_hand) {   // the points came in queue order: the same forward permute as the indices
            const int dstl = ((lane < m) ? rank : lane) << 2;
            p4.x = __int_as_float(__builtin_amdgcn_ds_permute(dstl, __float_as_int(cpx)));
            p4.y = __int_as_float(__builtin_amdgcn_ds_permute(dstl, __float_as_int(cpy)));
            p4.z = __int_as_float(__builtin_amdgcn_ds_permute(dstl, __float_as_int(cpz)));
        }
        if (mine) {
            double pw[3];
            if (pts_in_hand) insert_point_pw_of(pr, bc, p4, pw);
            else insert_point_pw<FROM_PV>(pr, bc, pts, pv, sidx, pw, p4);
            int node = root;
            unsigned int st = rst, pf = rpf;
            int npts = rnpts, newp = rnewp, block = rblock, layer = rlayer;
            for (int depth = 0; depth <= LK_MAX_LAYER; ++depth) {
                const bool pl = (pf & LK_PLANE_IS_PLANE) != 0;
                if (!(st & LK_NODE_INIT_OCTO) || pl || layer >= pr.max_layer) {
                    tnode = node;
                    t_npts = npts, t_newp = newp, t_block = block, t_layer = layer, t_state = st, t_plane = pl ? 1 : 0;
                    break;
                }
                const lk_node_rec* nr = &map.nodes[node];
                const int oct = octant_of(pw, nr->voxel_center);
                const int child = nr->child[oct];
                if (child < 0) {
                    tnode = -1, tparent = node, toct = oct;
                    break;
                }
                node = child;
                const lk_node_rec* cr = &map.nodes[child];
                st = cr->state, npts = cr->npts, newp = cr->new_points, block = cr->block, layer = cr->layer;
                pf = map.planes[child].flags;
                if (depth == LK_MAX_LAYER) {   // cannot happen (layer <= max_layer <= LK_MAX_LAYER stops the walk above); kept total
                    tnode = node;
                    t_npts = npts, t_newp = newp, t_block = block, t_layer = layer, t_state = st, t_plane = (pf & LK_PLANE_IS_PLANE) ? 1 : 0;
                }
            }
        }
        // one group per distinct target
        int ngroups = 0;
        unsigned long long first_grp = 0ull;
        {
            unsigned long long todo = __ballot(mine);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int Tn = __builtin_amdgcn_readlane(tnode, leader), Tp = __builtin_amdgcn_readlane(tparent, leader),
                          To = __builtin_amdgcn_readlane(toct, leader);
                const unsigned long long grp = __ballot(mine && tnode == Tn && tparent == Tp && toct == To) & todo;
                if (ngroups == 0) first_grp = grp;
                todo &= ~grp;
                ++ngroups;
            }
        }
#ifndef LK_X_NOINLINE
#define LK_X_NOINLINE 0   // debug / A-B only: 1 = every root hands its groups to lk_insert_apply_kernel
#endif
        if (OV && cow_src) {   // fused only when the root itself is the one leaf all its points go to
            const int l0 = __ffsll((long long)__ballot(mine)) - 1;
            if (ngroups != 1 || LK_X_NOINLINE || __builtin_amdgcn_readlane(tnode, l0 < 0 ? 0 : l0) != root) cow_finalise();
        }
        ROOT_HIST(5, tr_);
        if (ngroups <= LK_INLINE_GROUPS && !LK_X_NOINLINE) {
            // ---- one leaf group (98 % of the roots) or a few: applied here, from registers, one after the other.  A group's points are
            // compacted to lanes 0 .. g-1 in lane (= input) order - the identity when the root is one group.  The launch for the emitted
            // groups then finds nothing to do in the steady state (a handful of groups cost it ~10 us per bucket otherwise).
            unsigned long long todo = __ballot(mine);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int Tn = __builtin_amdgcn_readlane(tnode, leader), Tp = __builtin_amdgcn_readlane(tparent, leader),
                          To = __builtin_amdgcn_readlane(toct, leader);
                const unsigned long long grp = __ballot(mine && tnode == Tn && tparent == Tp && toct == To) & todo;
                todo &= ~grp;
                const int g = __popcll(grp);
                const bool in = ((grp >> lane) & 1ull) != 0;
                const int dst = (in ? __popcll(grp & ((1ull << lane) - 1ull)) : 63) << 2;   // g == 64: every lane is a member
                const float cx = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(p4.x)));
                const float cy = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(p4.y)));
                const float cz = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(p4.z)));
                const int cidx = __builtin_amdgcn_ds_permute(dst, sidx);
                LeafInfo li;
                li.npts = __builtin_amdgcn_readlane(t_npts, leader), li.new_points = __builtin_amdgcn_readlane(t_newp, leader);
                li.block = __builtin_amdgcn_readlane(t_block, leader), li.layer = __builtin_amdgcn_readlane(t_layer, leader);
                li.state = (unsigned int)__builtin_amdgcn_readlane((int)t_state, leader), li.is_plane = __builtin_amdgcn_readlane(t_plane, leader);
                auto point_at = [&](int rr, bool valid, PtU& pt) {
                    if (FROM_PV) {
                        const int idx = __shfl(cidx, rr, LK_WAVE);
                        if (valid) load_pt(pv, nullptr, idx, pt.pw, pt.var);
                    } else {
                        const float qx = __shfl(cx, rr, LK_WAVE), qy = __shfl(cy, rr, LK_WAVE), qz = __shfl(cz, rr, LK_WAVE);
#if LK_X_ATTR & 4
                        if (valid) {
                            pt.pw[0] = qx, pt.pw[1] = qy, pt.pw[2] = qz;
                            pt.var[0] = pt.var[3] = pt.var[5] = 1e-4, pt.var[1] = pt.var[2] = pt.var[4] = 0.0;
                        }
#else
                        if (valid) geom_to_pt(point_geom(qx, qy, qz, bc, pr), pt);
#endif
                    }
                };
                auto store_idx = [&](int gb) {
                    if (lane < g) map.gidx[gb + lane] = cidx;
                };
                const bool cow_done = apply_leaf<OV>(map, pr, Tn, Tp, To, g, root, li, -1, point_at, store_idx, OV ? cow_src : nullptr,
                                                     OV ? &jobs[(size_t)job_i * job_stride + tix] : nullptr, OV ? &jobhdr[(size_t)job_i * job_stride + tix] : nullptr);
                ++job_i;
                if (OV && cow_src) {
                    if (cow_done) cow_src = nullptr;
                    else cow_finalise();
                }
            }
            ROOT_HIST(ngroups == 1 ? 1 : 2, tr_);
            continue;
        }
        // ---- several groups: one descriptor per group, the group's indices in lane (= input) order
        int gbase = 0, ibase = 0;
        if (lane == 0) {
            gbase = (int)atomicAdd(&map.counters[LK_CTR_GROUPS], (unsigned int)ngroups);
            ibase = (int)atomicAdd(&map.counters[LK_CTR_GIDX], (unsigned int)m);
        }
        gbase = bcast0(gbase), ibase = bcast0(ibase);
        if (gbase + ngroups > (int)map.max_scan || ibase + m > (int)map.max_scan) {
            if (lane == 0) atomicOr(&map.counters[LK_CTR_ERR], LK_E_SCRATCH_FULL);
            continue;
        }
        {
            unsigned long long todo = __ballot(mine);
            int gi = 0, off = ibase;
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int Tn = __builtin_amdgcn_readlane(tnode, leader), Tp = __builtin_amdgcn_readlane(tparent, leader),
                          To = __builtin_amdgcn_readlane(toct, leader);
                const unsigned long long grp = __ballot(mine && tnode == Tn && tparent == Tp && toct == To) & todo;
                todo &= ~grp;
                const int g = __popcll(grp);
                if ((grp >> lane) & 1ull) map.gidx[off + __popcll(grp & ((1ull << lane) - 1ull))] = sidx;
                LkGroup d;
                d.leaf = Tn, d.parent = Tp, d.oct = To, d.off = off, d.count = g, d.kind = 0, d.root = root, d.pad = 0;
                d.npts = __builtin_amdgcn_readlane(t_npts, leader), d.new_points = __builtin_amdgcn_readlane(t_newp, leader);
                d.block = __builtin_amdgcn_readlane(t_block, leader), d.layer = __builtin_amdgcn_readlane(t_layer, leader);
                d.state = (unsigned int)__builtin_amdgcn_readlane((int)t_state, leader), d.is_plane = __builtin_amdgcn_readlane(t_plane, leader);
                d.pad2[0] = d.pad2[1] = 0;
                if (lane == 0) groups[gbase + gi] = d;
                off += g;
                ++gi;
            }
        }
        ROOT_HIST(2, tr_);
    }
}

// One wave per emitted group (roots with several leaf groups).
template <bool FROM_PV>
__device__ __forceinline__ void dev_insert_apply(const LkMap& map, const LkParams& pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                           const lk_pt_rec* __restrict__ pv, int n, const int wave, const int nwaves) {
    const int lane = threadIdx.x & 63;
    const int n_groups = (int)min(map.counters[LK_CTR_GROUPS], map.max_scan);
    if (n_groups == 0) return;
    const LkGroup* groups = reinterpret_cast<const LkGroup*>(map.groups);
    BucketConst bc;
    if (!FROM_PV) load_bucket_const(&filters[0], pr, bc);
    for (int t = wave; t < n_groups; t += nwaves) {
        const int4* dq = reinterpret_cast<const int4*>(&groups[t]);
        const int4 d0 = dq[0], d1 = dq[1], d2 = dq[2], d3 = dq[3];
        const int Tn = bcast0(d0.x), Tp = bcast0(d0.y), To = bcast0(d0.z), off = bcast0(d0.w), g = bcast0(d1.x), root = bcast0(d1.z);
        LeafInfo li;
        li.npts = bcast0(d2.x), li.new_points = bcast0(d2.y), li.block = bcast0(d2.z), li.layer = bcast0(d2.w);
        li.state = (unsigned int)bcast0(d3.x), li.is_plane = bcast0(d3.y);
        auto point_at = [&](int rr, bool valid, PtU& pt) {
            if (valid) {
                const int idx = map.gidx[off + rr];
                if (FROM_PV) {
                    load_pt(pv, nullptr, idx, pt.pw, pt.var);
                } else {
                    const float4 p = reinterpret_cast<const float4*>(pts)[idx];
                    geom_to_pt(point_geom(p.x, p.y, p.z, bc, pr), pt);
                }
            }
        };
        auto store_idx = [&](int) {};
        apply_leaf(map, pr, Tn, Tp, To, g, root, li, off, point_at, store_idx);
    }
}


// The generic code path of the insert for the few groups the register simulation hands over (a voxel that has to be
// cut: init_octo_tree / cut_octo_tree voxel_map.cc:119-183; leftover points after a flip to a tree; roots with more
// than 64 queued points): one wave per item, the per-point state machine dev_update_octo / dev_init_octo<L>.
template <bool FROM_PV>
__device__ __forceinline__ void dev_insert_fallback(const LkMap& map, const LkParams& pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                              const lk_pt_rec* __restrict__ pv, int n, const int wave, const int nwaves) {
    const int lane = threadIdx.x & 63;
    const int n_items = (int)min(map.counters[LK_CTR_FALLBACK], map.max_scan);
    if (n_items == 0) return;
    const LkGroup* items = reinterpret_cast<const LkGroup*>(map.groups) + map.max_scan;
    BucketConst bc;
    if (!FROM_PV) load_bucket_const(&filters[0], pr, bc);
    auto point_of = [&](int idx, PtU& pt) {
        if (FROM_PV) {
            load_pt(pv, nullptr, idx, pt.pw, pt.var);
        } else {
            const float4 p = reinterpret_cast<const float4*>(pts)[idx];
            PointGeom g = point_geom(p.x, p.y, p.z, bc, pr);
            pt.pw[0] = g.p_w.x, pt.pw[1] = g.p_w.y, pt.pw[2] = g.p_w.z;
            pt.var[0] = g.var.xx, pt.var[1] = g.var.xy, pt.var[2] = g.var.xz;
            pt.var[3] = g.var.yy, pt.var[4] = g.var.yz, pt.var[5] = g.var.zz;
        }
    };
    for (int t = wave; t < n_items; t += nwaves) {
        const int4 d0 = reinterpret_cast<const int4*>(&items[t])[0], d1 = reinterpret_cast<const int4*>(&items[t])[1];
        const int leaf = bcast0(d0.x), do_init = bcast0(d0.z), off = bcast0(d0.w), g = bcast0(d1.x), kind = bcast0(d1.y),
                  root = bcast0(d1.z), L = bcast0(d1.w);
        if (kind == 1) {
            // very long list: successive-minimum selection, one point at a time
            lk_node_rec* nd = &map.nodes[root];
            int last = -1;
            for (int step = 0; step < g; ++step) {
                int best = 0x7fffffff;
                for (int j = lane; j < g; j += LK_WAVE) {
                    int v = map.scratch[off + j];
                    if (v > last && v < best) best = v;
                }
                best = wave_min_i(best);
                last = best;
                PtU pt;
                point_of(best, pt);
                dev_update_octo(map, pr, root, pt);
                unsigned int st = (unsigned int)bcast0((int)nd->state);
                unsigned int pf = (unsigned int)bcast0((int)map.planes[root].flags);
                if ((st & LK_NODE_INIT_OCTO) && (pf & LK_PLANE_IS_PLANE) && !(st & LK_NODE_UPDATE_ENABLE)) break;
            }
            continue;
        }
        if (do_init) {
            switch (L) {
                case 0: dev_init_octo<0>(map, pr, leaf); break;
                case 1: dev_init_octo<1>(map, pr, leaf); break;
                case 2: dev_init_octo<2>(map, pr, leaf); break;
                default: dev_init_octo<3>(map, pr, leaf); break;
            }
        }
        for (int tpos = 0; tpos < g; ++tpos) {
            PtU pt;
            point_of(bcast0(map.gidx[off + tpos]), pt);
            dev_update_octo(map, pr, leaf, pt);
        }
    }
}

#ifndef LK_ROOT_WAVES
#define LK_ROOT_WAVES 2   // waves per SIMD the stream path's root pass is compiled for
#endif
template <bool FROM_PV>
__global__ void __launch_bounds__(LK_MB, LK_ROOT_WAVES)
    lk_insert_root_kernel(LkMap map, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                          const lk_pt_rec* __restrict__ pv, int n) {
    dev_insert_root<FROM_PV>(map, pr, filters, pts, pv, n, (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}
// Small buckets (the reference's 2 ms time bins hold tens of points): the whole ordered insert as ONE launch.  Every workgroup runs
// the root pass over its share of the touched roots; the workgroup that finishes LAST (a ticket) then applies whatever leaf groups
// the roots with several groups emitted and runs the generic fallback items - both usually none.  Two launches of ~5 us each saved
// per bucket on a dependent chain of ~25 us; not for large buckets, where hundreds of emitted groups want hundreds of workgroups.
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_MB)
    lk_insert_small_kernel(LkMap map, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts, int n);
#else
__global__ void __launch_bounds__(LK_MB)
    lk_insert_small_kernel(LkMap map, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts, int n) {
    __shared__ int is_last;
    dev_insert_root<false>(map, pr, filters, pts, (const lk_pt_rec*)nullptr, n, (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int t = atomicAdd(&map.spec[LK_SPEC_TICKET2], 1u);
        is_last = t == gridDim.x - 1;
        if (is_last) map.spec[LK_SPEC_TICKET2] = 0;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();   // acquire: the other workgroups' descriptors, items and tree updates
    dev_insert_apply<false>(map, pr, filters, pts, (const lk_pt_rec*)nullptr, n, (int)(threadIdx.x >> 6), LK_MB >> 6);
    __threadfence();
    __syncthreads();
    dev_insert_fallback<false>(map, pr, filters, pts, (const lk_pt_rec*)nullptr, n, (int)(threadIdx.x >> 6), LK_MB >> 6);
}
#endif
// (Round 5, measured and not kept: apply + fallback as ONE launch - all workgroups apply, the last one (ticket) runs the fallback items.
// A kernel that contains the fallback code needs 256 VGPRs + 6.3 KB of scratch per lane, and a 256-workgroup launch of THAT costs ~25 us
// even when every workgroup leaves after two counter reads: 0.475 against 0.375 ms per 5 x 20 000-point scan.  profiles/EXPERIMENTS.md.)
template <bool FROM_PV>
__global__ void __launch_bounds__(LK_MB)
    lk_insert_apply_kernel(LkMap map, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                           const lk_pt_rec* __restrict__ pv, int n) {
    dev_insert_apply<FROM_PV>(map, pr, filters, pts, pv, n, (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
}
template <bool FROM_PV>
__global__ void __launch_bounds__(LK_MB)
    lk_insert_fallback_kernel(LkMap map, LkParams pr, const LkFilter* filters, const lk_point* __restrict__ pts,
                           const lk_pt_rec* __restrict__ pv, int n) {
    dev_insert_fallback<FROM_PV>(map, pr, filters, pts, pv, n, (int)((blockIdx.x * LK_MB + threadIdx.x) >> 6), (int)((gridDim.x * LK_MB) >> 6));
    // last kernel of a bucket's insert: its last workgroup publishes the bucket's epoch (pipelined stream path: what a verify wave
    // with suspect points waits for, lk_verify_kernel)
    if (map.epoch != 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned int t = atomicAdd(&map.spec[LK_SPEC_TICKET], 1u);
            if (t == gridDim.x - 1) {
                map.spec[LK_SPEC_TICKET] = 0;
                __hip_atomic_store(&map.spec[LK_SPEC_DONE], map.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// hashing half of UpdateVoxelMap for caller-supplied pointWithVar records
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_queue_pv_kernel(LkMap map, LkParams pr, const lk_pt_rec* __restrict__ pv, int n);
#else
__global__ void __launch_bounds__(256) lk_queue_pv_kernel(LkMap map, LkParams pr, const lk_pt_rec* __restrict__ pv, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int key[3];
    key_floor(V3{pv[i].pw[0], pv[i].pw[1], pv[i].pw[2]}, pr.voxel_size_f, key);
    int root = root_find_or_create(map, pr, key);
    if (root < 0) return;
    queue_point_on_root(map, root, i);
}
#endif

// ------------------------------------------------------------------ first-frame build (voxel_map.cc:287-334)
// per point: world point (f32 -> f64), first-frame variance (:303-307), root voxel id
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256)
    lk_build_points_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const float* __restrict__ xyz_world,
                           const float* __restrict__ xyz_body, int n, lk_pt_rec* __restrict__ bpts,
                           unsigned int* __restrict__ root_of, int* __restrict__ idx);
#else
__global__ void __launch_bounds__(256)
    lk_build_points_kernel(LkMap map, LkParams pr, const LkFilter* __restrict__ filters, const float* __restrict__ xyz_world,
                           const float* __restrict__ xyz_body, int n, lk_pt_rec* __restrict__ bpts,
                           unsigned int* __restrict__ root_of, int* __restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    BucketConst bc;
    load_bucket_const(&filters[0], pr, bc);
    V3 pw = V3{(double)xyz_world[3 * i], (double)xyz_world[3 * i + 1], (double)xyz_world[3 * i + 2]};
    V3 pb = V3{(double)xyz_body[3 * i], (double)xyz_body[3 * i + 1], (double)xyz_body[3 * i + 2]};
    S3 body = calc_body_cov(pb, pr);
    if (pb.z == 0) pb.z = 0.0001;  // calcBodyCov mutates its argument before the crossmat is formed
    double K[9];
    skew3(pb, K);
#pragma unroll
    for (int q = 0; q < 9; ++q) K[q] = -K[q];
    S3 a = congruence(bc.RE, body);
    S3 b = congruence(K, bc.Prr);
    lk_pt_rec r;
    r.pw[0] = pw.x, r.pw[1] = pw.y, r.pw[2] = pw.z;
    r.var[0] = a.xx + b.xx + bc.Ppp.xx, r.var[1] = a.xy + b.xy + bc.Ppp.xy, r.var[2] = a.xz + b.xz + bc.Ppp.xz;
    r.var[3] = a.yy + b.yy + bc.Ppp.yy, r.var[4] = a.yz + b.yz + bc.Ppp.yz, r.var[5] = a.zz + b.zz + bc.Ppp.zz;
    bpts[i] = r;
    int key[3];
    key_floor(pw, pr.voxel_size_f, key);
    int root = root_find_or_create(map, pr, key);
    root_of[i] = (root < 0) ? 0xffffffffu : (unsigned int)root;
    idx[i] = i;
}
#endif

// after the stable sort by root id: segment bounds per root, touched-root list
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256)
    lk_build_segments_kernel(LkMap map, const unsigned int* __restrict__ keys, int n);
#else
__global__ void __launch_bounds__(256)
    lk_build_segments_kernel(LkMap map, const unsigned int* __restrict__ keys, int n) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    unsigned int k = keys[s];
    if (k == 0xffffffffu) return;
    if (s == 0 || keys[s - 1] != k) {
        map.nodes[k].pad_[1] = (unsigned int)s;
        unsigned int t = atomicAdd(&map.counters[LK_CTR_TOUCHED], 1u);
        map.touched[t] = (int)k;
    }
    if (s == n - 1 || keys[s + 1] != k) map.nodes[k].pad_[2] = (unsigned int)(s + 1);
}
#endif

// init_octo_tree / cut_octo_tree over an index segment (points in bpts, indices idx_in[begin..begin+count))
template <int L>
__device__ __noinline__ void dev_build_node(const LkMap m, const LkParams pr, int node, const lk_pt_rec* bpts, int* idx_in,
                                            int* idx_out, int begin, int count) {
    const int lane = threadIdx.x & 63;
    lk_node_rec* nd = &m.nodes[node];
    NodeRegs r = node_load(nd);
    r.npts = count;
    r.new_points = count;
    bool keep = true;
    if (count > pr.layer_init_num[L]) {
        const bool is_plane = dev_init_plane(&m.planes[node], &m.match[node], pr.planer_threshold, bpts, idx_in + begin, count);
        if (is_plane) {
            r.state &= ~LK_NODE_OCTO_STATE;
            if (count > pr.max_points_num) {
                node_freeze(m, r);
                keep = false;
            }
        } else {
            r.state |= LK_NODE_OCTO_STATE;
            if (L >= pr.max_layer) {
                r.state &= ~LK_NODE_OCTO_STATE;
            } else if constexpr (L < LK_MAX_LAYER) {
                keep = false;  // dead points
                double center[3] = {nd->voxel_center[0], nd->voxel_center[1], nd->voxel_center[2]};
                const float quater = nd->quater_length;
                int cnt[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) cnt[o] = 0;
                for (int j0 = 0; j0 < count; j0 += LK_WAVE) {
                    int j = j0 + lane;
                    int oct = -1;
                    if (j < count) {
                        const lk_pt_rec* p = &bpts[idx_in[begin + j]];
                        oct = octant_of(p->pw, center);
                    }
#pragma unroll
                    for (int o = 0; o < 8; ++o) cnt[o] += __popcll(__ballot(oct == o));
                }
                int off[8];
                int run = begin;
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    off[o] = run;
                    run += cnt[o];
                }
                int fill[8];
#pragma unroll
                for (int o = 0; o < 8; ++o) fill[o] = 0;
                const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                for (int j0 = 0; j0 < count; j0 += LK_WAVE) {
                    int j = j0 + lane;
                    int oct = -1, id = -1;
                    if (j < count) {
                        id = idx_in[begin + j];
                        oct = octant_of(bpts[id].pw, center);
                    }
#pragma unroll
                    for (int o = 0; o < 8; ++o) {
                        unsigned long long mk = __ballot(oct == o);
                        if (oct == o) idx_out[off[o] + fill[o] + __popcll(mk & lt)] = id;
                        fill[o] += __popcll(mk);
                    }
                }
                wave_fence();
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    if (cnt[o] > 0) {
                        int child = create_child(m, node, o, center, quater, L);
                        dev_build_node<L + 1>(m, pr, child, bpts, idx_out, idx_in, off[o], cnt[o]);
                    }
                }
            }
        }
        r.state |= LK_NODE_INIT_OCTO;
        r.new_points = 0;
    }
    if (keep && r.npts > 0) {
        if (r.npts <= LK_BLOCK_PTS) {
            r.block = alloc_block(m);
            for (int j = lane; j < r.npts; j += LK_WAVE) m.blocks[r.block].pts[j] = bpts[idx_in[begin + j]];
        } else {
            r.state |= LK_NODE_PTS_DROPPED;
        }
    }
    node_store(nd, r);
}

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_MB)
    lk_build_tree_kernel(LkMap map, LkParams pr, const lk_pt_rec* __restrict__ bpts, int* idxA, int* idxB);
#else
__global__ void __launch_bounds__(LK_MB)
    lk_build_tree_kernel(LkMap map, LkParams pr, const lk_pt_rec* __restrict__ bpts, int* idxA, int* idxB) {
    const int wave = (blockIdx.x * LK_MB + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * LK_MB) >> 6;
    const int n_touched = (int)map.counters[LK_CTR_TOUCHED];
    for (int t = wave; t < n_touched; t += nwaves) {
        const int root = bcast0(map.touched[t]);
        const int b = bcast0((int)map.nodes[root].pad_[1]), e = bcast0((int)map.nodes[root].pad_[2]);
        dev_build_node<0>(map, pr, root, bpts, idxA, idxB, b, e - b);
    }
}
#endif

// Start of a bucket's insert phase: clear the per-bucket counters and make the blocks retired during the previous
// bucket allocatable (see pop_or_bump_block).  One workgroup.
__device__ __forceinline__ void dev_bucket_begin(const LkMap& map) {
    __shared__ int base, nfreed;
    if (threadIdx.x == 0) {
        int fc = (int)map.counters[LK_CTR_FREE];
        base = fc > 0 ? fc : 0;
        unsigned int nf = map.counters[LK_CTR_FREED];
        nfreed = (int)(nf < map.max_blocks ? nf : map.max_blocks);
        if (base + nfreed > (int)map.max_blocks) nfreed = (int)map.max_blocks - base;
        map.counters[LK_CTR_TOUCHED] = 0;
        map.counters[LK_CTR_SCRATCH] = 0;
        map.counters[LK_CTR_HEAVY] = 0;
        map.counters[LK_CTR_GROUPS] = 0;
        map.counters[LK_CTR_GIDX] = 0;
        map.counters[LK_CTR_FALLBACK] = 0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nfreed; i += (int)blockDim.x) map.free_list[base + i] = map.freed_next[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        map.counters[LK_CTR_FREE] = (unsigned int)(base + nfreed);
        map.counters[LK_CTR_FREED] = 0;
    }
}
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_bucket_begin_kernel(LkMap map);
#else
__global__ void __launch_bounds__(256) lk_bucket_begin_kernel(LkMap map) { dev_bucket_begin(map); }
#endif
// dev_bucket_begin for ONE WAVE of a larger workgroup (the scan-resident stream kernel's insert wave): no workgroup barrier
__device__ __forceinline__ void dev_bucket_begin_wave(const LkMap& map) {
    const int lane = threadIdx.x & 63;
    int base = 0, nfreed = 0;
    if (lane == 0) {
        const int fc = (int)map.counters[LK_CTR_FREE];
        base = fc > 0 ? fc : 0;
        const unsigned int nf = map.counters[LK_CTR_FREED];
        nfreed = (int)(nf < map.max_blocks ? nf : map.max_blocks);
        if (base + nfreed > (int)map.max_blocks) nfreed = (int)map.max_blocks - base;
        map.counters[LK_CTR_TOUCHED] = 0;
        map.counters[LK_CTR_SCRATCH] = 0;
        map.counters[LK_CTR_HEAVY] = 0;
        map.counters[LK_CTR_GROUPS] = 0;
        map.counters[LK_CTR_GIDX] = 0;
        map.counters[LK_CTR_FALLBACK] = 0;
    }
    base = bcast0(base), nfreed = bcast0(nfreed);
    for (int i = lane; i < nfreed; i += LK_WAVE) map.free_list[base + i] = map.freed_next[i];
    wave_fence();
    if (lane == 0) {
        map.counters[LK_CTR_FREE] = (unsigned int)(base + nfreed);
        map.counters[LK_CTR_FREED] = 0;
    }
    wave_fence();
}
// Stamp, BEFORE any of them is processed, every touched root whose planes may change in this bucket (the rule of dev_insert_root):
// one lane per root.  After this pass LkMap::dirty is final for the bucket - what a speculative residual pass running beside the
// insert needs to know to decide whether it has to wait for it (scan-resident stream kernel).
__device__ __forceinline__ void dev_stamp_dirty_roots(const LkMap& map, const LkParams& pr, int n_touched) {
    const int lane = threadIdx.x & 63;
    for (int t = lane; t < n_touched; t += LK_WAVE) {
        const int root = map.touched[t];
        const lk_node_rec* nd = &map.nodes[root];
        if (!root_is_light(pr, (int)nd->pad_[0], nd->state, map.planes[root].flags, nd->npts, nd->new_points)) map.dirty[root] = map.epoch;
    }
}

// ------------------------------------------------------------------ pool initialisation
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_pool_init_kernel(LkMap map, unsigned int n_hash);
#else
__global__ void __launch_bounds__(256) lk_pool_init_kernel(LkMap map, unsigned int n_hash) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_hash) map.hash[i] = make_int4((int)0x80000000, (int)0x80000000, (int)0x80000000, LK_EMPTY);
    if (i < map.max_nodes) {
        lk_node_rec* nd = &map.nodes[i];
        for (int c = 0; c < 8; ++c) nd->child[c] = -1;
        nd->voxel_center[0] = nd->voxel_center[1] = nd->voxel_center[2] = 0.0;
        nd->quater_length = 0.f;
        nd->layer = 0, nd->npts = 0, nd->new_points = 0, nd->state = 0, nd->block = -1;
        nd->key[0] = nd->key[1] = nd->key[2] = 0;
        nd->list_head = -1;
        for (int c = 0; c < 8; ++c) nd->pad_[c] = 0;
        map.planes[i].flags = 0;
        map.match[i].flags = 0;
    }
    if (i < LK_CTR_COUNT) map.counters[i] = 0;
}
#endif

// ---- frozen-map grid (LkMap::grid): bounding box of the root keys, then one cell per root
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_grid_bounds_kernel(LkMap map, unsigned int n_hash, int* __restrict__ mm /* min xyz, max xyz */);
#else
__global__ void __launch_bounds__(256) lk_grid_bounds_kernel(LkMap map, unsigned int n_hash, int* __restrict__ mm /* min xyz, max xyz */) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
    if (i < n_hash) {
        const int4 e = map.hash[i];
        if (e.w >= 0) lo[0] = hi[0] = e.x, lo[1] = hi[1] = e.y, lo[2] = hi[2] = e.z;
    }
    // one set of atomics per wave, not per occupied slot (all of them hit the same six words)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = min(lo[c], __shfl_xor(lo[c], o, LK_WAVE));
            hi[c] = max(hi[c], __shfl_xor(hi[c], o, LK_WAVE));
        }
    }
    if ((threadIdx.x & 63) == 0 && lo[0] <= hi[0]) {
        atomicMin(&mm[0], lo[0]), atomicMin(&mm[1], lo[1]), atomicMin(&mm[2], lo[2]);
        atomicMax(&mm[3], hi[0]), atomicMax(&mm[4], hi[1]), atomicMax(&mm[5], hi[2]);
    }
}
#endif
// One thread per root: a plane root's record goes into its cell; any other root gets a list header and its subtree's plane
// nodes, in pre-order (children in index order, a plane is not descended into, nothing below max_layer), appended behind the
// grid (`cursor` = next free record; the order of the lists among each other does not matter).
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_grid_fill_kernel(LkMap map, unsigned int n_hash, int max_layer, unsigned int* __restrict__ cursor,
                                                           unsigned int cand_end);
#else
__global__ void __launch_bounds__(256) lk_grid_fill_kernel(LkMap map, unsigned int n_hash, int max_layer, unsigned int* __restrict__ cursor,
                                                           unsigned int cand_end) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hash) return;
    const int4 e = map.hash[i];
    if (e.w < 0) return;
    const size_t c = ((size_t)(e.z - map.gmin[2]) * (size_t)map.gdim[1] + (size_t)(e.y - map.gmin[1])) * (size_t)map.gdim[0] + (size_t)(e.x - map.gmin[0]);
    lk_match_rec r = map.match[e.w];
    r.pad_ = (unsigned int)e.w;
    if (!(r.flags & LK_PLANE_IS_PLANE)) {
        // pass 1: count the plane nodes of the subtree; pass 2: copy them
        unsigned int first = 0, count = 0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                if (count == 0) break;
                first = atomicAdd(cursor, count);
                if (first + count > cand_end) {
                    // The list region is sized for every non-root node once.  A map whose child links share nodes (only an imported
                    // blob could: lk_map_import_dev range-checks ids, not the tree shape) can need more: flag it - frozen_map() then
                    // keeps this snapshot on the hash table instead of silently matching against a root without candidates.
                    atomicExch(cursor + 1, 1u);
                    count = 0;
                    break;
                }
            }
            unsigned int k = 0;
            int stack_node[LK_MAX_LAYER + 1], stack_ci[LK_MAX_LAYER + 1];
            int level = 0;
            stack_node[0] = e.w, stack_ci[0] = 0;
            while (level >= 0) {
                const int node = stack_node[level];
                if (stack_ci[level] == 0 && level > 0) {   // first visit of a non-root node
                    const unsigned int fl = map.match[node].flags;
                    if (fl & LK_PLANE_IS_PLANE) {
                        if (pass == 1) {
                            lk_match_rec cr = map.match[node];
                            cr.pad_ = (unsigned int)node;
                            map.match[first + k] = cr;
                        }
                        ++k;
                        --level;
                        continue;
                    }
                }
                if (level >= max_layer || level >= LK_MAX_LAYER) {
                    --level;
                    continue;
                }
                int child = -1;
                while (stack_ci[level] < 8 && child < 0) child = map.nodes[node].child[stack_ci[level]++];
                if (child >= 0) {
                    ++level;
                    stack_node[level] = child, stack_ci[level] = 0;
                } else {
                    --level;
                }
            }
            if (pass == 0) count = k;
        }
        r.flags = LK_GRID_LIST;
        unsigned int fc[2] = {first, count};
        memcpy(&r.center[0], fc, sizeof(fc));
    }
    map.match[map.grid_base + c] = r;
}
#endif

// derive the compact match records of imported planes (lk_map_import / lk_map_import_dev)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_derive_match_kernel(LkMap map, int n);
#else
__global__ void __launch_bounds__(256) lk_derive_match_kernel(LkMap map, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (map.planes[i].flags & LK_PLANE_IS_PLANE)
        lk_derive_match(&map.planes[i], &map.match[i]);
    else
        map.match[i].flags = map.planes[i].flags;
}
#endif

// rebuild the hash from imported root records (lk_map_import)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_hash_insert_kernel(LkMap map, LkParams pr, const lk_root_rec* __restrict__ roots, int n);
#else
__global__ void __launch_bounds__(256) lk_hash_insert_kernel(LkMap map, LkParams pr, const lk_root_rec* __restrict__ roots, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned int s = lk_hash3(roots[i].key[0], roots[i].key[1], roots[i].key[2]) & map.hash_mask;
    int* slotw = reinterpret_cast<int*>(map.hash);
    for (unsigned int trips = 0; trips <= map.hash_mask; ++trips) {
        if (atomicCAS(&slotw[4 * s + 3], LK_EMPTY, roots[i].node) == LK_EMPTY) {
            slotw[4 * s + 0] = roots[i].key[0];
            slotw[4 * s + 1] = roots[i].key[1];
            slotw[4 * s + 2] = roots[i].key[2];
            return;
        }
        s = (s + 1) & map.hash_mask;
    }
    atomicOr(&map.counters[LK_CTR_ERR], LK_E_HASH_FULL);
}
#endif

// ------------------------------------------------------------------ local map sliding (voxel_map.cc:552-594)
// clearMemOutOfMap deletes every root voxel (and its whole octree) whose key lies strictly outside a key-space box.
// Here that is a COMPACTION of the pools, so a long run's device memory is bounded by the live map:
//   lk_slide_mark_kernel   one thread per hash slot: roots inside the box are appended to `kept` and their subtree's
//                          nodes / point blocks flagged alive; roots outside are counted
//   (rocPRIM exclusive scans of the two flag arrays give the dense new ids; order = old order, deterministic)
//   lk_slide_move_*        copy alive records to their new index in temporaries, child / block links renumbered
//   lk_slide_hash_clear + lk_slide_remap_roots + lk_hash_insert_kernel   rebuild the table from the kept roots
// Nothing on the per-bucket path changes: allocation stays a bump pointer (+ the point-block free list, which is
// emptied here because retired blocks are not alive).
struct LkSlideBox {
    int x_max, x_min, y_max, y_min, z_max, z_min;
};

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256)
    lk_slide_mark_kernel(LkMap map, LkSlideBox box, unsigned int n_hash, unsigned int* __restrict__ alive_node,
                         unsigned int* __restrict__ alive_block, lk_root_rec* __restrict__ kept,
                         unsigned int* __restrict__ cnt /* [0] kept, [1] removed */);
#else
__global__ void __launch_bounds__(256)
    lk_slide_mark_kernel(LkMap map, LkSlideBox box, unsigned int n_hash, unsigned int* __restrict__ alive_node,
                         unsigned int* __restrict__ alive_block, lk_root_rec* __restrict__ kept,
                         unsigned int* __restrict__ cnt /* [0] kept, [1] removed */) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hash) return;
    const int4 e = map.hash[i];
    if (e.w < 0) return;
    const bool should_remove = e.x > box.x_max || e.x < box.x_min || e.y > box.y_max || e.y < box.y_min ||
                               e.z > box.z_max || e.z < box.z_min;  // voxel_map.cc:578-579
    if (should_remove) {
        atomicAdd(&cnt[1], 1u);
        return;
    }
    const unsigned int k = atomicAdd(&cnt[0], 1u);
    kept[k].key[0] = e.x, kept[k].key[1] = e.y, kept[k].key[2] = e.z;
    kept[k].node = e.w;
    // pre-order walk of the subtree (depth <= LK_MAX_LAYER)
    int st_node[LK_MAX_LAYER + 2], st_ci[LK_MAX_LAYER + 2];
    int depth = 0;
    st_node[0] = e.w, st_ci[0] = 0;
    alive_node[e.w] = 1u;
    {
        const int b = map.nodes[e.w].block;
        if (b >= 0) alive_block[b] = 1u;
    }
    while (depth >= 0) {
        const int ci = st_ci[depth];
        if (ci == 8) {
            --depth;
            continue;
        }
        st_ci[depth] = ci + 1;
        const int child = map.nodes[st_node[depth]].child[ci];
        if (child < 0) continue;
        alive_node[child] = 1u;
        const int b = map.nodes[child].block;
        if (b >= 0) alive_block[b] = 1u;
        if (depth + 1 < LK_MAX_LAYER + 2) {
            ++depth;
            st_node[depth] = child, st_ci[depth] = 0;
        }
    }
}
#endif

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256)
    lk_slide_move_nodes_kernel(LkMap map, unsigned int n_nodes, const unsigned int* __restrict__ alive_node,
                               const unsigned int* __restrict__ new_node, const unsigned int* __restrict__ new_block,
                               lk_node_rec* __restrict__ tn, lk_plane_rec* __restrict__ tp, lk_match_rec* __restrict__ tm);
#else
__global__ void __launch_bounds__(256)
    lk_slide_move_nodes_kernel(LkMap map, unsigned int n_nodes, const unsigned int* __restrict__ alive_node,
                               const unsigned int* __restrict__ new_node, const unsigned int* __restrict__ new_block,
                               lk_node_rec* __restrict__ tn, lk_plane_rec* __restrict__ tp, lk_match_rec* __restrict__ tm) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes || !alive_node[i]) return;
    const unsigned int j = new_node[i];
    lk_node_rec r = map.nodes[i];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (r.child[c] >= 0) r.child[c] = (int)new_node[r.child[c]];
    if (r.block >= 0) r.block = (int)new_block[r.block];
    r.list_head = -1;  // bucket-local queue state is not carried across a slide
    r.pad_[0] = 0;
    tn[j] = r;
    tp[j] = map.planes[i];
    tm[j] = map.match[i];
}
#endif

// one 64-thread workgroup per alive point block (3744 B = 234 x 16 B)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(64)
    lk_slide_move_blocks_kernel(LkMap map, const unsigned int* __restrict__ alive_block,
                                const unsigned int* __restrict__ new_block, lk_block_rec* __restrict__ tb);
#else
__global__ void __launch_bounds__(64)
    lk_slide_move_blocks_kernel(LkMap map, const unsigned int* __restrict__ alive_block,
                                const unsigned int* __restrict__ new_block, lk_block_rec* __restrict__ tb) {
    const unsigned int b = blockIdx.x;
    if (!alive_block[b]) return;
    const uint4* src = reinterpret_cast<const uint4*>(&map.blocks[b]);
    uint4* dst = reinterpret_cast<uint4*>(&tb[new_block[b]]);
    for (unsigned int k = threadIdx.x; k < sizeof(lk_block_rec) / 16; k += 64) dst[k] = src[k];
}
#endif

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256) lk_slide_hash_clear_kernel(LkMap map, unsigned int n_hash);
#else
__global__ void __launch_bounds__(256) lk_slide_hash_clear_kernel(LkMap map, unsigned int n_hash) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_hash) map.hash[i] = make_int4((int)0x80000000, (int)0x80000000, (int)0x80000000, LK_EMPTY);
}
#endif

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(256)
    lk_slide_remap_roots_kernel(lk_root_rec* __restrict__ kept, unsigned int n, const unsigned int* __restrict__ new_node);
#else
__global__ void __launch_bounds__(256)
    lk_slide_remap_roots_kernel(lk_root_rec* __restrict__ kept, unsigned int n, const unsigned int* __restrict__ new_node) {
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) kept[i].node = (int)new_node[kept[i].node];
}
#endif

#ifdef LK_KERNELS_ELSEWHERE
__global__ void lk_slide_counters_kernel(LkMap map, unsigned int n_nodes, unsigned int n_blocks, unsigned int n_roots);
#else
__global__ void lk_slide_counters_kernel(LkMap map, unsigned int n_nodes, unsigned int n_blocks, unsigned int n_roots) {
    map.counters[LK_CTR_NODES] = n_nodes;
    map.counters[LK_CTR_BLOCKS] = n_blocks;
    map.counters[LK_CTR_ROOTS] = n_roots;
    map.counters[LK_CTR_FREE] = 0;
    map.counters[LK_CTR_FREED] = 0;
}
#endif

