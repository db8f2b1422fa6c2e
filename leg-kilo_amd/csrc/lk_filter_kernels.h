// lk_filter_kernels.h — single-workgroup ESKF kernels (one 256-thread block per filter slot).
//   lk_predict_kernel     ESKF::predict x2 as issued by KILO.cc:111-115              (eskf.cc:64-89)
//   lk_update_kernel      reduce block partials -> 6x6 information-form update       (eskf.cc:91-113)
//   lk_update_wave_kernel (+ _ragged)  the same + the next bucket's predict as ONE WAVE per slot, for batch replay
//   lk_imu_kernel         predictUpdateImu  (KILO.cc:235-258, eskf.cc:125-135)
//   lk_kin_kernel         predictUpdateKinImu (KILO.cc:260-314, eskf.cc:137-145)
//   lk_obs_update_kernel  updateByPoints / updateByKinImu on caller-supplied rows (class-surface calls)
// All dense 30x30 algebra runs out of LDS in fp64 (P is 7.2 KB).  One filter is latency-, not
// throughput-bound; the grid dimension is the slot index so that batch replay updates many
// filters per launch.
#pragma once
#include "lk_device.h"

#define LK_FB 256  // threads per filter block
#ifndef LK_MFMA_COV
#define LK_MFMA_COV 0   // 1 (A/B build): the covariance update P -= P[:,0:6] X of the one-wave update core on v_mfma_f64_16x16x4_f64
#endif
#ifndef LK_X_P
#define LK_X_P 0   // perf attribution only (never set in the product build): bit 1 skip the rotations of the wave predict, 2 the
#endif             // covariance products, 4 the Q term, 8 the Gauss-Jordan sweep of the wave update, 16 its P update, 32 its (+)

struct FilterSmem {
    double P[900];
    double A[900];
    double B[900];
    double H[18 * 30];
    double PHT[30 * 18];
    double S[18 * 18];
    double G[18 * 31];
    double vec[64];
    double fac[32];
    int piv;
};

// ---- state (+) delta, eskf.cc:18-29 (thread 0 only)
__device__ __forceinline__ void state_boxplus(double* x, const double* d) {
    double E[9], Rn[9];
    exp3_1e5(d[0], d[1], d[2], E);
    mat3_mul(x, E, Rn);
#pragma unroll
    for (int i = 0; i < 9; ++i) x[i] = Rn[i];
#pragma unroll
    for (int i = 0; i < 27; ++i) x[9 + i] += d[3 + i];
}

// ---- ESKF::predict(dt_cov,false,true) then predict(dt,true,false), KILO.cc:111-115
// On exit sm.P holds the propagated covariance (also written to f->P) and f->x is propagated.
__device__ void dev_predict(LkFilter* f, const double* __restrict__ Q, double t, FilterSmem& sm) {
    const int tid = threadIdx.x;
    const double dt_cov = t - f->last_update_t;
    const double dt = t - f->last_predict_t;
    for (int i = tid; i < 900; i += LK_FB) sm.P[i] = f->P[i];
    if (tid < 36) sm.vec[tid] = f->x[tid];
    __syncthreads();
    // The two rotations of a predict depend on the old state only: thread 0 builds the non-trivial blocks of Fx (getFx,
    // eskf.cc:72-81: E = Exp(-dt_cov w) -> sm.A[0..8], B60 = -dt_cov R [a]x -> sm.A[9..17]) while thread 64 - another wave -
    // propagates the state (getFunctionf + operator+=, eskf.cc:64-70,18-29) into its registers.
    double Rn[9], dpv[6];
    if (tid == 0 || tid == 64) {
        const double* x = sm.vec;
        const double sc = tid == 0 ? -dt_cov : dt;
        const double thr = tid == 0 ? 0.0000001 : 0.00001;   // math_utils.hpp:19-32 / :54-68
        double E[9];
        exp_so3_thr(sc * x[27], sc * x[28], sc * x[29], thr, E);
        if (tid == 0) {
            V3 a = V3{x[24], x[25], x[26]};
            double K[9], mR[9], B60[9];
            skew3(a, K);
            for (int i = 0; i < 9; ++i) mR[i] = (-dt_cov) * x[i];
            mat3_mul(mR, K, B60);
            for (int i = 0; i < 9; ++i) sm.A[i] = E[i], sm.A[9 + i] = B60[i];
        } else {
            V3 Ra = mat3_mul_v(x, V3{x[24], x[25], x[26]});
            for (int i = 0; i < 3; ++i) dpv[i] = dt * x[12 + i];
            dpv[3] = dt * (Ra.x + x[21]), dpv[4] = dt * (Ra.y + x[22]), dpv[5] = dt * (Ra.z + x[23]);
            mat3_mul(x, E, Rn);
        }
    }
    __syncthreads();
    // Fx P Fx^T over the three non-identity row blocks of Fx, each as one straight-line expression over its non-zero terms
    // (rows 0..2 = [E | dt I at 21], rows 3..5 = [I | dt I at 6], rows 6..8 = [B60 | I | dt I at 15 | dt R at 18]): the same
    // sums in the same order as the dense 30-term dot products (the skipped terms are exact zeros) and as wave_predict_core.
    const double* E = sm.A;
    const double* B60 = sm.A + 9;
    const double* xr = sm.vec;
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (tid < 90) {
        const int i = tid / 30, c = tid % 30;
        double s = 0.0;
        s += E[3 * i + 0] * sm.P[0 * 30 + c];
        s += E[3 * i + 1] * sm.P[1 * 30 + c];
        s += E[3 * i + 2] * sm.P[2 * 30 + c];
        s += dt_cov * sm.P[(21 + i) * 30 + c];
        n0 = s;
        s = 0.0;
        s += 1.0 * sm.P[(3 + i) * 30 + c];
        s += dt_cov * sm.P[(6 + i) * 30 + c];
        n1 = s;
        s = 0.0;
        s += B60[3 * i + 0] * sm.P[0 * 30 + c];
        s += B60[3 * i + 1] * sm.P[1 * 30 + c];
        s += B60[3 * i + 2] * sm.P[2 * 30 + c];
        s += 1.0 * sm.P[(6 + i) * 30 + c];
        s += dt_cov * sm.P[(15 + i) * 30 + c];
        s += (dt_cov * xr[3 * i + 0]) * sm.P[18 * 30 + c];
        s += (dt_cov * xr[3 * i + 1]) * sm.P[19 * 30 + c];
        s += (dt_cov * xr[3 * i + 2]) * sm.P[20 * 30 + c];
        n2 = s;
    }
    __syncthreads();
    if (tid < 90) sm.P[tid] = n0, sm.P[90 + tid] = n1, sm.P[180 + tid] = n2;
    __syncthreads();
    if (tid < 90) {
        const int i = tid / 3, c = tid % 3;
        const double* Bi = &sm.P[i * 30];
        double s = 0.0;
        s += Bi[0] * E[3 * c + 0];
        s += Bi[1] * E[3 * c + 1];
        s += Bi[2] * E[3 * c + 2];
        s += Bi[21 + c] * dt_cov;
        n0 = s;
        s = 0.0;
        s += Bi[3 + c] * 1.0;
        s += Bi[6 + c] * dt_cov;
        n1 = s;
        s = 0.0;
        s += Bi[0] * B60[3 * c + 0];
        s += Bi[1] * B60[3 * c + 1];
        s += Bi[2] * B60[3 * c + 2];
        s += Bi[6 + c] * 1.0;
        s += Bi[15 + c] * dt_cov;
        s += Bi[18] * (dt_cov * xr[3 * c + 0]);
        s += Bi[19] * (dt_cov * xr[3 * c + 1]);
        s += Bi[20] * (dt_cov * xr[3 * c + 2]);
        n2 = s;
    }
    __syncthreads();
    if (tid < 90) {
        const int i = tid / 3, c = tid % 3;
        sm.P[i * 30 + c] = n0, sm.P[i * 30 + 3 + c] = n1, sm.P[i * 30 + 6 + c] = n2;
    }
    __syncthreads();
    const double dt2 = dt_cov * dt_cov;
    for (int e = tid; e < 900; e += LK_FB) {  // P += dt^2 Q
        double v = sm.P[e] + dt2 * Q[e];
        sm.P[e] = v;
        f->P[e] = v;
    }
    if (tid == 64) {
        double* x = f->x;
        for (int i = 0; i < 9; ++i) x[i] = Rn[i];
        for (int i = 0; i < 6; ++i) x[9 + i] += dpv[i];
        f->last_predict_t = t;
    }
    __syncthreads();
}

// ---- X = S^-1 G by Gauss-Jordan with partial pivoting, in place (G := X).  S: M x M (ld 18),
// G: M x NG (ld 31).  Block-parallel over the trailing entries.
__device__ void dev_solve(FilterSmem& sm, int M, int NG) {
    const int tid = threadIdx.x;
    for (int k = 0; k < M; ++k) {
        if (tid == 0) {
            int p = k;
            double best = fabs(sm.S[k * 18 + k]);
            for (int i = k + 1; i < M; ++i) {
                double v = fabs(sm.S[i * 18 + k]);
                if (v > best) best = v, p = i;
            }
            sm.piv = p;
        }
        __syncthreads();
        const int p = sm.piv;
        if (p != k) {
            for (int c = tid; c < M + NG; c += LK_FB) {
                double* a = (c < M) ? &sm.S[k * 18 + c] : &sm.G[k * 31 + (c - M)];
                double* b = (c < M) ? &sm.S[p * 18 + c] : &sm.G[p * 31 + (c - M)];
                double t = *a;
                *a = *b;
                *b = t;
            }
        }
        __syncthreads();
        if (tid < M) sm.fac[tid] = (tid == k) ? 0.0 : sm.S[tid * 18 + k] / sm.S[k * 18 + k];
        __syncthreads();
        for (int e = tid; e < M * (M + NG); e += LK_FB) {
            int i = e / (M + NG), c = e % (M + NG);
            if (i == k) continue;
            double fi = sm.fac[i];
            if (c < M)
                sm.S[i * 18 + c] -= fi * sm.S[k * 18 + c];
            else
                sm.G[i * 31 + (c - M)] -= fi * sm.G[k * 31 + (c - M)];
        }
        __syncthreads();
    }
    for (int e = tid; e < M * NG; e += LK_FB) {
        int i = e / NG, c = e % NG;
        sm.G[i * 31 + c] = sm.G[i * 31 + c] / sm.S[i * 18 + i];
    }
    __syncthreads();
}

// ---- dx = PHT X[:,30];  P <- P - PHT X[:,0:30];  x (+)= dx.   PHT: 30 x M (ld 18), X = sm.G.
__device__ void dev_kalman_apply(LkFilter* f, FilterSmem& sm, int M) {
    const int tid = threadIdx.x;
    if (tid < 30) {
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += sm.PHT[tid * 18 + m] * sm.G[m * 31 + 30];
        sm.vec[tid] = s;
    }
    for (int e = tid; e < 900; e += LK_FB) {
        int i = e / 30, j = e % 30;
        double s = 0.0;
        for (int m = 0; m < M; ++m) s += sm.PHT[i * 18 + m] * sm.G[m * 31 + j];
        f->P[e] = sm.P[e] - s;
    }
    __syncthreads();
    if (tid == 0) state_boxplus(f->x, sm.vec);
    __syncthreads();
}

#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB) lk_predict_kernel(LkFilter* filters, const double* __restrict__ Q, double t);
#else
__global__ void __launch_bounds__(LK_FB) lk_predict_kernel(LkFilter* filters, const double* __restrict__ Q, double t) {
    __shared__ FilterSmem sm;
    dev_predict(&filters[blockIdx.x], Q, t, sm);
}
#endif

// plain ESKF::predict(dt, prop_state, prop_cov) for the class-surface call (eskf.cc:83-89)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB)
    lk_predict_dt_kernel(LkFilter* filters, const double* __restrict__ Q, double dt, int prop_state, int prop_cov);
#else
__global__ void __launch_bounds__(LK_FB)
    lk_predict_dt_kernel(LkFilter* filters, const double* __restrict__ Q, double dt, int prop_state, int prop_cov) {
    __shared__ FilterSmem sm;
    LkFilter* f = &filters[blockIdx.x];
    // dev_predict derives its two dt's from the stored times; emulate with temporaries
    const int tid = threadIdx.x;
    __shared__ double save[2];
    if (tid == 0) {
        save[0] = f->last_predict_t, save[1] = f->last_update_t;
    }
    __syncthreads();
    // order inside predict(): state first, then covariance with the NEW state (eskf.cc:84-88)
    if (prop_state) {
        if (tid == 0) {
            double* x = f->x;
            double d[30];
            for (int i = 0; i < 30; ++i) d[i] = 0.0;
            V3 Ra = mat3_mul_v(x, V3{x[24], x[25], x[26]});
            for (int i = 0; i < 3; ++i) d[i] = dt * x[27 + i], d[3 + i] = dt * x[12 + i];
            d[6] = dt * (Ra.x + x[21]), d[7] = dt * (Ra.y + x[22]), d[8] = dt * (Ra.z + x[23]);
            state_boxplus(x, d);
        }
        __syncthreads();
    }
    if (prop_cov) {
        if (tid == 0) {
            f->last_update_t = 0.0;   // dt_cov = dt - 0
            f->last_predict_t = dt;   // dt_state = 0 -> identity state step
        }
        __syncthreads();
        dev_predict(f, Q, dt, sm);
    }
    __syncthreads();
    if (tid == 0) {
        f->last_predict_t = save[0], f->last_update_t = save[1];
    }
}
#endif

// getFx / getFunctionf read-outs for the class surface
#ifdef LK_KERNELS_ELSEWHERE
__global__ void lk_fx_kernel(const LkFilter* filters, int slot, double dt, double* Fx, double* fvec);
#else
__global__ void lk_fx_kernel(const LkFilter* filters, int slot, double dt, double* Fx, double* fvec) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double* x = filters[slot].x;
    for (int i = 0; i < 900; ++i) Fx[i] = ((i / 30) == (i % 30)) ? 1.0 : 0.0;
    V3 w = V3{x[27], x[28], x[29]}, a = V3{x[24], x[25], x[26]};
    double E[9], K[9], mR[9], B60[9];
    expv_1e7(V3{(-dt) * w.x, (-dt) * w.y, (-dt) * w.z}, E);
    skew3(a, K);
    for (int i = 0; i < 9; ++i) mR[i] = (-dt) * x[i];
    mat3_mul(mR, K, B60);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            Fx[(0 + i) * 30 + 0 + j] = E[3 * i + j];
            Fx[(0 + i) * 30 + 21 + j] = (i == j) ? dt : 0.0;
            Fx[(3 + i) * 30 + 6 + j] = (i == j) ? dt : 0.0;
            Fx[(6 + i) * 30 + 0 + j] = B60[3 * i + j];
            Fx[(6 + i) * 30 + 15 + j] = (i == j) ? dt : 0.0;
            Fx[(6 + i) * 30 + 18 + j] = dt * x[3 * i + j];
        }
    for (int i = 0; i < 30; ++i) fvec[i] = 0.0;
    V3 Ra = mat3_mul_v(x, a);
    for (int i = 0; i < 3; ++i) fvec[i] = dt * x[27 + i], fvec[3 + i] = dt * x[12 + i];
    fvec[6] = dt * (Ra.x + x[21]), fvec[7] = dt * (Ra.y + x[22]), fvec[8] = dt * (Ra.z + x[23]);
}
#endif

// ---- information-form point update from A (21, upper tri), b (6):  eskf.cc:91-113 via
//   S = I6 + A P66,  G = [A P[0:6,:] | b],  X = S^-1 G,  dx = P[:,0:6] X[:,30],  P -= P[:,0:6] X[:,0:30]
__device__ void dev_point_update(LkFilter* f, FilterSmem& sm, const double* A21, const double* b6) {
    const int tid = threadIdx.x;
    for (int i = tid; i < 900; i += LK_FB) sm.P[i] = f->P[i];
    if (tid < 36) {  // expand symmetric A into sm.A[0..35]
        int i = tid / 6, j = tid % 6;
        int r = i < j ? i : j, c = i < j ? j : i;
        sm.A[tid] = A21[r * 6 - r * (r - 1) / 2 + (c - r)];
    }
    __syncthreads();
    for (int e = tid; e < 6 * 30; e += LK_FB) {  // G[:,0:30] = A * P[0:6,:]
        int i = e / 30, j = e % 30;
        double s = 0.0;
        for (int k = 0; k < 6; ++k) s += sm.A[i * 6 + k] * sm.P[k * 30 + j];
        sm.G[i * 31 + j] = s;
    }
    if (tid < 6) sm.G[tid * 31 + 30] = b6[tid];
    for (int e = tid; e < 30 * 6; e += LK_FB) sm.PHT[(e / 6) * 18 + (e % 6)] = sm.P[(e / 6) * 30 + (e % 6)];
    __syncthreads();
    if (tid < 36) {  // S = I + A * P66  (= I + G[:,0:6])
        int i = tid / 6, j = tid % 6;
        sm.S[i * 18 + j] = ((i == j) ? 1.0 : 0.0) + sm.G[i * 31 + j];
    }
    __syncthreads();
    dev_solve(sm, 6, 31);
    dev_kalman_apply(f, sm, 6);
}

#ifndef LK_BLOCK_UPDATE_WAVE
#define LK_BLOCK_UPDATE_WAVE 1   // the 256-thread kernels solve and apply the point update in their wave 0 (0 = dev_point_update, barrier-separated)
#endif
__device__ void dev_point_update_wave0(LkFilter* f, FilterSmem& sm, const double* tot, int N);   // defined behind wave_update_core
// The bucket's totals [A(21) b(6) sumR count] are in tot[] (LDS, visible to the whole workgroup): bookkeeping of
// KILO.cc:193,211-212 and the information-form update.  Called by all LK_FB threads.
__device__ void dev_update_from_totals(LkFilter* f, FilterSmem& sm, double* tot, double t) {
    const int tid = threadIdx.x;
    const int N = (int)(tot[28] + 0.5);
    if (tid == 0) {
        f->n_buckets += 1;
        f->last_N = N;
        f->updated = N > 0;
        if (N > 0) {
            f->n_updates += 1;
            f->n_effect += (unsigned long long)N;
            f->last_update_t = t;  // KILO.cc:212
        }
    }
    if (N > 0 && LK_BLOCK_UPDATE_WAVE) {
        dev_point_update_wave0(f, sm, tot, N);
    } else if (N > 0) {
        if (N == 1) {  // eskf.cc:98-104: s = 1/(0.0001 + hPh^T + r)  <=>  r' = r + 1e-4
            double r = tot[27];
            double sc = r / (r + 0.0001);
            __syncthreads();
            if (tid < 27) tot[tid] *= sc;
        }
        __syncthreads();
        dev_point_update(f, sm, &tot[0], &tot[21]);
    }
}

// reduce the per-wave partial records (fixed order -> deterministic) and update; partials: [nblk][LK_NPART] per slot.
// do_predict != 0 (batch replay on a frozen map, where nothing reads the state between update(k) and predict(k+1)):
// the predict of the NEXT bucket (time t_next) runs in the same launch.
// the bucket's totals from the per-wave partial records, in a fixed order (all LK_FB threads; tot[] is valid behind the last barrier)
__device__ __forceinline__ void dev_reduce_partials(const double* __restrict__ part, int nblk, double (*red)[LK_NPART], double* tot) {
    const int tid = threadIdx.x;
    {
        const int j = tid % LK_NPART, g = tid / LK_NPART;  // 8 groups x 32 components
        // four independent accumulators keep four loads in flight; combined in a fixed order
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = g;
        for (; b + 24 < nblk; b += 32) {
            s0 += part[(size_t)b * LK_NPART + j];
            s1 += part[(size_t)(b + 8) * LK_NPART + j];
            s2 += part[(size_t)(b + 16) * LK_NPART + j];
            s3 += part[(size_t)(b + 24) * LK_NPART + j];
        }
        for (; b < nblk; b += 8) s0 += part[(size_t)b * LK_NPART + j];
        red[g][j] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (tid < LK_NPART) {
        double s = 0.0;
        for (int g = 0; g < 8; ++g) s += red[g][tid];
        tot[tid] = s;
    }
    __syncthreads();
}
__device__ __forceinline__ void dev_update_reduce(LkFilter* f, const double* __restrict__ part, int nblk, double t, const double* __restrict__ Q,
                                                  double t_next, int do_predict, FilterSmem& sm, double (*red)[LK_NPART], double* tot) {
    dev_reduce_partials(part, nblk, red, tot);
    dev_update_from_totals(f, sm, tot, t);
    if (do_predict) {
        __syncthreads();  // f->x, f->P, f->last_update_t written above are re-read by dev_predict
        dev_predict(f, Q, t_next, sm);
    }
}
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB)
    lk_update_kernel(LkFilter* filters, const double* __restrict__ partials, int nblk, size_t slot_stride, double t,
                     const double* __restrict__ Q, double t_next, int do_predict);
#else
__global__ void __launch_bounds__(LK_FB)
    lk_update_kernel(LkFilter* filters, const double* __restrict__ partials, int nblk, size_t slot_stride, double t,
                     const double* __restrict__ Q, double t_next, int do_predict) {
    __shared__ FilterSmem sm;
    __shared__ double red[8][LK_NPART];
    __shared__ double tot[LK_NPART];
    dev_update_reduce(&filters[blockIdx.x], partials + (size_t)blockIdx.x * slot_stride, nblk, t, Q, t_next, do_predict, sm, red, tot);
}
#endif
// What the insert of a bucket needs of the posterior (load_bucket_const: R, p, the rotation / position blocks of P - all in rows
// 0..5 of P - and `updated`), copied aside: in the pipelined stream path the insert runs on its own HIP stream while the main
// stream already propagates filters[0] to the next bucket.
__device__ __forceinline__ void dev_snapshot_posterior(const LkFilter* f, LkFilter* snap) {
    __syncthreads();   // the update's global writes (this workgroup's) are visible to all its threads
    const int tid = threadIdx.x;
    if (tid < LK_STATE_DOUBLES) snap->x[tid] = f->x[tid];
    if (tid < 180) snap->P[tid] = f->P[tid];
    if (tid == 0) snap->updated = f->updated, snap->last_N = f->last_N;
}
// ---- lk_update_kernel for batch replay as ONE WAVE per filter slot, in the resource footprint of a residual workgroup
// (64 threads, 7 680 B of LDS, <= 96 VGPRs).  A 256-thread update workgroup needs 40 KB of LDS and four wave slots at
// once; on a CU saturated with residual workgroups of another stream those never become free together, so the update of
// batch k could not run in the shadow of the residual launch of batch k+1.  This one takes exactly the slot a retiring
// residual wave leaves behind.  P lives in LDS (7 200 B), x and the 18 non-trivial constants of Fx in the remaining
// 480 B; everything else is in registers: the 6 x 37 augmented system [S | G] is held one COLUMN per lane, so the
// Gauss-Jordan sweep is lane-local apart from broadcasts of the pivot column.  Every sum is taken in the order
// lk_update_kernel takes it (the same grouping of the per-wave partials, the same dot-product order; terms of Fx that
// are exact zeros are skipped, which cannot change a finite sum), so the two kernels agree bit for bit -
// test_batch_replay_frozen_map compares the two entry points that use them.
// mode bit 0: reduce + update (eskf.cc:91-113 and the bookkeeping of KILO.cc:193,211-212);  bit 1: predict to t_next
// (KILO.cc:111-115) in the same launch.
#ifdef LK_DEBUG_RES
__device__ unsigned long long lk_core_dbg[16];   // DEBUG BUILD ONLY: 100 MHz ticks per phase of the one-wave cores (MW instances): [0..4] predict, [8..13] update, [7] / [15] calls
#define CORE_T0 unsigned long long ct0_ = MW ? wall_clock64() : 0ull
#define CORE_STAMP(k) do { if (MW) { const unsigned long long t1_ = wall_clock64(); if (lane == 0) atomicAdd(&lk_core_dbg[k], t1_ - ct0_); ct0_ = t1_; } } while (0)
#define CORE_COUNT(k) do { if (MW && lane == 0) atomicAdd(&lk_core_dbg[k], 1ull); } while (0)
#else
#define CORE_T0 do { } while (0)
#define CORE_STAMP(k) do { } while (0)
#define CORE_COUNT(k) do { } while (0)
#endif
struct WaveSmem {
    double P[900];
    double x[36];
    double fx[18];  // E (rows 0..2 of Fx, cols 0..2), then B60 (rows 6..8, cols 0..2): getFx, eskf.cc:72-81
    double pad_[6];
};
static_assert(sizeof(WaveSmem) == 7680, "one residual workgroup's worth of LDS");

// Barrier of the one-wave filter cores.  They were written for single-wave workgroups, where __syncthreads() is the wave's own
// barrier; MW = true lets ONE wave of a larger workgroup run them (the scan-resident stream kernel, legkilo_hip.hip).  The state they
// work on (WaveSmem and the row area) is touched by that wave alone, and the LDS executes one wave's instructions in issue order: what
// is needed is that the COMPILER keeps the order - wavefront-scope fences + a wave barrier, no s_waitcnt.  (Workgroup scope, as first
// written, drains the wave's LDS AND vector-memory queues at every one of the ~15 barriers of a bucket.)
#ifndef LK_CORE_SYNC_WG
#define LK_CORE_SYNC_WG 0
#endif
template <bool MW>
__device__ __forceinline__ void core_sync() {
    if (MW && LK_CORE_SYNC_WG) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else if (MW) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
__device__ __forceinline__ int tri6(int i, int j) {  // index of (i,j) in the packed upper triangle of a 6x6
    int r = i < j ? i : j, c = i < j ? j : i;
    return r * 6 - r * (r - 1) / 2 + (c - r);
}

// The N > 0 branch of updateByPoints on LDS-resident state: tot[j] in lanes 0..31 of totv (see dev_update_wave).
template <bool MW = false>
__device__ __forceinline__ void wave_update_core(WaveSmem& sm, double totv, int N, int lane) {
    if (N == 1) {  // eskf.cc:98-104
        double r = lane_bcast_u(totv, 27);
        double sc = r / (r + 0.0001);
        if (lane < 27) totv *= sc;
    }
    core_sync<MW>();  // sm.P, sm.x loaded
    CORE_T0;
    CORE_COUNT(15);
    // -- augmented column of this lane: lanes 0..5 S[:,lane] = I + (A P)[:,lane]; lanes 6..35 G[:,lane-6]; lane 36 b
    const int pc = lane < 6 ? lane : (lane < 36 ? lane - 6 : 29);
    double col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += lane_bcast_u(totv, tri6(i, k)) * sm.P[k * 30 + pc];
        double bi = lane_bcast_u(totv, 21 + i);
        col[i] = lane == 36 ? bi : (lane < 6 ? ((i == lane) ? 1.0 : 0.0) + s : s);
    }
    CORE_STAMP(8);
    // -- Gauss-Jordan with partial pivoting (dev_solve), one column per lane
#pragma unroll
    for (int k = 0; k < ((LK_X_P & 8) ? 0 : 6); ++k) {
        int p = k;
        double best = fabs(col[k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            double v = fabs(col[i]);
            if (v > best) best = v, p = i;
        }
        p = lane_bcast_u(p, k);  // the pivot search belongs to column k
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (i == p) {
                double tmp = col[k];
                col[k] = col[i];
                col[i] = tmp;
            }
        // row factors S[i][k] / S[k][k]: lane i divides ONE of them (the column lives in lane k's registers), then they are
        // broadcast - one fp64 division per step on the critical path instead of five; the same quotients, the same bits
        double sk[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) sk[i] = lane_bcast_u(col[i], k);
        const double mine = lane == 0 ? sk[0] : lane == 1 ? sk[1] : lane == 2 ? sk[2] : lane == 3 ? sk[3] : lane == 4 ? sk[4] : sk[5];
        const double fac = mine / sk[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == k) continue;
            const double fi = lane_bcast_u(fac, i);
            col[i] -= fi * col[k];
        }
    }
    CORE_STAMP(9);
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = col[i] / lane_bcast_u(col[i], i);  // X = G / diag(S)
    // -- dx = P[:,0:6] X[:,30]
    double dxv = 0.0;
    {
        const int i = lane < 30 ? lane : 29;
#pragma unroll
        for (int m = 0; m < 6; ++m) dxv += sm.P[i * 30 + m] * lane_bcast_u(col[m], 36);
        asm volatile("" : "+v"(dxv));  // finished here: keeps its six operands from living across the loop below
    }
#if LK_MFMA_COV
    // -- P -= P[:,0:6] X[:,0:30] on the matrix pipe (A/B build, -DLK_MFMA_COV=1): the 30 x 30 result as 2 x 2 tiles of
    // v_mfma_f64_16x16x4_f64, K = 6 padded to 8 -> eight instructions.  Operand layout measured on gfx950
    // (tools/probes/mfma_f64_16x16x4_layout.hip): lane l feeds A[l % 16][l / 16] and B[l / 16][l % 16]; d[v] of lane l is
    // D[l / 16 + 4 v][l % 16].  A = -P[:,0:6] from LDS, B = X gathered from the lanes that hold its columns (X[m][c] = col[m] of lane
    // 6 + c), C = P.  A different summation order than the VALU form below (k = 0..3, then 4..7): not bit-identical to it.
    {
        typedef double lk_d4 __attribute__((ext_vector_type(4)));
        const int r16 = lane & 15, k4 = lane >> 4;
        double Bop[2][2], Aop[2][2];   // [k-step][tile]
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int j = 16 * tj + r16;
            const int src = 6 + (j < 30 ? j : 0);
            double t[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) t[m] = __shfl(col[m], src, LK_WAVE);
            Bop[0][tj] = j < 30 ? (k4 == 0 ? t[0] : k4 == 1 ? t[1] : k4 == 2 ? t[2] : t[3]) : 0.0;
            Bop[1][tj] = (j < 30 && k4 < 2) ? (k4 == 0 ? t[4] : t[5]) : 0.0;
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int i = 16 * ti + r16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k = 4 * ks + k4;
                Aop[ks][ti] = (i < 30 && k < 6) ? -sm.P[i * 30 + k] : 0.0;
            }
        }
        lk_d4 acc[2][2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * ti + k4 + 4 * v, j = 16 * tj + r16;
                    acc[ti][tj][v] = (i < 30 && j < 30) ? sm.P[i * 30 + j] : 0.0;
                }
        core_sync<MW>();   // every operand has been read before any entry of P is rewritten
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[0][ti], Bop[0][tj], acc[ti][tj], 0, 0, 0);
                acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[1][ti], Bop[1][tj], acc[ti][tj], 0, 0, 0);
            }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * ti + k4 + 4 * v, j = 16 * tj + r16;
                    if (i < 30 && j < 30) sm.P[i * 30 + j] = acc[ti][tj][v];
                }
        core_sync<MW>();
    }
#else
    CORE_STAMP(10);
    // -- P -= P[:,0:6] X[:,0:30]: lane -> column lane % 30, rows 15 * (lane / 30) ...  A row's new values depend on
    // that row only, so five rows at a time are read, then written.
    {
        const int jc = lane % 30, i0 = lane < 60 ? 15 * (lane / 30) : 15;
        double X[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) X[m] = __shfl(col[m], 6 + jc, LK_WAVE);
#pragma unroll 1
        for (int r0 = 0; r0 < ((LK_X_P & 16) ? 0 : 15); r0 += 5) {
            double nv[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int i = i0 + r0 + r;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; ++m) s += sm.P[i * 30 + m] * X[m];
                nv[r] = sm.P[i * 30 + jc] - s;
            }
            core_sync<MW>();
            if (lane < 60) {
#pragma unroll
                for (int r = 0; r < 5; ++r) sm.P[(i0 + r0 + r) * 30 + jc] = nv[r];
            }
            core_sync<MW>();
        }
    }
#endif
    CORE_STAMP(11);
    // -- x (+)= dx (eskf.cc:18-29): rotation by lane 0, the 27 additive components by lanes 3..29
    const double d0 = lane_bcast_u(dxv, 0), d1 = lane_bcast_u(dxv, 1), d2 = lane_bcast_u(dxv, 2);
    if (lane == 0 && !(LK_X_P & 32)) {
        double E[9], Rn[9];
        exp3_1e5(d0, d1, d2, E);
        mat3_mul(sm.x, E, Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) sm.x[i] = Rn[i];
    }
    if (lane >= 3 && lane < 30) sm.x[6 + lane] += dxv;
    CORE_STAMP(12);
}

// The point update of the 256-thread kernels (lk_update_kernel, lk_update_snap_kernel, lk_small_bucket_kernel) through the one-wave core:
// the workgroup stages P and x in LDS, its wave 0 runs wave_update_core - the 6 x 37 system one column per lane, Gauss-Jordan in
// registers: six dependent steps instead of six times four workgroup barriers with single-thread pivot searches between them - and the
// workgroup writes the posterior back.  The same sums in the same order as dev_point_update (the two have always had to agree bit for
// bit: test_batch_replay_frozen_map), N == 1 handled inside the core.  The staging area is FilterSmem::A/B, idle during an update.
__device__ void dev_point_update_wave0(LkFilter* f, FilterSmem& sm, const double* tot, int N) {
    static_assert(sizeof(WaveSmem) <= sizeof(double) * 1800, "WaveSmem must fit FilterSmem::A + B");
    WaveSmem& w = *reinterpret_cast<WaveSmem*>(&sm.A[0]);
    const int tid = threadIdx.x;
    for (int i = tid; i < 900; i += LK_FB) w.P[i] = f->P[i];
    if (tid < 36) w.x[tid] = f->x[tid];
    __syncthreads();
    if (tid < LK_WAVE) wave_update_core<true>(w, tid < 32 ? tot[tid] : 0.0, N, tid);
    __syncthreads();
    for (int i = tid; i < 900; i += LK_FB) f->P[i] = w.P[i];
    if (tid < 36) f->x[tid] = w.x[tid];
    __syncthreads();
}

// updateByImu (eskf.cc:125-135) with the rows of KILO.cc:246-253 on LDS-resident, already propagated state: H selects
// (ba + imu_a) and (bw + imu_w), i.e. P H^T = P[:,9:15] + P[:,18:24].  Same one-column-per-lane Gauss-Jordan as
// wave_update_core (kept as a separate copy: that one is register-tuned for lk_update_wave_kernel's 96-VGPR budget);
// sums in the order lk_imu_kernel takes them.  acc / gyr / Rn6 are wave-uniform.
template <bool MW = false>
__device__ __forceinline__ void wave_imu_update_core(WaveSmem& sm, const double* acc, const double* gyr, double acc_scale,
                                                     const double* Rn6, int lane) {
    const double* x = sm.x;
    double z[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        z[i] = acc_scale * acc[i] - x[24 + i] - x[15 + i];  // (g/|a|) a - imu_a - ba
        z[3 + i] = gyr[i] - x[27 + i] - x[18 + i];          // w - imu_w - bw
    }
    auto pht = [&](int i, int m) { return sm.P[i * 30 + 9 + m] + sm.P[i * 30 + 18 + m]; };
    double col[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int c6 = lane < 6 ? lane : 0, j = (lane >= 6 && lane < 36) ? lane - 6 : 0;
        const double sv = (pht(9 + i, c6) + pht(18 + i, c6)) + ((i == c6) ? Rn6[i] : 0.0);
        const double gv = sm.P[(9 + i) * 30 + j] + sm.P[(18 + i) * 30 + j];
        col[i] = lane == 36 ? z[i] : (lane < 6 ? sv : gv);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(col[k]);
#pragma unroll
        for (int i = k + 1; i < 6; ++i) {
            double v = fabs(col[i]);
            if (v > best) best = v, p = i;
        }
        p = lane_bcast_u(p, k);
#pragma unroll
        for (int i = k + 1; i < 6; ++i)
            if (i == p) {
                double tmp = col[k];
                col[k] = col[i];
                col[i] = tmp;
            }
        // row factors S[i][k] / S[k][k]: lane i divides ONE of them (the column lives in lane k's registers), then they are
        // broadcast - one fp64 division per step on the critical path instead of five; the same quotients, the same bits
        double sk[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) sk[i] = lane_bcast_u(col[i], k);
        const double mine = lane == 0 ? sk[0] : lane == 1 ? sk[1] : lane == 2 ? sk[2] : lane == 3 ? sk[3] : lane == 4 ? sk[4] : sk[5];
        const double fac = mine / sk[k];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i == k) continue;
            const double fi = lane_bcast_u(fac, i);
            col[i] -= fi * col[k];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) col[i] = col[i] / lane_bcast_u(col[i], i);
    double dxv = 0.0;
    {
        const int i = lane < 30 ? lane : 29;
#pragma unroll
        for (int m = 0; m < 6; ++m) dxv += pht(i, m) * lane_bcast_u(col[m], 36);
        asm volatile("" : "+v"(dxv));
    }
    {
        const int jc = lane % 30, i0 = lane < 60 ? 15 * (lane / 30) : 15;
        double X[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) X[m] = __shfl(col[m], 6 + jc, LK_WAVE);
#pragma unroll 1
        for (int r0 = 0; r0 < 15; r0 += 5) {
            double nv[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const int i = i0 + r0 + r;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; ++m) s += pht(i, m) * X[m];
                nv[r] = sm.P[i * 30 + jc] - s;
            }
            core_sync<MW>();
            if (lane < 60) {
#pragma unroll
                for (int r = 0; r < 5; ++r) sm.P[(i0 + r0 + r) * 30 + jc] = nv[r];
            }
            core_sync<MW>();
        }
    }
    const double d0 = lane_bcast_u(dxv, 0), d1 = lane_bcast_u(dxv, 1), d2 = lane_bcast_u(dxv, 2);
    if (lane == 0) {
        double E[9], Rn[9];
        exp3_1e5(d0, d1, d2, E);
        mat3_mul(sm.x, E, Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) sm.x[i] = Rn[i];
    }
    if (lane >= 3 && lane < 30) sm.x[6 + lane] += dxv;
    core_sync<MW>();
}

// updateByKinImu (eskf.cc:137-145) with the rows of predictUpdateKinImu (KILO.cc:267-309) on LDS-resident, already propagated
// state, as ONE WAVE: the leg-fusion counterpart of wave_imu_update_core for the scan-resident replay kernel.  M = 6 + 3 c rows
// (c = feet in contact, <= 18).  H is never stored: rows 0..5 select (ba + imu_a) and (bw + imu_w); a contact's three rows are
// [ -R [w x p + v]x | 0 | I | ... | -R [p]x ] at columns 0..2 / 6..8 / 21..23, so H v for any column v is seven products.  P H^T
// (30 x M) goes to the wave's scratch (the residual rows' LDS region, idle during a message update); the augmented system
// [S | H P | z] is held one COLUMN per lane (lanes 0..17 S, 18..47 H P, 48 z) and swept by Gauss-Jordan with partial pivoting;
// the pivot column is published through LDS so that the M row factors are computed by M lanes at once (one fp64 divide per
// lane and step instead of M).  Every sum runs over the non-zero terms of the dense products of lk_kin_kernel /
// dev_dense_update in the same order (zero terms cannot change a finite sum), divisions are the same divisions: the two
// kernels agree to the last bit on finite data.  msg = one lk_kin_imu (33 doubles); Rn6 / kin_noise are wave-uniform.
struct KinScratch {
    double pht[30 * 18];
    double hb[4][18];    // per contact index: b0 = -R [w x p + v]x (9), b21 = -R [p]x (9), row-major
    double z[18], rd[18], fac[18], colk[18];
};
static_assert(sizeof(KinScratch) <= 64 * 15 * sizeof(double), "KinScratch must fit the residual rows' LDS region");

template <bool MW = false>
__device__ __forceinline__ void wave_kin_update_core(WaveSmem& sm, double* scratch, const double* __restrict__ msg, double acc_scale,
                                                     const double* Rn6, double kin_noise, int lane) {
    KinScratch& ks = *reinterpret_cast<KinScratch*>(scratch);
    const double* x = sm.x;
    const int* contact = reinterpret_cast<const int*>(msg + 25);
    int cmask = 0;
#pragma unroll
    for (int leg = 0; leg < 4; ++leg) cmask |= (contact[leg] != 0) ? (1 << leg) : 0;
    const int M = 6 + 3 * __popc(cmask);
    if (lane < 6) {   // IMU rows, KILO.cc:281-286
        const int i = lane < 3 ? lane : lane - 3;
        ks.z[lane] = lane < 3 ? acc_scale * msg[27 + i] - x[24 + i] - x[15 + i] : msg[30 + i] - x[27 + i] - x[18 + i];
        ks.rd[lane] = Rn6[lane];
    }
    if (lane < 4 && ((cmask >> lane) & 1)) {   // one lane per foot in contact, KILO.cc:290-309
        const int leg = lane, idx = __popc(cmask & ((1 << leg) - 1)), r0 = 6 + 3 * idx;
        double Wk[9], mRot[9], K1[9], K2[9], b0[9], b21[9];
        skew3(V3{x[27], x[28], x[29]}, Wk);
#pragma unroll
        for (int i = 0; i < 9; ++i) mRot[i] = -x[i];
        const V3 fp = V3{msg[1 + 3 * leg], msg[2 + 3 * leg], msg[3 + 3 * leg]};
        const V3 fv = V3{msg[13 + 3 * leg], msg[14 + 3 * leg], msg[15 + 3 * leg]};
        const V3 wp = mat3_mul_v(Wk, fp);
        const V3 wpv = V3{wp.x + fv.x, wp.y + fv.y, wp.z + fv.z};
        skew3(wpv, K1);
        skew3(fp, K2);
        mat3_mul(mRot, K1, b0);
        mat3_mul(mRot, K2, b21);
#pragma unroll
        for (int i = 0; i < 9; ++i) ks.hb[idx][i] = b0[i], ks.hb[idx][9 + i] = b21[i];
        const V3 Rw = mat3_mul_v(x, wpv);
        ks.z[r0 + 0] = -x[12] - Rw.x, ks.z[r0 + 1] = -x[13] - Rw.y, ks.z[r0 + 2] = -x[14] - Rw.z;
        ks.rd[r0 + 0] = kin_noise, ks.rd[r0 + 1] = kin_noise, ks.rd[r0 + 2] = kin_noise;
    }
    core_sync<MW>();
    // -- P H^T, entry (i, m): dot of row i of P with row m of H over H's non-zero columns, ascending
    for (int e = lane; e < 30 * 18; e += LK_WAVE) {
        const int i = e / 18, m = e % 18;
        if (m >= M) continue;
        const double* Pi = &sm.P[i * 30];
        double s;
        if (m < 6) {
            s = Pi[9 + m];
            s += Pi[18 + m];
        } else {
            const int k = (m - 6) / 3, r = (m - 6) % 3;
            const double* b0 = &ks.hb[k][3 * r];
            const double* b21 = &ks.hb[k][9 + 3 * r];
            s = Pi[0] * b0[0];
            s += Pi[1] * b0[1];
            s += Pi[2] * b0[2];
            s += Pi[6 + r];
            s += Pi[21] * b21[0];
            s += Pi[22] * b21[1];
            s += Pi[23] * b21[2];
        }
        ks.pht[i * 18 + m] = s;
    }
    core_sync<MW>();
    // -- this lane's column of [S | H P | z]
    double col[18];
    {
        const bool isS = lane < 18, isG = lane >= 18 && lane < 48;
        const int c = isS ? (lane < M ? lane : 0) : 0, j = isG ? lane - 18 : 0;
        auto v = [&](int row) { return isS ? ks.pht[row * 18 + c] : sm.P[row * 30 + j]; };
        const double v0 = v(0), v1 = v(1), v2 = v(2), v6 = v(6), v7 = v(7), v8 = v(8), v21 = v(21), v22 = v(22), v23 = v(23);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double s = v(9 + a);
            s += v(18 + a);
            if (isS && a == lane) s = s + ks.rd[a];
            col[a] = lane == 48 ? ks.z[a] : s;
        }
#pragma unroll
        for (int a = 6; a < 18; ++a) {
            col[a] = 0.0;
            if (a < M) {
                const int k = (a - 6) / 3, r = (a - 6) % 3;
                const double* b0 = &ks.hb[k][3 * r];
                const double* b21 = &ks.hb[k][9 + 3 * r];
                const double v6r = r == 0 ? v6 : (r == 1 ? v7 : v8);
                double s = b0[0] * v0;
                s += b0[1] * v1;
                s += b0[2] * v2;
                s += v6r;
                s += b21[0] * v21;
                s += b21[1] * v22;
                s += b21[2] * v23;
                if (isS && a == lane) s = s + ks.rd[a];
                col[a] = lane == 48 ? ks.z[a] : s;
            }
        }
        if ((isS && lane >= M) || lane > 48) {
#pragma unroll
            for (int a = 0; a < 18; ++a) col[a] = 0.0;
        }
    }
    // -- Gauss-Jordan with partial pivoting (dev_solve), one column per lane, row factors through LDS
#pragma unroll
    for (int k = 0; k < 18; ++k) {
        if (k < M) {
            if (lane == k) {
#pragma unroll
                for (int i = 0; i < 18; ++i) ks.colk[i] = col[i];
            }
            core_sync<MW>();
            int p = k;
            double best = fabs(ks.colk[k]);
#pragma unroll
            for (int i = k + 1; i < 18; ++i) {
                if (i < M) {
                    const double vv = fabs(ks.colk[i]);
                    if (vv > best) best = vv, p = i;
                }
            }
#pragma unroll
            for (int i = k + 1; i < 18; ++i)
                if (i == p) {
                    const double tmp = col[k];
                    col[k] = col[i];
                    col[i] = tmp;
                }
            if (lane < M) {   // row `lane` after the swap: rows k and p of the published column trade places
                const int src = lane == k ? p : (lane == p ? k : lane);
                ks.fac[lane] = lane == k ? 0.0 : ks.colk[src] / ks.colk[p];
            }
            core_sync<MW>();
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                if (i != k && i < M) col[i] -= ks.fac[i] * col[k];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 18; ++i)
        if (i < M) col[i] = col[i] / lane_bcast_u(col[i], i);   // X = G / diag(S)
    // -- dx = P H^T X[:,30]
    double dxv = 0.0;
    {
        const int i = lane < 30 ? lane : 29;
#pragma unroll
        for (int m = 0; m < 18; ++m) {
            const double xm = lane_bcast_u(col[m], 48);
            if (m < M) dxv += ks.pht[i * 18 + m] * xm;
        }
    }
    // -- P -= P H^T X[:,0:30]: lane -> column lane % 30, rows 15 * (lane / 30) ...; an entry is read and written by one lane only
    {
        const int jc = lane % 30, i0 = lane < 60 ? 15 * (lane / 30) : 15;
        double X[18];
#pragma unroll
        for (int m = 0; m < 18; ++m) X[m] = __shfl(col[m], 18 + jc, LK_WAVE);
        if (lane < 60) {
#pragma unroll 1
            for (int r = 0; r < 15; ++r) {
                const int i = i0 + r;
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 18; ++m)
                    if (m < M) s += ks.pht[i * 18 + m] * X[m];
                sm.P[i * 30 + jc] = sm.P[i * 30 + jc] - s;
            }
        }
    }
    core_sync<MW>();
    const double d0 = lane_bcast_u(dxv, 0), d1 = lane_bcast_u(dxv, 1), d2 = lane_bcast_u(dxv, 2);
    if (lane == 0) {
        double E[9], Rn[9];
        exp3_1e5(d0, d1, d2, E);
        mat3_mul(sm.x, E, Rn);
#pragma unroll
        for (int i = 0; i < 9; ++i) sm.x[i] = Rn[i];
    }
    if (lane >= 3 && lane < 30) sm.x[6 + lane] += dxv;
    core_sync<MW>();
}

// ESKF::predict(dt_cov, false, true) then predict(dt, true, false) (KILO.cc:111-115) on LDS-resident state.
template <bool MW = false>
__device__ __forceinline__ void wave_predict_core(WaveSmem& sm, const double* __restrict__ Q, double dt_cov, double dt, int lane,
                                                  const bool q_diag = false) {
    // The two rotations of a predict - Exp(-dt_cov w) for Fx (getFx, eskf.cc:74) and Exp(dt w) for the state (operator+=,
    // eskf.cc:19) - depend on the OLD state only: lanes 0 and 1 evaluate one each in the same instruction stream (a wave
    // executes both sides of a lane-0 / lane-1 branch one after the other; sqrt, three divisions and a sin / cos pair are a
    // ~4 k-cycle dependent chain each).  Lane 1 keeps the propagated rotation / position / velocity in registers until the
    // covariance product has read the old rotation out of sm.x.
    double Rn[9], dpv[6];
    CORE_T0;
    CORE_COUNT(7);
    // the diagonal of Q is the only global-memory operand of a predict: requested here, used behind the covariance products (a load issued
    // where it is used is a trip to the L2 on a chain that has nothing else to do)
    double qd = 0.0;
    if (q_diag && lane < 30) qd = Q[lane * 31];
    if (lane < 2 && !(LK_X_P & 1)) {
        const double* x = sm.x;
        const double sc = lane == 0 ? -dt_cov : dt;
        const double thr = lane == 0 ? 0.0000001 : 0.00001;   // math_utils.hpp:19-32 / :54-68
        double E[9];
        exp_so3_thr(sc * x[27], sc * x[28], sc * x[29], thr, E);
        if (lane == 0) {  // getFx, eskf.cc:72-81
            V3 a = V3{x[24], x[25], x[26]};
            double K[9], mR[9], B60[9];
            skew3(a, K);
            for (int i = 0; i < 9; ++i) mR[i] = (-dt_cov) * x[i];
            mat3_mul(mR, K, B60);
            for (int i = 0; i < 9; ++i) sm.fx[i] = E[i], sm.fx[9 + i] = B60[i];
        } else {          // getFunctionf + operator+=, eskf.cc:64-70,18-29 (state_boxplus with d[9..29] = 0)
            V3 Ra = mat3_mul_v(x, V3{x[24], x[25], x[26]});
            for (int i = 0; i < 3; ++i) dpv[i] = dt * x[12 + i];
            dpv[3] = dt * (Ra.x + x[21]), dpv[4] = dt * (Ra.y + x[22]), dpv[5] = dt * (Ra.z + x[23]);
            mat3_mul(x, E, Rn);
        }
    }
    core_sync<MW>();
    CORE_STAMP(0);
    const double* E = sm.fx;
    const double* B60 = sm.fx + 9;
    // Fx differs from I in three row blocks of different shape (eskf.cc:72-81): rows 0..2 = [E | dt I at col 21], rows 3..5 =
    // [I | dt I at col 6], rows 6..8 = [B60 | I | dt I at col 15 | dt R at col 18].  Each block is handled by ALL lanes with one
    // straight-line expression (90 entries = two rounds of 64 lanes) instead of one loop whose lanes fall into different blocks
    // and execute all three shapes in turn; every entry is the same sum in the same order as before (bit-identical).
    if (!(LK_X_P & 2)) {  // rows 0..8 of B = Fx * P, in place (rows >= 9 of B are rows of P)
        double nb[6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q, i = e < 90 ? e / 30 : 0, c = e % 30;
            {
                double s = 0.0;
                s += E[3 * i + 0] * sm.P[0 * 30 + c];
                s += E[3 * i + 1] * sm.P[1 * 30 + c];
                s += E[3 * i + 2] * sm.P[2 * 30 + c];
                s += dt_cov * sm.P[(21 + i) * 30 + c];
                nb[q] = s;
            }
            {
                double s = 0.0;
                s += 1.0 * sm.P[(3 + i) * 30 + c];
                s += dt_cov * sm.P[(6 + i) * 30 + c];
                nb[2 + q] = s;
            }
            {
                double s = 0.0;
                s += B60[3 * i + 0] * sm.P[0 * 30 + c];
                s += B60[3 * i + 1] * sm.P[1 * 30 + c];
                s += B60[3 * i + 2] * sm.P[2 * 30 + c];
                s += 1.0 * sm.P[(6 + i) * 30 + c];
                s += dt_cov * sm.P[(15 + i) * 30 + c];
                s += (dt_cov * sm.x[3 * i + 0]) * sm.P[18 * 30 + c];
                s += (dt_cov * sm.x[3 * i + 1]) * sm.P[19 * 30 + c];
                s += (dt_cov * sm.x[3 * i + 2]) * sm.P[20 * 30 + c];
                nb[4 + q] = s;
            }
        }
        core_sync<MW>();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            if (e < 90) sm.P[e] = nb[q], sm.P[90 + e] = nb[2 + q], sm.P[180 + e] = nb[4 + q];
        }
    }
    core_sync<MW>();
    CORE_STAMP(1);
    if (!(LK_X_P & 2)) {  // columns 0..8 of B * Fx^T, in place: the same three shapes, 30 rows x 3 columns each
        double nc[6];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q, i = e < 90 ? e / 3 : 0, c = e % 3;
            const double* Bi = &sm.P[i * 30];
            {
                double s = 0.0;
                s += Bi[0] * E[3 * c + 0];
                s += Bi[1] * E[3 * c + 1];
                s += Bi[2] * E[3 * c + 2];
                s += Bi[21 + c] * dt_cov;
                nc[q] = s;
            }
            {
                double s = 0.0;
                s += Bi[3 + c] * 1.0;
                s += Bi[6 + c] * dt_cov;
                nc[2 + q] = s;
            }
            {
                double s = 0.0;
                s += Bi[0] * B60[3 * c + 0];
                s += Bi[1] * B60[3 * c + 1];
                s += Bi[2] * B60[3 * c + 2];
                s += Bi[6 + c] * 1.0;
                s += Bi[15 + c] * dt_cov;
                s += Bi[18] * (dt_cov * sm.x[3 * c + 0]);
                s += Bi[19] * (dt_cov * sm.x[3 * c + 1]);
                s += Bi[20] * (dt_cov * sm.x[3 * c + 2]);
                nc[4 + q] = s;
            }
        }
        core_sync<MW>();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            if (e < 90) {
                const int i = e / 3, c = e % 3;
                sm.P[i * 30 + c] = nc[q], sm.P[i * 30 + 3 + c] = nc[2 + q], sm.P[i * 30 + 6 + c] = nc[4 + q];
            }
        }
    }
    core_sync<MW>();
    CORE_STAMP(2);
    const double dt2 = dt_cov * dt_cov;
    if (LK_X_P & 4) {
    } else if (q_diag) {   // Q of initProcessCovQ (eskf.cc:47-62) is diagonal: the other 870 terms are + dt^2 * 0
        if (lane < 30) sm.P[lane * 31] = sm.P[lane * 31] + dt2 * qd;
    } else {
        for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = sm.P[e] + dt2 * Q[e];
    }
    if (lane == 1 && !(LK_X_P & 1)) {
        double* x = sm.x;
        for (int i = 0; i < 9; ++i) x[i] = Rn[i];
        for (int i = 0; i < 6; ++i) x[9 + i] += dpv[i];
    }
    core_sync<MW>();
    CORE_STAMP(3);
}

__device__ __forceinline__ void dev_update_wave(LkFilter* f, WaveSmem& sm, const double* __restrict__ part, int nblk, double t,
                                                const double* __restrict__ Q, double t_next, int mode) {
    const int lane = threadIdx.x;
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t;
    const double t_pred = f->last_predict_t;
    bool dirty = false;  // sm.P / sm.x differ from f->P / f->x

    if (mode & 1) {
        // -- totals: lane (j, hh) carries the groups g = hh, hh+2, hh+4, hh+6 of lk_update_kernel, four accumulators each
        const int j = lane & 31, hh = lane >> 5;
        double acc[4][4];
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[gi][q] = 0.0;
        for (int m = 0; m * 32 < nblk; ++m) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                int b = hh + 2 * gi + 32 * m;
                if (b + 24 < nblk) {
                    acc[gi][0] += part[(size_t)b * LK_NPART + j];
                    acc[gi][1] += part[(size_t)(b + 8) * LK_NPART + j];
                    acc[gi][2] += part[(size_t)(b + 16) * LK_NPART + j];
                    acc[gi][3] += part[(size_t)(b + 24) * LK_NPART + j];
                } else {
                    for (; b < nblk; b += 8) acc[gi][0] += part[(size_t)b * LK_NPART + j];
                }
            }
        }
        double totv = 0.0;  // tot[j] in lanes 0..31
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            double mine = (acc[gi][0] + acc[gi][1]) + (acc[gi][2] + acc[gi][3]);
            double other = __shfl(mine, (lane + 32) & 63, LK_WAVE);
            totv += mine;   // group 2*gi   (meaningful in lanes 0..31)
            totv += other;  // group 2*gi+1
        }
        const int N = (int)(lane_bcast_u(totv, 28) + 0.5);
        if (lane == 0) {
            f->n_buckets += 1;
            f->last_N = N;
            f->updated = N > 0;
            if (N > 0) {
                f->n_updates += 1;
                f->n_effect += (unsigned long long)N;
                f->last_update_t = t;  // KILO.cc:212
            }
        }
        if (N > 0) {
            t_upd = t;
            dirty = true;
            wave_update_core(sm, totv, N, lane);
        }
    }
    __syncthreads();

    if (mode & 2) {
        const double dt_cov = t_next - t_upd;
        const double dt = t_next - t_pred;
        dirty = true;
        wave_predict_core(sm, Q, dt_cov, dt, lane);
        if (lane == 0) f->last_predict_t = t_next;
    }
    if (dirty) {
        for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
        if (lane < 36) f->x[lane] = sm.x[lane];
    }
}

// Register budget of the one-wave update / predict kernels of the batch entries, in waves per SIMD.  A batch launches one wave per scan - 1 024 waves on 1 024
// SIMDs - so occupancy buys nothing here, and at the residual kernel's five waves (96 registers) the filter cores spilled 76-100 B per lane.
#ifndef LK_UPD_WAVES
#define LK_UPD_WAVES 2
#endif
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_WAVE, LK_UPD_WAVES)
    lk_update_wave_kernel(LkFilter* filters, const double* __restrict__ partials, int nblk, size_t slot_stride, double t,
                          const double* __restrict__ Q, double t_next, int mode);
#else
__global__ void __launch_bounds__(LK_WAVE, LK_UPD_WAVES)
    lk_update_wave_kernel(LkFilter* filters, const double* __restrict__ partials, int nblk, size_t slot_stride, double t,
                          const double* __restrict__ Q, double t_next, int mode) {
    __shared__ WaveSmem sm;
    dev_update_wave(&filters[blockIdx.x], sm, partials + (size_t)blockIdx.x * slot_stride, nblk, t, Q, t_next, mode);
}
#endif

// Ragged batch: bucket b of every scan that has one (b == -1: the predict to each scan's first bucket); times, bucket
// sizes and "is there a next bucket" come from the scan's own tables.
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_WAVE, LK_UPD_WAVES)
    lk_update_wave_ragged_kernel(LkFilter* filters, const double* __restrict__ partials, size_t slot_stride,
                                 const double* __restrict__ Q, LkRagged rg, int b, int update_only = 0);
#else
__global__ void __launch_bounds__(LK_WAVE, LK_UPD_WAVES)
    lk_update_wave_ragged_kernel(LkFilter* filters, const double* __restrict__ partials, size_t slot_stride,
                                 const double* __restrict__ Q, LkRagged rg, int b, int update_only = 0) {
    __shared__ WaveSmem sm;
    const int slot = blockIdx.x;
    const int nbk = rag_nb(rg, slot);
    if (b >= nbk) return;
    if (update_only) {   // the batch WITH insert: the posterior of bucket b first (the insert reads it), the way to bucket b + 1 is lk_rag_advance_kernel's
        const unsigned long long* po1 = rag_pt_off(rg, slot);
        dev_update_wave(&filters[slot], sm, partials + (size_t)slot * slot_stride, ((int)(po1[b + 1] - po1[b]) + LK_WAVE - 1) / LK_WAVE, rag_t(rg, slot)[b], Q, 0.0, 1);
        return;
    }
    const double* T = rag_t(rg, slot);
    const double* part = partials + (size_t)slot * slot_stride;
    if (b < 0) {
        dev_update_wave(&filters[slot], sm, part, 0, 0.0, Q, T[0], 2);
        return;
    }
    const unsigned long long* po = rag_pt_off(rg, slot);
    const int n = (int)(po[b + 1] - po[b]);
    const bool has_next = b + 1 < nbk;
    dev_update_wave(&filters[slot], sm, part, (n + LK_WAVE - 1) / LK_WAVE, T[b], Q, has_next ? T[b + 1] : 0.0, has_next ? 3 : 1);
}
#endif

#ifdef LK_KERNELS_ELSEWHERE
__global__ void lk_set_times_ragged_kernel(LkFilter* filters, int n, const double* __restrict__ t_begin);
#else
__global__ void lk_set_times_ragged_kernel(LkFilter* filters, int n, const double* __restrict__ t_begin) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) filters[s].last_predict_t = t_begin[s], filters[s].last_update_t = t_begin[s];
}
#endif

// updateByPoints(ObsShared&) on caller rows (class-surface call): one block accumulates A, b
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB)
    lk_obs_points_kernel(LkFilter* filters, int slot, const double* __restrict__ h6, const double* __restrict__ z,
                         const double* __restrict__ R, int N);
#else
__global__ void __launch_bounds__(LK_FB)
    lk_obs_points_kernel(LkFilter* filters, int slot, const double* __restrict__ h6, const double* __restrict__ z,
                         const double* __restrict__ R, int N) {
    __shared__ FilterSmem sm;
    __shared__ double red[LK_FB / LK_WAVE][LK_NPART];
    __shared__ double tot[LK_NPART];
    const int tid = threadIdx.x;
    double acc[27];
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    for (int k = tid; k < N; k += LK_FB) {
        double h[6];
        for (int c = 0; c < 6; ++c) h[c] = h6[(size_t)k * 6 + c];
        double r = R[k];
        if (N == 1) r = r + 0.0001;
        double ri = 1.0 / r;
        int q = 0;
        for (int i = 0; i < 6; ++i) {
            double hi = h[i] * ri;
            for (int j = i; j < 6; ++j) acc[q++] += hi * h[j];
            acc[21 + i] += hi * z[k];
        }
    }
    for (int i = 0; i < 27; ++i) {
        double v = wave_sum(acc[i]);
        if ((tid & 63) == 0) red[tid >> 6][i] = v;
    }
    __syncthreads();
    if (tid < 27) {
        double s = 0.0;
        for (int w = 0; w < LK_FB / LK_WAVE; ++w) s += red[w][tid];
        tot[tid] = s;
    }
    __syncthreads();
    if (N > 0) dev_point_update(&filters[slot], sm, &tot[0], &tot[21]);
}
#endif

// ---- IMU rows: z and R, KILO.cc:246-253 (thread 0), H = I on cols 9..14 and 18..23
__device__ void dev_imu_rows(const LkFilter* f, const double* acc, const double* gyr, double acc_scale,
                             const double* Rn6, FilterSmem& sm) {
    const double* x = f->x;
    for (int i = 0; i < 3; ++i) {
        sm.vec[32 + i] = acc_scale * acc[i] - x[24 + i] - x[15 + i];      // (g/|a|) a - imu_a - ba
        sm.vec[32 + 3 + i] = gyr[i] - x[27 + i] - x[18 + i];              // w - imu_w - bw
    }
    for (int i = 0; i < 6; ++i) sm.vec[40 + i] = Rn6[i];
}

struct LkImuArgs {
    double t;
    double acc[3], gyr[3];
    double acc_scale;  // gravity_ / acc_norm_
    double Rn[6];      // acc, acc, acc_z, gyr, gyr, gyr meas noise
};

// predictUpdateImu, KILO.cc:235-258 + updateByImu, eskf.cc:125-135
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB) lk_imu_kernel(LkFilter* filters, const double* __restrict__ Q, LkImuArgs a);
#else
__global__ void __launch_bounds__(LK_FB) lk_imu_kernel(LkFilter* filters, const double* __restrict__ Q, LkImuArgs a) {
    __shared__ FilterSmem sm;
    LkFilter* f = &filters[blockIdx.x];
    const int tid = threadIdx.x;
    dev_predict(f, Q, a.t, sm);  // leaves sm.P = propagated covariance
    if (tid == 0) dev_imu_rows(f, a.acc, a.gyr, a.acc_scale, a.Rn, sm);
    for (int e = tid; e < 30 * 6; e += LK_FB) {
        int i = e / 6, k = e % 6;
        sm.PHT[i * 18 + k] = sm.P[i * 30 + 9 + k] + sm.P[i * 30 + 18 + k];
    }
    for (int e = tid; e < 6 * 30; e += LK_FB) {
        int k = e / 30, j = e % 30;
        sm.G[k * 31 + j] = sm.P[(9 + k) * 30 + j] + sm.P[(18 + k) * 30 + j];  // HP
    }
    __syncthreads();
    if (tid < 36) {
        int i = tid / 6, j = tid % 6;
        sm.S[i * 18 + j] = sm.PHT[(9 + i) * 18 + j] + sm.PHT[(18 + i) * 18 + j] + ((i == j) ? sm.vec[40 + i] : 0.0);
    }
    if (tid < 6) sm.G[tid * 31 + 30] = sm.vec[32 + tid];
    __syncthreads();
    dev_solve(sm, 6, 31);
    dev_kalman_apply(f, sm, 6);
    if (tid == 0) f->last_update_t = a.t;  // KILO.cc:256
}
#endif

// updateByImu(ObsShared&) on caller rows, no predict (class-surface call)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB)
    lk_obs_imu_kernel(LkFilter* filters, int slot, const double* __restrict__ z6, const double* __restrict__ R6);
#else
__global__ void __launch_bounds__(LK_FB)
    lk_obs_imu_kernel(LkFilter* filters, int slot, const double* __restrict__ z6, const double* __restrict__ R6) {
    __shared__ FilterSmem sm;
    LkFilter* f = &filters[slot];
    const int tid = threadIdx.x;
    for (int i = tid; i < 900; i += LK_FB) sm.P[i] = f->P[i];
    __syncthreads();
    for (int e = tid; e < 30 * 6; e += LK_FB) {
        int i = e / 6, k = e % 6;
        sm.PHT[i * 18 + k] = sm.P[i * 30 + 9 + k] + sm.P[i * 30 + 18 + k];
    }
    for (int e = tid; e < 6 * 30; e += LK_FB) {
        int k = e / 30, j = e % 30;
        sm.G[k * 31 + j] = sm.P[(9 + k) * 30 + j] + sm.P[(18 + k) * 30 + j];
    }
    __syncthreads();
    if (tid < 36) {
        int i = tid / 6, j = tid % 6;
        sm.S[i * 18 + j] = sm.PHT[(9 + i) * 18 + j] + sm.PHT[(18 + i) * 18 + j] + ((i == j) ? R6[i] : 0.0);
    }
    if (tid < 6) sm.G[tid * 31 + 30] = z6[tid];
    __syncthreads();
    dev_solve(sm, 6, 31);
    dev_kalman_apply(f, sm, 6);
}
#endif

// dense-H update shared by the kin path: sm.H (M x 30), sm.vec[32..] = z, sm.fac-free R in sm.B[0..M)
__device__ void dev_dense_update(LkFilter* f, FilterSmem& sm, int M) {
    const int tid = threadIdx.x;
    for (int e = tid; e < 30 * M; e += LK_FB) {  // PHT = P H^T
        int i = e / M, m = e % M;
        double s = 0.0;
        for (int j = 0; j < 30; ++j) s += sm.P[i * 30 + j] * sm.H[m * 30 + j];
        sm.PHT[i * 18 + m] = s;
    }
    for (int e = tid; e < M * 30; e += LK_FB) {  // G[:,0:30] = H P
        int m = e / 30, j = e % 30;
        double s = 0.0;
        for (int i = 0; i < 30; ++i) s += sm.H[m * 30 + i] * sm.P[i * 30 + j];
        sm.G[m * 31 + j] = s;
    }
    if (tid < M) sm.G[tid * 31 + 30] = sm.vec[32 + tid];
    __syncthreads();
    for (int e = tid; e < M * M; e += LK_FB) {  // S = H PHT + diag(R)
        int a = e / M, b = e % M;
        double s = 0.0;
        for (int j = 0; j < 30; ++j) s += sm.H[a * 30 + j] * sm.PHT[j * 18 + b];
        sm.S[a * 18 + b] = s + ((a == b) ? sm.B[a] : 0.0);
    }
    __syncthreads();
    dev_solve(sm, M, 31);
    dev_kalman_apply(f, sm, M);
}

struct LkKinArgs {
    lk_kin_imu k;
    double acc_scale;
    double Rn[6];
    double kin_noise;
};

// predictUpdateKinImu, KILO.cc:260-314 + updateByKinImu, eskf.cc:137-145
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB) lk_kin_kernel(LkFilter* filters, const double* __restrict__ Q, LkKinArgs a);
#else
__global__ void __launch_bounds__(LK_FB) lk_kin_kernel(LkFilter* filters, const double* __restrict__ Q, LkKinArgs a) {
    __shared__ FilterSmem sm;
    __shared__ int sM;
    LkFilter* f = &filters[blockIdx.x];
    const int tid = threadIdx.x;
    dev_predict(f, Q, a.k.time_stamp, sm);
    for (int i = tid; i < 18 * 30; i += LK_FB) sm.H[i] = 0.0;
    __syncthreads();
    if (tid == 0) {
        const double* x = f->x;
        dev_imu_rows(f, a.k.acc, a.k.gyr, a.acc_scale, a.Rn, sm);
        for (int i = 0; i < 6; ++i) {
            sm.H[i * 30 + 9 + i] = 1.0;
            sm.H[i * 30 + 18 + i] = 1.0;
            sm.B[i] = a.Rn[i];
        }
        int idx = 0;
        double Wk[9], mRot[9];
        skew3(V3{x[27], x[28], x[29]}, Wk);
        for (int i = 0; i < 9; ++i) mRot[i] = -x[i];
        for (int leg = 0; leg < 4; ++leg) {
            if (!a.k.contact[leg]) continue;
            V3 fp = V3{a.k.foot_pos[leg][0], a.k.foot_pos[leg][1], a.k.foot_pos[leg][2]};
            V3 fv = V3{a.k.foot_vel[leg][0], a.k.foot_vel[leg][1], a.k.foot_vel[leg][2]};
            V3 wp = mat3_mul_v(Wk, fp);
            V3 wpv = V3{wp.x + fv.x, wp.y + fv.y, wp.z + fv.z};
            double K1[9], K2[9], b0[9], b21[9];
            skew3(wpv, K1);
            skew3(fp, K2);
            mat3_mul(mRot, K1, b0);
            mat3_mul(mRot, K2, b21);
            int r0 = 6 + 3 * idx;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    sm.H[(r0 + r) * 30 + 0 + c] = b0[3 * r + c];
                    sm.H[(r0 + r) * 30 + 6 + c] = (r == c) ? 1.0 : 0.0;
                    sm.H[(r0 + r) * 30 + 21 + c] = b21[3 * r + c];
                }
            V3 Rw = mat3_mul_v(x, wpv);
            sm.vec[32 + r0 + 0] = -x[12] - Rw.x;
            sm.vec[32 + r0 + 1] = -x[13] - Rw.y;
            sm.vec[32 + r0 + 2] = -x[14] - Rw.z;
            for (int r = 0; r < 3; ++r) sm.B[r0 + r] = a.kin_noise;
            idx++;
        }
        sM = 6 + 3 * idx;
    }
    __syncthreads();
    dev_dense_update(f, sm, sM);
    if (tid == 0) f->last_update_t = a.k.time_stamp;  // KILO.cc:312
}
#endif

// updateByKinImu(ObsShared&) on caller rows, no predict (class-surface call)
#ifdef LK_KERNELS_ELSEWHERE
__global__ void __launch_bounds__(LK_FB)
    lk_obs_kin_kernel(LkFilter* filters, int slot, const double* __restrict__ ki_h, const double* __restrict__ ki_z,
                      const double* __restrict__ ki_R, int M);
#else
__global__ void __launch_bounds__(LK_FB)
    lk_obs_kin_kernel(LkFilter* filters, int slot, const double* __restrict__ ki_h, const double* __restrict__ ki_z,
                      const double* __restrict__ ki_R, int M) {
    __shared__ FilterSmem sm;
    LkFilter* f = &filters[slot];
    const int tid = threadIdx.x;
    for (int i = tid; i < 900; i += LK_FB) sm.P[i] = f->P[i];
    for (int i = tid; i < M * 30; i += LK_FB) sm.H[i] = ki_h[i];
    if (tid < M) {
        sm.vec[32 + tid] = ki_z[tid];
        sm.B[tid] = ki_R[tid];
    }
    __syncthreads();
    dev_dense_update(f, sm, M);
}
#endif
