// lk_stream.hip - the stream translation unit of liblegkilo_hip.so (see lk_internal.h): one live scan after the other WITH the map insert - the per-bucket
// launches, the scan-resident and grid-resident stream kernels, the pipelined path - and the KILO-path entry points that run them (lk_update_points,
// lk_process_scan(_dev), lk_process_raw_scan, ...).
#define LK_TU_STREAM 1
#include "lk_internal.h"

extern "C" {
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
// The message updates of the scan-resident stream kernel as CALLS (-DLK_MSG_CALL=1; A/B): a kinematic + IMU message is an 18-column Gauss-Jordan in registers,
// run a few dozen times per scan; inlined, its register demand is the whole kernel's (256 + 240 B of scratch for MSG == 2 against 96 B without messages) and the
// per-bucket code pays the spills.
#ifndef LK_MSG_CALL
#define LK_MSG_CALL 0
#endif
// -DLK_PT_PREFETCH=1 (A/B, measured: no gain): the filter wave asks for the next bucket's scan points as soon as the current bucket's tiles are done
#ifndef LK_PT_PREFETCH
#define LK_PT_PREFETCH 0
#endif
#if LK_MSG_CALL
__device__ __attribute__((noinline)) void stream_kin_update_call(WaveSmem& sm, double* scratch, const double* msg, double acc_scale, const double* Rn6, double kin_noise, int lane) {
    wave_kin_update_core<true>(sm, scratch, msg, acc_scale, Rn6, kin_noise, lane);
}
__device__ __attribute__((noinline)) void stream_imu_update_call(WaveSmem& sm, const double* acc, const double* gyr, double acc_scale, const double* Rn6, int lane) {
    wave_imu_update_core<true>(sm, acc, gyr, acc_scale, Rn6, lane);
}
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
#if LK_PT_PREFETCH
    const unsigned long long pts_end = po[nbk];   // one past the scan's last point
    float4 pf = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
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
#if LK_MSG_CALL
            if (MSG == 2)
                stream_kin_update_call(sm, rows, mm, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else
                stream_imu_update_call(sm, mm + 1, mm + 4, rg.acc_scale, rg.Rn, lane);
#else
            if (MSG == 2)
                wave_kin_update_core<true>(sm, rows, mm, rg.acc_scale, rg.Rn, rg.kin_noise, lane);
            else
                wave_imu_update_core<true>(sm, mm + 1, mm + 4, rg.acc_scale, rg.Rn, lane);
#endif
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
#if LK_PT_PREFETCH
        // the NEXT bucket's points, requested now: their trip runs beside this bucket's wait, update and predict instead of at the head of the next tile (the
        // value is only kept until the update is done - the tile's own load then finds the lines in the CU's L1)
        {
            const unsigned long long q = base + (unsigned long long)n + (unsigned long long)lane;
            if (q < pts_end) pf = reinterpret_cast<const float4*>(pts)[q];
        }
#endif
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
#if LK_PT_PREFETCH
        asm volatile("" ::"v"(pf.x), "v"(pf.y), "v"(pf.z), "v"(pf.w));   // the request has to exist; nothing reads the value
#endif
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

}  // extern "C"

extern "C" {
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

}  // extern "C"

extern "C" {
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

int run_scan(lk_handle* h, const lk_point* pts, const lk_point* d_pts, size_t n, double t_begin, const lk_imu* imus,
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

}  // extern "C"

