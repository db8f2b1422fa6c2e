// lk_ovscan.hip - the scan-resident kernel of the recorded-run batch replay WITH insert (lk_batch_replay_overlay_ragged_dev, small buckets), a translation
// unit of its own: its two instantiations take as long to compile as the rest of the overlay unit (see lk_internal.h; the host loop around it -
// rounds, fallback launches - is overlay_ragged_launch in lk_overlay.hip).
#define LK_TU_OVSCAN 1
#include "lk_internal.h"

extern "C++" {
// Round 6: a recorded scan's WHOLE bucket chain with its map insert in one launch, one wave per scan.  Launch by launch the batch costs, per
// bucket index, the SLOWEST slot's chain of every kernel plus a launch boundary each (profiles/r06_ragged_overlay_pmc.json: 87 us per index over seven launches)
// - whatever the slot at hand has to do, which is usually little: a dozen points, a plane fit in every fourth bucket.  Here a scan runs its own buckets back to
// back: the front of lk_rag_ov_front_kernel (messages, predict, residual tiles, update - state and covariance stay in LDS from bucket to bucket; the posterior
// still goes to the filter record, which the insert reads), then the bodies of lk_ov_mid_kernel (copy-on-write, point geometry, lane-per-root pass) and of
// lk_ov_tail_kernel (generic root pass, plane fits, apply).  What the launch cannot hold is the fallback code (256 registers + 6.4 KB of scratch per lane,
// lk_ov_insert_fallback_kernel): a scan whose bucket leaves fallback items stops behind that bucket (cur[slot] = the next one, fb_b[slot] = this one), the host
// runs the fallback launch for the stopped scans and launches again - LkResume's protocol of the stream path.  Same device functions in the same order per
// scan: bit-identical to the launch-by-launch form (test_batch_replay_overlay_ragged; LEGKILO_RAG_RESIDENT=0 is the A/B).
// Between two phases of the scan kernel: workgroup-scope fence + barrier, as in the fused kernels above.  (Its workgroup IS one wave, so a wavefront-scope fence
// + wave barrier - the compiler keeps the order, nothing is waited for - would do: -DLK_SCAN_SYNC_WG=0, measured 10.86 against 10.85 ms per batch, green.  The
// waits are not what a bucket costs; the stronger form stays.)
#ifndef LK_SCAN_SYNC_WG
#define LK_SCAN_SYNC_WG 1
#endif
#if LK_SCAN_SYNC_WG
#define LK_SCAN_PHASE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __syncthreads(); } while (0)
#else
#define LK_SCAN_PHASE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#ifndef LK_SCAN_WAVES
#define LK_SCAN_WAVES 1   // waves per SIMD the register allocation aims at
#endif
template <bool XID>
__global__ void __launch_bounds__(LK_WAVE, LK_SCAN_WAVES)
    lk_rag_ov_scan_kernel(LkMap base, LkOverlay ov, LkParams pr, LkFilter* filters, const double* __restrict__ Q, LkRagged rg, const lk_point* __restrict__ d_pts,
                          int msg_kind, int* __restrict__ cur, int* __restrict__ fb_b, unsigned int* __restrict__ pending) {
    __shared__ WaveSmem sm;
    __shared__ double rows[64 * LK_ROW2];
    __shared__ int owner[LK_WAVE * 5];
    const int slot = blockIdx.x, lane = threadIdx.x;
    const LkMap pm = ov_slot_map(ov, (unsigned int)slot);
    const int nbk = rag_nb(rg, slot);
    int b = cur[slot];
    if (b >= nbk) {   // finished in an earlier launch.  Its LAST bucket may have left fallback items (it stopped behind them with nothing left to resume): they
        dev_bucket_begin_wave(pm);   // have been served by the launch in between - the next fallback launch must not find them again
        return;
    }
    LkFilter* f = &filters[slot];
    const double* T = rag_t(rg, slot);
    const unsigned long long* po = rag_pt_off(rg, slot);
    for (int e = lane; e < 900; e += LK_WAVE) sm.P[e] = f->P[e];
    if (lane < 36) sm.x[lane] = f->x[lane];
    double t_upd = f->last_update_t, t_pred = f->last_predict_t;
    __syncthreads();
    LkOvView ovv;
    ovv.keys = ov.keys + (size_t)slot * ov.hash_cap;
    ovv.hash_mask = ov.hash_cap - 1;
    ovv.match = ov.match + (size_t)slot * ov.nodes_cap;
    ovv.nodes = ov.nodes + (size_t)slot * ov.nodes_cap;
    ovv.bits = ov.bits + (size_t)slot * ov.bit_words;
    ResidualOut ro;
    ro.h6 = nullptr, ro.z = nullptr, ro.R = nullptr, ro.valid = nullptr, ro.world = nullptr, ro.ids = nullptr;
#pragma unroll 1
    for (; b < nbk; ++b) {
        // ---- front (lk_rag_ov_front_kernel)
        const double tb = T[b];
        if (msg_kind) {
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
        double totv = 0.0;   // tot[j] in lanes 0..31
        for (int i0 = 0; i0 < n; i0 += LK_WAVE) {
            __builtin_amdgcn_wave_barrier();   // the previous tile's reads of the rows are complete
            const double a = residual_tile<false, 3, XID, true, false>(base, pr, bc, reinterpret_cast<const float4*>(pts), i0 + lane, n, rows, lane, ro, (size_t)0, &ovv);
            totv += (lane < 29) ? a : 0.0;
        }
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
        if (N > 0) {
            wave_update_core(sm, totv, N, lane);
            t_upd = tb;
        }
        __syncthreads();
        // the insert reads the posterior through the filter record (see lk_rag_ov_front_kernel) - of the covariance only what load_bucket_const takes, the two 3 x 3
        // blocks of rotation and position: those twelve entries and the state go out every bucket, the whole covariance when the scan leaves the launch
        if (lane < 12) {
            const int e = lane < 3 ? lane : lane < 5 ? 28 + lane : lane == 5 ? 62 : lane < 9 ? 87 + lane : lane < 11 ? 115 + lane : 155;   // 0 1 2 31 32 62 | 93 94 95 124 125 155
            f->P[e] = sm.P[e];
        }
        if (lane < 36) f->x[lane] = sm.x[lane];
        LK_SCAN_PHASE_SYNC();
        dev_bucket_begin_wave(pm);
        for (int i = lane; i < n; i += LK_WAVE) {
            const int r = ov_reproject_point(base, ov, pr, filters, pts, i, (unsigned int)slot);
            ov.ptroot[(size_t)slot * ov.scan_cap + i] = r;
        }
        LK_SCAN_PHASE_SYNC();
        // ---- middle (lk_ov_mid_kernel<true>)
        const LkPtSrc src = {d_pts, 0, 0, rg.pt_off, rg.nb, rg.ldb, b, nullptr};
        ov_materialise_body<true>(base, ov, pr, (unsigned int)slot, 0, 1, lane, LK_WAVE);
        LK_SCAN_PHASE_SYNC();
        for (int i0 = 0; i0 < n; i0 += LK_WAVE) ov_point_geom_body(ov, pr, filters, src, (unsigned int)slot, (i0 + lane) >> 8, (i0 + lane) & 255);
        LK_SCAN_PHASE_SYNC();
        ov_root_lane_body(base, ov, pr, (unsigned int)slot, 0, 1, lane);
        LK_SCAN_PHASE_SYNC();
        // ---- tail (lk_ov_tail_kernel)
        if (!pm.counters[LK_CTR_ERR] && n != 0)
            dev_insert_root<false, true, true>(pm, pr, filters + slot, pts, (const lk_pt_rec*)nullptr, n, 0, 1, &base,
                                               ov.jobs + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS, ov.hash_cap, nullptr,
                                               ov.jobhdr + (size_t)slot * ov.hash_cap * LK_INLINE_GROUPS);
        LK_SCAN_PHASE_SYNC();
        ov_fit_eig_body(base, ov, pr, (unsigned int)slot, lane, LK_WAVE);
        LK_SCAN_PHASE_SYNC();
        ov_fit_group_body(base, ov, pr, (unsigned int)slot, owner, 0, 1, lane);
        LK_SCAN_PHASE_SYNC();
        if (!pm.counters[LK_CTR_ERR] && n != 0) dev_insert_apply<false>(pm, pr, filters + slot, pts, (const lk_pt_rec*)nullptr, n, 0, 1);
        LK_SCAN_PHASE_SYNC();
        // ---- what this launch cannot do: the fallback items of this bucket, or a pool that has run over (the call fails / grows and starts again)
        const unsigned int err = pm.counters[LK_CTR_ERR], nfb = pm.counters[LK_CTR_FALLBACK];
        if (err) {
            if (lane == 0) cur[slot] = nbk;
            return;
        }
        if (nfb) {
            for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
            if (lane == 0) {
                cur[slot] = b + 1, fb_b[slot] = b;
                atomicAdd(pending, 1u);
            }
            return;
        }
    }
    for (int e = lane; e < 900; e += LK_WAVE) f->P[e] = sm.P[e];
    dev_bucket_begin_wave(pm);   // finished: the launches behind this one find its work lists empty
    if (lane == 0) cur[slot] = nbk, fb_b[slot] = -1;
}
}   // extern "C++"

extern "C" int ov_scan_launch(lk_handle* h, bool xid, int S, hipStream_t st, const LkMap& fmap, const LkOverlay& ov, LkFilter* fl, const LkRagged& rg, const lk_point* d_pts,
                              int msg_kind, int* cur, int* fb_b, unsigned int* pending) {
    const auto scan_kernel = xid ? lk_rag_ov_scan_kernel<true> : lk_rag_ov_scan_kernel<false>;
    LAUNCH(h, "rag_ov_scan", hipLaunchKernelGGL(scan_kernel, dim3(S), dim3(LK_WAVE), 0, st, fmap, ov, h->pr, fl, h->d_Q, rg, d_pts, msg_kind, cur, fb_b, pending));
    return LK_OK;
}
