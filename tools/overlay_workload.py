#!/usr/bin/env python
"""The overlay batch replay (lk_batch_replay_overlay_dev) as a stand-alone workload for profilers and sweeps: the bench's scene and map
(first frame + warm-up scans with insert), U distinct 100 000-point scans tiled to S slots with distinct priors, R replays.
Prints one JSON line: ms per batch, scans/s, per-kernel ms (event pairs around every launch, a separate replay).

    python tools/overlay_workload.py --slots 1024 --unique 32 --reps 3 [--cache-dir /tmp/lkcache]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (scene generation, map building)
from legkilo_amd import binding, config, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--unique", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--map-warm", type=int, default=20)
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--stats", action="store_true", help="call lk_stream_stats after the replays (a -DLK_DEBUG_INS build prints its insert-phase stamps and histograms there)")
    ap.add_argument("--cache-dir", default="")
    args = ap.parse_args()
    S, U = args.slots, min(args.unique, args.slots)
    P = config.LEG_FUSION
    traj = synth.Trajectory()
    t0 = 5.0
    warm_t = [t0 + 3.0 * k for k in range(args.map_warm)]
    jobs = [("dense", (bench.scan_time(5.0, 32 * u), bench.N_BUCKETS, 5005 + 32 * u, 1_000_003 + 32 * u)) for u in range(U)]
    jobs.append(("first", (t0,)))
    jobs += [("dense", (tb, bench.N_BUCKETS, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]

    def cpath(j):
        return os.path.join(args.cache_dir, "lk_" + j[0] + "_" + "_".join(repr(v) for v in j[1]) + ".npy")

    if args.cache_dir:
        os.makedirs(args.cache_dir, exist_ok=True)
        missing = [j for j in jobs if not os.path.exists(cpath(j))]
        for j, arr in zip(missing, bench.generate(missing, min(64, os.cpu_count() or 1))):
            np.save(cpath(j), arr)
        gen = [np.load(cpath(j)) for j in jobs]
    else:
        gen = bench.generate(jobs, min(64, os.cpu_count() or 1))
    scans, first, warm = gen[:U], gen[U], gen[U + 1:]
    off, dt = synth.buckets_of(scans[0])
    tile = np.arange(S) % U
    xs = np.stack([synth.initial_state(traj, bench.scan_time(5.0, 32 * int(tile[s])), P, np.random.default_rng(9009 + s), 0.02, 0.5) for s in range(S)])
    Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
    cfg = config.make_config(P, device_id=0, n_slots=S, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
    g = binding.LegKiloHip(cfg)
    bench.build_map(g, traj, P, first, warm, warm_t)
    allpts = np.ascontiguousarray(np.concatenate([scans[u] for u in tile]))
    d_pts, d_x, d_P = g.device_malloc(allpts.nbytes), g.device_malloc(xs.nbytes), g.device_malloc(Ps.nbytes)
    g.h2d(d_pts, allpts)
    g.h2d(d_x, np.ascontiguousarray(xs))
    g.h2d(d_P, np.ascontiguousarray(Ps))

    def run(want=False):
        g.batch_set_priors_dev(d_x, d_P, S)
        g.synchronize()
        t = time.perf_counter()
        p = g.batch_replay_overlay_dev(d_pts, S, bench.N_PTS, 0.0, off, dt, want_poses=want)
        return time.perf_counter() - t, p

    run()
    ts = [run()[0] for _ in range(args.reps)]
    out = {"slots": S, "unique": U, "ms_per_batch": round(float(np.median(ts)) * 1e3, 3), "scans_per_s": round(S / float(np.median(ts)), 1),
           "private_max": dict(zip(("roots", "nodes", "blocks"), g.overlay_stats()))}
    if not args.no_profile:
        g.profile_reset()
        g.profile_enable(1)
        run()
        g.profile_enable(0)
        out["kernel_ms"] = {k: round(g.profile_get(k)[1], 3) for k in ("ov_reset", "predict", "ov_residual", "update", "ov_begin", "ov_reproject", "ov_materialise", "ov_point_geom", "ov_root_lane", "ov_insert_root", "ov_base_sums", "ov_fit_eig", "ov_fit_lane",
                                                                          "ov_insert_apply", "ov_insert_fallback")}
    if args.stats:
        g.stream_stats()
    print(json.dumps(out))
    g.close()


if __name__ == "__main__":
    main()
