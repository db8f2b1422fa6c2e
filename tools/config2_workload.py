#!/usr/bin/env python
"""BASELINE config 2 at bandwidth size (lk_batch_residuals_dev) as a stand-alone workload for the profiler: the bench's scene and map, the
FIRST S scans of the bench batch (same generator jobs, same cache keys as bench.py --cache-dir), every scan under its prior, residual rows
materialised in HBM, R launches.  Two calibration launches of KNOWN byte counts run in the same process (a torch fill and a torch
elementwise multiply over the h6 buffer: WRITE_SIZE / FETCH_SIZE are uncalibrated on gfx950 - MI355X_MICROARCH.md, HBM section).
Prints one JSON line.

    python tools/config2_workload.py --slots 256 --reps 5 [--cache-dir /tmp/lkcache]
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
    ap.add_argument("--slots", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--map-warm", type=int, default=20)
    ap.add_argument("--cache-dir", default="")
    ap.add_argument("--no-calib", action="store_true")
    args = ap.parse_args()
    S = args.slots
    P = config.LEG_FUSION
    traj = synth.Trajectory()
    t0 = 5.0
    warm_t = [t0 + 3.0 * k for k in range(args.map_warm)]
    jobs = [("dense", (bench.scan_time(5.0, u), bench.N_BUCKETS, 5005 + u, 1_000_003 + u)) for u in range(S)]
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
    scans, first, warm = gen[:S], gen[S], gen[S + 1:]
    xs = np.stack([synth.initial_state(traj, bench.scan_time(5.0, s), P, np.random.default_rng(9009 + s), 0.02, 0.5) for s in range(S)])
    Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
    import torch

    dev = torch.device("cuda", 0)
    cfg = config.make_config(P, device_id=0, n_slots=S, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
    g = binding.LegKiloHip(cfg)
    bench.build_map(g, traj, P, first, warm, warm_t)
    N = S * bench.N_PTS
    d_pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(scans)).view(np.uint8)).to(dev)
    d_rows = torch.empty((N, 8), dtype=torch.float64, device=dev)
    d_h6 = d_rows.view(-1)[: N * 6].view(N, 6)     # calibration source below: N x 48 B
    d_v = torch.empty(N, dtype=torch.uint8, device=dev)
    g.batch_set_priors(xs, Ps)
    torch.cuda.synchronize()

    def run():
        g.batch_residuals_dev(d_pts.data_ptr(), S, bench.N_PTS, d_rows.data_ptr(), d_v.data_ptr())

    run()
    g.synchronize()
    ts = time.perf_counter()
    for _ in range(args.reps):
        run()
    g.synchronize()
    t = (time.perf_counter() - ts) / args.reps
    out = {"slots": S, "points_per_launch": N, "launches": args.reps + 1, "ms_per_launch_wall": round(t * 1e3, 4), "ps_per_point": round(t * 1e12 / N, 2),
           "matched_fraction": round(float(d_v.to(torch.float32).mean().item()), 4), "alg_bytes_per_point": bench.C2_BYTES_PER_POINT,
           "alg_GBs": round(bench.C2_BYTES_PER_POINT * N / t / 1e9, 1)}
    if not args.no_calib:
        # known byte counts, same process, same profiler pass: fill = N x 48 B written; copy = N x 48 B read + N x 48 B written
        cal = torch.empty_like(d_h6)
        torch.cuda.synchronize()
        for _ in range(3):
            cal.fill_(1.5)
        for _ in range(3):
            torch.mul(d_h6, 2.0, out=cal)     # an elementwise kernel: N x 48 B read (16 B per lane, coalesced), N x 48 B written
        torch.cuda.synchronize()
        out["calibration"] = {"fill_bytes_written": N * 48, "mul_bytes_read": N * 48, "mul_bytes_written": N * 48, "fills": 3, "muls": 3}
    print(json.dumps(out))
    g.close()


if __name__ == "__main__":
    main()
