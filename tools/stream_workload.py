"""The single-stream workloads of the bench line on their own (no batch replay, no CPU leg): first frame + a few warm-up scans, then
N scans of config 3 (5 x 20k), config 3 with 51 buckets, or config-1 VLP scans, streamed with insert.  For kernel traces and A/B runs:
    python tools/stream_workload.py [--kind 5|51|vlp] [--scans 8] [--warm 6] [--spec 0|1]
Prints ms per scan (median) and the pipeline statistics."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lk_pkg  # noqa: E402

lk_pkg.load()
import bench as B  # noqa: E402
from legkilo_amd import binding, config, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="5")
ap.add_argument("--scans", type=int, default=8)
ap.add_argument("--warm", type=int, default=6)
ap.add_argument("--spec", type=int, default=-1)
ap.add_argument("--reps", type=int, default=1)
args = ap.parse_args()
P = config.LEG_FUSION
B._init_worker()
world, traj = B._W, B._T
t0 = 5.0
warm_t = [t0 + 0.1 + 2.5 * k for k in range(args.warm)]
t_after = (warm_t[-1] if warm_t else t0) + 0.5
jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
if args.kind == "vlp":
    jobs += [("vlp", (t_after + 0.1 * k, 7007 + k)) for k in range(args.scans)]
else:
    nb = int(args.kind)
    jobs += [("dense", (t_after + 0.1 * k, nb, 8008 + k, 8108 + k)) for k in range(args.scans)]
gen = B.generate(jobs, min(32, os.cpu_count() or 1))
first, warm, scans = gen[0], gen[1:1 + args.warm], gen[1 + args.warm:]
cfg = config.make_config(P, n_slots=1, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
if args.spec >= 0:
    g.stream_pipeline(args.spec)
B.build_map(g, traj, P, first, warm, warm_t)
print("map", g.map_stats(), file=sys.stderr)
g.stream_stats()   # (debug builds print and RESET their device-side histograms here: what follows is the timed stream alone)
print("---- timed stream", file=sys.stderr)
redo0 = g.stream_resident_redo()
for rep in range(args.reps):
    g.set_state(synth.initial_state(traj, t_after, P), 1e-6 * np.eye(30))
    g.set_times(t_after, t_after)
    tl = []
    if args.kind == "vlp":
        for k, sc in enumerate(scans):
            tc = time.perf_counter()
            g.process_scan(sc, t_after + 0.1 * k)
            tl.append(time.perf_counter() - tc)
    else:
        d = g.device_malloc(sum(sc.nbytes for sc in scans))
        g.h2d(d, np.concatenate(scans))
        o = 0
        for k, sc in enumerate(scans):
            off, dt = synth.buckets_of(sc)
            g.synchronize()
            tc = time.perf_counter()
            pose = g.process_scan_dev(d + o, len(sc), t_after + 0.1 * k, off, dt)
            tl.append(time.perf_counter() - tc)
            o += sc.nbytes
        g.device_free(d)
    print(f"kind {args.kind} spec {args.spec} xcc_mask {g.stream_grid_placement():#x} rep {rep}: ms/scan median {1e3 * float(np.median(tl[1:])):.3f} min {1e3 * min(tl[1:]):.3f} "
          f"buckets/scan {len(synth.buckets_of(scans[-1])[0]) - 1} points {len(scans[-1])} stats {g.stream_stats()} resident: redo {g.stream_resident_redo() - redo0} of {sum(len(synth.buckets_of(sc)[0]) - 1 for sc in scans)} buckets, (scans, relaunches) {g.stream_resident_stats()}")
g.close()
