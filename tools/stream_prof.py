"""Per-kernel HIP-event timings (lk_profile_enable: every launch bracketed by events and synchronised) of the single-stream paths:
    python tools/stream_prof.py [--kind 5|51|vlp]
With profiling on the library takes the per-bucket launches (no scan-resident kernel, no pipelined insert)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lk_pkg  # noqa: E402

lk_pkg.load()
import bench as B  # noqa: E402
from legkilo_amd import binding, config, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="5")
args = ap.parse_args()
P = config.LEG_FUSION
B._init_worker()
world, traj = B._W, B._T
t0 = 5.0
warm_t = [t0 + 0.1 + 2.5 * k for k in range(6)]
t_after = warm_t[-1] + 0.5
jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
jobs += [("vlp", (t_after + 0.1 * k, 7007 + k)) for k in range(4)] if args.kind == "vlp" else \
    [("dense", (t_after + 0.1 * k, int(args.kind), 8008 + k, 8108 + k)) for k in range(4)]
gen = B.generate(jobs, min(32, os.cpu_count() or 1))
first, warm, scans = gen[0], gen[1:7], gen[7:]
cfg = config.make_config(P, n_slots=1, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
B.build_map(g, traj, P, first, warm, warm_t)
print("map", g.map_stats())
g.set_state(synth.initial_state(traj, t_after, P), 1e-6 * np.eye(30))
g.set_times(t_after, t_after)
g.process_scan(scans[0], t_after)
g.profile_reset()
g.profile_enable(1)
for k in range(1, 4):
    g.process_scan(scans[k], t_after + 0.1 * k)
g.profile_enable(0)
tot = 0.0
for name in ("small_bucket", "predict", "residual", "update", "reproject", "insert_root", "insert", "insert_fallback"):
    n, ms = g.profile_get(name)
    print(f"{name:16s} launches {n:5d}  avg {1e3 * ms / max(n, 1):8.1f} us")
    tot += ms
print("sum per scan %.1f us" % (1e3 * tot / 3), "map", g.map_stats())
