"""Does the root kernel's duration depend on WHERE the pools were allocated?  Several handles in ONE process (kept alive, so every
handle gets different addresses), the same map and the same scans through each, per-kernel HIP-event timings (lk_profile_enable).
    python tools/placement_probe.py [--handles 4]"""
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
ap.add_argument("--handles", type=int, default=4)
args = ap.parse_args()
P = config.LEG_FUSION
B._init_worker()
world, traj = B._W, B._T
t0 = 5.0
warm_t = [t0 + 0.1 + 2.5 * k for k in range(6)]
t_after = warm_t[-1] + 0.5
jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
jobs += [("dense", (t_after + 0.1 * k, 5, 8008 + k, 8108 + k)) for k in range(9)]
gen = B.generate(jobs, min(32, os.cpu_count() or 1))
first, warm, scans = gen[0], gen[1:7], gen[7:]
cfg = config.make_config(P, n_slots=1, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
keep = []
for hnd in range(args.handles):
    g = binding.LegKiloHip(cfg)
    keep.append(g)
    B.build_map(g, traj, P, first, warm, warm_t)
    g.set_state(synth.initial_state(traj, t_after, P), 1e-6 * np.eye(30))
    g.set_times(t_after, t_after)
    g.process_scan(scans[0], t_after)
    g.profile_reset()
    g.profile_enable(1)
    for k in range(1, 9):
        g.process_scan(scans[k], t_after + 0.1 * k)
    g.profile_enable(0)
    out = []
    for name in ("residual", "update", "reproject", "insert_root", "insert", "insert_fallback"):
        n, ms = g.profile_get(name)
        out.append(f"{name} {1e3 * ms / max(n, 1):6.1f}")
    print(f"handle {hnd}: us per launch: " + "  ".join(out), flush=True)
