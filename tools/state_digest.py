"""Digest of the filter state and the map after a short stream - to compare library builds / environment switches bit for bit:
    LEGKILO_PREDICT_IN_ROOT=0 python tools/state_digest.py --kind 51   vs   LEGKILO_PREDICT_IN_ROOT=1 ...
The map is compared through its canonical form (tests/scenes.py: pool order depends on racing allocations, the trees do not)."""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lk_pkg  # noqa: E402

lk_pkg.load()
import bench as B  # noqa: E402
from legkilo_amd import binding, config, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="5")
ap.add_argument("--scans", type=int, default=6)
ap.add_argument("--save", default="", help="write state + map export to this .npz")
ap.add_argument("--compare", nargs=2, default=None, help="compare two saved runs bit for bit (state, canonical map) and exit")
args = ap.parse_args()
if args.compare:
    import scenes
    a, b = np.load(args.compare[0]), np.load(args.compare[1])
    same_x, same_P = np.array_equal(a["x"], b["x"]), np.array_equal(a["P"], b["P"])
    roots = scenes.maps_identical(a["map"], b["map"])
    print(f"state identical: x {same_x} P {same_P}; maps identical over {roots} roots")
    sys.exit(0 if same_x and same_P else 1)
P = config.LEG_FUSION
B._init_worker()
world, traj = B._W, B._T
t0 = 5.0
warm_t = [t0 + 0.1 + 2.5 * k for k in range(4)]
t_after = warm_t[-1] + 0.5
jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
jobs += [("dense", (t_after + 0.1 * k, int(args.kind), 8008 + k, 8108 + k)) for k in range(args.scans)]
gen = B.generate(jobs, min(32, os.cpu_count() or 1))
first, warm, scans = gen[0], gen[1:5], gen[5:]
cfg = config.make_config(P, n_slots=1, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
B.build_map(g, traj, P, first, warm, warm_t)
g.set_state(synth.initial_state(traj, t_after, P), 1e-6 * np.eye(30))
g.set_times(t_after, t_after)
neff = []
for k, sc in enumerate(scans):
    pose = g.process_scan(sc, t_after + 0.1 * k)
    neff.append(int(g.get_counters()["n_effect"]) if hasattr(g, "get_counters") else 0)
x, Pm = g.get_state()
h = hashlib.sha256()
h.update(np.ascontiguousarray(x).tobytes())
h.update(np.ascontiguousarray(Pm).tobytes())
print("state sha256", h.hexdigest()[:32], "pos", np.asarray(x)[9:12], "map", g.map_stats())
if args.save:
    np.savez(args.save, x=np.asarray(x), P=np.asarray(Pm), map=np.asarray(g.map_export()))
g.close()
