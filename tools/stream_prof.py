"""Per-kernel HIP-event timings of the single-stream config-3 path (one 100k-pt scan = 5 buckets)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lk_pkg; lk_pkg.load()
import bench as B
from legkilo_amd import binding, config, synth
P = config.LEG_FUSION
world, traj = synth.World(), synth.Trajectory()
cfg = config.make_config(P, n_slots=1, max_roots=1 << 17, max_nodes=1 << 18, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
t_after = B.build_map(g, world, traj, P, 5.0, 6)
print("map", g.map_stats())
scans = [synth.dense_scan(world, traj, t_after + 0.1 * k, P, n=B.N_PTS, n_buckets=B.N_BUCKETS, seed_scan=8008 + k, seed_noise=8108 + k) for k in range(4)]
off, dt = synth.buckets_of(scans[0])
d = g.device_malloc(4 * B.N_PTS * 16); g.h2d(d, np.concatenate(scans))
g.process_scan_dev(d, B.N_PTS, t_after, off, dt)
g.profile_reset(); g.profile_enable(1)
for k in range(1, 4):
    g.process_scan_dev(d + k * B.N_PTS * 16, B.N_PTS, t_after + 0.1 * k, off, dt)
g.profile_enable(0)
tot = 0
for name in ("small_bucket", "predict", "residual", "update", "reproject", "insert_light", "insert_group", "insert", "insert_fallback"):
    n, ms = g.profile_get(name)
    print(f"{name:14s} launches {n:3d}  avg {1e3*ms/max(n,1):8.1f} us")
    tot += ms
print("sum per scan %.1f us" % (1e3 * tot / 3), "map", g.map_stats())
