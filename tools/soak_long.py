#!/usr/bin/env python
"""One-off long soak: N consecutive 100 k-point scans with map insert, a map slide every 10 scans, HIP path vs oracle."""
import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import lk_pkg; lk_pkg.load()
from legkilo_amd import synth, binding as hip_lib
import oracle_binding as oracle_lib, scenes
oracle_lib.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
scene = scenes.Scene()
o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
g = hip_lib.LegKiloHip(scene.cfg())
t0 = 3.0
for obj in (o, g):
    x0 = scenes.init_filter(obj, scene, t0)
    scenes.first_frame(obj, scene, t0, x0, dense=60000)
worst_pos, worst_ne, tg = 0.0, 0, 0.0
traj_o, traj_g, traj_gt = [], [], []
for k in range(N):
    tb = t0 + 0.1 * k
    pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=9100 + k, seed_noise=9200 + k)
    po, _ = o.process_scan(pts, tb)
    t1 = time.perf_counter(); pg, _ = g.process_scan(pts, tb); tg += time.perf_counter() - t1
    xo, _ = o.get_state(); xg, _ = g.get_state()
    traj_o.append(xo[9:12].copy()); traj_g.append(xg[9:12].copy()); traj_gt.append(np.asarray(scene.traj.pos(tb + 0.08)).reshape(3))
    worst_pos = max(worst_pos, float(np.abs(xo[9:12] - xg[9:12]).max()))
    worst_ne = max(worst_ne, abs(int(po.n_effect) - int(pg.n_effect)))
    if k % 10 == 9:
        so, sg = o.map_slide(xo[9:12], 0.0, 30), g.map_slide(xg[9:12], 0.0, 30)
        print("scan", k, "slide", so, sg, "stats", g.map_stats(), "worst pos %.2e" % worst_pos, "worst dN", worst_ne, flush=True)
ko, kg = set(scenes.canon_map(o.map_export())), set(scenes.canon_map(g.map_export()))
print("done: voxel sets equal:", ko == kg, len(kg), "worst pos %.2e" % worst_pos, "worst dN", worst_ne, "gpu ms/scan %.2f" % (tg / N * 1e3))
ao, ag = scenes.ate(np.array(traj_o), np.array(traj_gt)), scenes.ate(np.array(traj_g), np.array(traj_gt))
print("ATE vs ground truth: oracle %.6f m, hip %.6f m, delta %.3e m; oracle-vs-hip ATE %.3e m" % (ao, ag, abs(ao - ag), scenes.ate(np.array(traj_o), np.array(traj_g))))
