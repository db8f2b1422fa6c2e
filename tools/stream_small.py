#!/usr/bin/env python
"""Per-scan time of a config-1 style stream (VLP-16 scan, ~4 k points after the voxel-grid filter, 2 ms time bins ->
many tiny buckets) through lk_process_scan: the regime of a real robot, bound by per-bucket latency."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lk_pkg; lk_pkg.load()
from legkilo_amd import binding, synth
import scenes
sc = scenes.Scene()
g = binding.LegKiloHip(sc.cfg())
t0 = 1.0
x0 = scenes.init_filter(g, sc, t0)
scenes.first_frame(g, sc, t0, x0)
inputs = []
for k in range(10):
    tb = t0 + 0.1 * k
    inputs.append((scenes.vlp_scan_input(sc, tb, k), synth.imu_stream(sc.traj, tb, tb + 0.1, seed=3003 + k), tb))
g.process_scan(inputs[0][0], inputs[0][2], imus=inputs[0][1])
ts = time.perf_counter()
nb = 0
for ds, imus, tb in inputs[1:]:
    pose, _ = g.process_scan(ds, tb, imus=imus)
    nb += pose.n_buckets
dt = (time.perf_counter() - ts) / (len(inputs) - 1)
print(f"{os.environ.get('LEGKILO_HIP_LIB', 'default')}: {dt * 1e3:.2f} ms per scan, {len(inputs[1][0])} points, {nb / (len(inputs) - 1):.0f} buckets per scan, "
      f"{dt * 1e6 / (nb / (len(inputs) - 1)):.1f} us per bucket")
g.close()
