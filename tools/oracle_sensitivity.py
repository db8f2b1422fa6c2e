#!/usr/bin/env python
"""CPU-only: two copies of the oracle on the same 100 k-point scans (map insert + a slide every 10 scans); copy B's
initial x position is perturbed by EPS metres.  Shows how a sub-ulp-scale difference grows in the closed loop, i.e. the
floor below which closed-loop trajectories of ANY two builds of the algorithm cannot be compared.
Usage: oracle_sensitivity.py N_SCANS EPS"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import lk_pkg; lk_pkg.load()
from legkilo_amd import synth
import oracle_binding as oracle_lib, scenes
oracle_lib.build()
N = int(sys.argv[1]); eps = float(sys.argv[2])
scene = scenes.Scene()
a = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
b = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
t0 = 3.0
for j, obj in enumerate((a, b)):
    x0 = scenes.init_filter(obj, scene, t0)
    scenes.first_frame(obj, scene, t0, x0, dense=60000)
x, P = b.get_state(); x = x.copy(); x[9] += eps; b.set_state(x, P)
ea, eb = [], []
for k in range(N):
    tb = t0 + 0.1 * k
    pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=9100 + k, seed_noise=9200 + k)
    pa, _ = a.process_scan(pts, tb); pb, _ = b.process_scan(pts, tb)
    xa, _ = a.get_state(); xb, _ = b.get_state()
    if k % 10 == 9:
        a.map_slide(xa[9:12], 0.0, 30); b.map_slide(xb[9:12], 0.0, 30)
    if k % 5 == 4: print(k, "pos diff %.2e" % np.abs(xa[9:12]-xb[9:12]).max(), "dN", int(pa.n_effect) - int(pb.n_effect), flush=True)
