#!/usr/bin/env python
"""Time of the ragged config-1 batch (2048 VLP-16 scans, ~360 two-ms buckets each, one wave per scan: lk_scan_wave_kernel) and
of the live config-1 stream, for A/B runs: LEGKILO_HIP_LIB=... python tools/ab_ragged.py [kin]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import lk_pkg; lk_pkg.load()
from legkilo_amd import abi, binding, config, synth
import scenes
import torch
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3) if mode == "kin" else None)
S = int(os.environ.get("AB_S", "2048"))
g = binding.LegKiloHip(sc.cfg(n_slots=S, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 16))
t0 = 1.0
x0 = scenes.init_filter(g, sc, t0)
scenes.first_frame(g, sc, t0, x0, dense=100000)
U = 8
scans = [scenes.vlp_scan_input(sc, t0 + 0.1 * k, k) for k in range(U)]
tbs = [t0 + 0.1 * k for k in range(U)]
tile = np.arange(S) % U
xs = np.stack([synth.initial_state(sc.traj, tbs[u], sc.P) for u in tile])
xs[:, 9:12] += np.random.default_rng(1).normal(0, 0.005, (S, 3))
Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
allp = np.ascontiguousarray(np.concatenate([scans[u] for u in tile]))
scan_off = np.r_[0, np.cumsum([len(scans[u]) for u in tile])]
tabs = [synth.buckets_of(s_) for s_ in scans]
kw = {}
if mode == "imu":
    kw["imus"] = [synth.imu_stream(sc.traj, tbs[u], tbs[u] + 0.1, seed=3003 + u) for u in tile]
if mode == "kin":
    ks = [synth.kin_stream(sc.traj, tbs[u], tbs[u] + 0.1, sc.P, seed=3003 + u) for u in range(U)]
    kw["kins"] = [ks[u] for u in tile]
tables = g.ragged_tables(scan_off, [tabs[u][0] for u in tile], [tabs[u][1] for u in tile], [tbs[u] for u in tile], **kw)
dev = torch.device("cuda")
d = torch.empty(allp.nbytes, dtype=torch.uint8, device=dev)
g.h2d(d.data_ptr(), allp)
dx, dP = torch.from_numpy(xs).to(dev), torch.from_numpy(Ps).to(dev)
g.set_acc_norm(9.81)
def run():
    g.batch_set_priors_dev(dx.data_ptr(), dP.data_ptr(), S)
    return g.batch_replay_ragged_dev(d.data_ptr(), tables)
run()
t = time.perf_counter()
for _ in range(3):
    poses = run()
el = (time.perf_counter() - t) / 3
p = np.frombuffer(poses, dtype=abi.pose_dtype())
nb = float(p["n_buckets"].mean())
print(f"{os.environ.get('LEGKILO_HIP_LIB', 'default').split('/')[-1]:28s} {mode}: batch {el * 1e3:7.2f} ms, {nb:.0f} buckets/scan, {el * 1e6 / nb:.2f} us per bucket level, "
      f"n_eff {p['n_effect'].mean():.1f}, checksum {p['pos'].sum():.12f}")
