#!/usr/bin/env python
"""Where does a config-4 scan's path time go?  The live run of bench.py's config-4 extra (diter.yaml, Ouster-shaped scans, leg fusion) three ways on
copies of one handle's state: with its ~49 kinematic + IMU messages per scan, with the same stamps as plain IMU messages, and with no messages at all
(timing only - the filter is re-armed from the full run's posterior before every scan, so that all three replay the same scan on the same map)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from legkilo_amd import binding, config, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P4 = config.DITER
traj = synth.Trajectory()
T0 = bench.T0_CONFIG4
jobs = [("ouster", (T0, 3999, True))] + [("ouster", (T0 + 0.1 * k, 4000 + k, False)) for k in range(N)]
gen = bench.generate(jobs, min(32, os.cpu_count() or 1))
static, msgs = gen[0], gen[1:]
cfg4 = config.make_config(P4, device_id=0, n_slots=1, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 15, max_scan_points=1 << 17)
res = {}
for mode in ("kin", "none"):
    g = binding.LegKiloHip(cfg4)
    x0 = synth.initial_state(traj, T0, P4)
    g.set_state(x0, 1e-6 * np.eye(30))
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    g.set_times(T0, T0)
    dec0, _, _ = g.decode_scan(static.tobytes(), len(static), bench.OUSTER_MSG_LAYOUT, P4["time_scale"], P4["filter_num"], P4["blind"], header_stamp=T0)
    xb = bench.xyz_of(dec0)
    g.map_build(bench.world_of(x0, xb, P4), xb)
    ts, nb = [], []
    for k in range(N):
        tb = T0 + 0.1 * k
        dec, b_, _ = g.decode_scan(msgs[k].tobytes(), len(msgs[k]), bench.OUSTER_MSG_LAYOUT, P4["time_scale"], P4["filter_num"], P4["blind"], header_stamp=tb)
        ds = g.preprocess_scan(dec, P4["voxel_grid_resolution"])
        kins = synth.kin_stream(traj, tb, tb + 0.1, P4, seed=5000 + k)
        if mode == "none":   # keep the filter near the truth without its messages: re-arm from the true pose
            g.set_state(synth.initial_state(traj, b_, P4), 1e-6 * np.eye(30))
            g.set_times(b_, b_)
        tc = time.perf_counter()
        pose, _ = g.process_scan(ds, b_, kins=kins if mode == "kin" else None)
        ts.append(time.perf_counter() - tc)
        nb.append(int(pose.n_buckets))
    res[mode] = {"ms_per_scan": round(float(np.median(ts[1:])) * 1e3, 3), "buckets": round(float(np.mean(nb)), 1), "relaunches": g.stream_resident_stats()[1]}
    g.close()
res["us_per_bucket_without_messages"] = round(res["none"]["ms_per_scan"] * 1e3 / res["none"]["buckets"], 2)
res["us_per_kin_message"] = round((res["kin"]["ms_per_scan"] - res["none"]["ms_per_scan"]) * 1e3 / 49.4, 1)
print(json.dumps(res))

# ---- the same question on the frozen-map scan wave (lk_scan_wave_kin_kernel: one wave per scan, the pure filter chain without the insert team)
g = binding.LegKiloHip(config.make_config(P4, device_id=0, n_slots=8, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 15, max_scan_points=1 << 17))
x0 = synth.initial_state(traj, T0, P4)
g.set_state(x0, 1e-6 * np.eye(30))
g.init_process_cov_q()
g.set_acc_norm(9.81)
g.set_times(T0, T0)
dec0, _, _ = g.decode_scan(static.tobytes(), len(static), bench.OUSTER_MSG_LAYOUT, P4["time_scale"], P4["filter_num"], P4["blind"], header_stamp=T0)
xb = bench.xyz_of(dec0)
g.map_build(bench.world_of(x0, xb, P4), xb)
scans, tbs, kk, xs = [], [], [], []
for k in range(8):
    tb = T0 + 0.1 * k
    dec, b_, _ = g.decode_scan(msgs[k].tobytes(), len(msgs[k]), bench.OUSTER_MSG_LAYOUT, P4["time_scale"], P4["filter_num"], P4["blind"], header_stamp=tb)
    scans.append(g.preprocess_scan(dec, P4["voxel_grid_resolution"]))
    tbs.append(b_)
    kk.append(synth.kin_stream(traj, tb, tb + 0.1, P4, seed=5000 + k))
    xs.append(synth.initial_state(traj, b_, P4))
Ps = [1e-6 * np.eye(30)] * 8
out = {}
for name, kw in (("kin", dict(kins=kk)), ("none", {})):
    g.batch_replay_ragged(scans, tbs, xs, Ps, **kw)
    t = []
    for _ in range(3):
        tc = time.perf_counter()
        g.batch_replay_ragged(scans, tbs, xs, Ps, **kw)
        t.append(time.perf_counter() - tc)
    out[name] = min(t) * 1e3
nbk = np.mean([len(synth.buckets_of(s)[1]) for s in scans])
print(json.dumps({"frozen_scan_wave_ms_8_scans": out, "buckets": float(nbk), "us_per_bucket_no_messages": out["none"] * 1e3 / nbk, "us_per_kin_message": (out["kin"] - out["none"]) * 1e3 / 49.4}))
