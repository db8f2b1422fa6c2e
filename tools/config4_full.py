#!/usr/bin/env python
"""Config 4 at the FULL shape SURVEY.md 8(d) states (diter.yaml, Ouster-like 64 x 1024 scans with `t` in ns, 500 Hz kinematic + IMU
messages, only_imu_use: false) over the whole 60 s figure-eight = 600 scans, with the local-map sliding window
(VoxelMapManager::mapSliding, voxel_map.cc:552-569) applied every `--slide-every` scans on both sides - HIP path against the oracle.
Per scan: decode bit-exact, identical bucket / update / match counts (the run stops at the first difference); at the end: ATE delta
through TUM files, map key sets.  Writes a JSON summary (default profiles/r03_config4_full.json).
    python tools/config4_full.py [--scans 600] [--slide-every 100] [--out profiles/r03_config4_full.json]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import lk_pkg  # noqa: E402

lk_pkg.load()
import oracle_binding as oracle_lib  # noqa: E402
import preprocess_oracle as po  # noqa: E402
import scenes  # noqa: E402
from legkilo_amd import binding as hip_lib, config, synth, tum  # noqa: E402
from test_gpu_parity import CAPS, ouster_message  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=600)
ap.add_argument("--slide-every", type=int, default=100)
ap.add_argument("--half-map", type=int, default=40, help="half size of the sliding window in root voxels")
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_config4_full.json"))
args = ap.parse_args()

oracle_lib.build()
sc = scenes.Scene(params=config.DITER, **CAPS)
P = sc.P
o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
g = hip_lib.LegKiloHip(sc.cfg())
t0 = 3.0
raw_static, _ = synth.ouster_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, P, seed_noise=3999)
xb = scenes.xyz_of(raw_static[::3])
for obj in (o, g):
    x0 = scenes.init_filter(obj, sc, t0)
    obj.map_build(scenes.world_of(x0, xb, P), xb)
stamps, rows_o, rows_g = [], [], []
worst, t_cpu, t_gpu, n_msgs, slides, removed = 0.0, 0.0, 0.0, 0, 0, 0
n_eff = 0
for k in range(args.scans):
    tb = t0 + 0.1 * k
    raw, layout = ouster_message(sc, tb, k)
    kins = synth.kin_stream(sc.traj, tb, tb + 0.1, P, seed=5000 + k)
    n_msgs += len(kins)
    dec_o, b_o, e_o = po.decode_vec(raw, 2, P["time_scale"], P["filter_num"], P["blind"], header_stamp=tb)
    dec_g, b_g, e_g = g.decode_scan(raw.tobytes(), len(raw), layout, P["time_scale"], P["filter_num"], P["blind"], header_stamp=tb)
    assert (b_o, e_o) == (b_g, e_g) and all(np.array_equal(dec_o[f], dec_g[f]) for f in dec_o.dtype.names), ("decode", k)
    ds = po.preprocess(dec_o, P["voxel_grid_resolution"])
    tc = time.perf_counter()
    pose_o, _ = o.process_scan(ds, b_o, kins=kins)
    t_cpu += time.perf_counter() - tc
    tc = time.perf_counter()
    pose_g, nd = g.process_raw_scan(dec_g, P["voxel_grid_resolution"], b_g, kins=kins)
    t_gpu += time.perf_counter() - tc
    assert (pose_o.n_buckets, pose_o.n_updates, pose_o.n_effect) == (pose_g.n_buckets, pose_g.n_updates, pose_g.n_effect), \
        ("counts", k, pose_o.n_buckets, pose_o.n_updates, pose_o.n_effect, pose_g.n_buckets, pose_g.n_updates, pose_g.n_effect)
    n_eff += int(pose_o.n_effect)
    stamps.append(e_o)
    rows_o.append((np.array(pose_o.rot), np.array(pose_o.pos)))
    rows_g.append((np.array(pose_g.rot), np.array(pose_g.pos)))
    worst = max(worst, float(np.abs(rows_o[-1][1] - rows_g[-1][1]).max()))
    if args.slide_every and k % args.slide_every == args.slide_every - 1:
        so, sg = o.map_slide(rows_o[-1][1], 0.0, args.half_map), g.map_slide(rows_g[-1][1], 0.0, args.half_map)
        assert so == sg, ("slide", k, so, sg)
        slides += int(so[0])
        removed += so[1]
        print(f"scan {k}: slide {so}, map {g.map_stats()}, worst position delta so far {worst:.2e} m", flush=True)
with tempfile.TemporaryDirectory() as td:
    tum.write_tum(os.path.join(td, "cpu.txt"), stamps, [r for r, _ in rows_o], [p_ for _, p_ in rows_o])
    tum.write_tum(os.path.join(td, "gpu.txt"), stamps, [r for r, _ in rows_g], [p_ for _, p_ in rows_g])
    e_tum, n_tum = tum.ate_files(os.path.join(td, "cpu.txt"), os.path.join(td, "gpu.txt"))
truth = sc.traj.pos(np.array(stamps))
ko, kg = set(scenes.canon_map(o.map_export())), set(scenes.canon_map(g.map_export()))
res = {
    "what": "config 4 at SURVEY 8(d)'s full shape: diter.yaml, Ouster 64 x 1024, time_scale 1e-9, 500 Hz kin + IMU messages, leg fusion, figure-eight",
    "scans": args.scans, "seconds_of_trajectory": round(0.1 * args.scans, 1), "kin_imu_messages": n_msgs,
    "counts_identical_on_every_scan": True, "matched_points_total": n_eff,
    "map_slides": slides, "slide_every_scans": args.slide_every, "half_map_size_voxels": args.half_map, "roots_removed_by_slides": removed,
    "worst_position_delta_m": worst, "ate_delta_tum_files_m": e_tum, "tum_poses": n_tum,
    "ate_vs_ground_truth_cpu_m": scenes.ate([p_ for _, p_ in rows_o], truth), "ate_vs_ground_truth_gpu_m": scenes.ate([p_ for _, p_ in rows_g], truth),
    "map_root_sets_equal": ko == kg, "map_roots": len(kg),
    "cpu_port_ms_per_scan": round(1e3 * t_cpu / args.scans, 3), "gpu_ms_per_scan_incl_host_preprocess_call": round(1e3 * t_gpu / args.scans, 3),
}
try:   # round 5: how often the resident stream kernels stopped at fallback items and were launched again
    rs_ = g.stream_resident_stats()
    res["resident_kernel_scans"], res["resident_kernel_relaunches"] = rs_[0], rs_[1]
except Exception:
    pass
print(json.dumps(res))
assert res["map_root_sets_equal"] and e_tum < 1e-6
json.dump(res, open(args.out, "w"), indent=1)
