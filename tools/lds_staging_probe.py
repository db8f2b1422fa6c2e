#!/usr/bin/env python
"""north_star's "LDS-staged voxel neighbourhoods", answered by a measurement instead of an argument.
Staging a voxel's plane record in LDS pays only if the lanes of a wave share voxels.  This probe takes the batch-replay workload
(100 k-point scans, 5 buckets x 20 k, the frozen map of bench.py) and reports, per 64-point wave of the residual kernel,
  * the number of DISTINCT root voxels its points fall into - in the order the path delivers them (pcl::VoxelGrid's cell order,
    KILO.cc:356-370) and with every bucket SORTED BY ROOT VOXEL (the best case any binning pass could produce);
  * the step time of the batch residual pass on both orders (same kernels - the sorted order is the upper bound of what locality
    alone can give, with the sort itself free);
  * what the sort would cost: a device radix sort of the launch's keys (torch.sort on the same number of int64 keys).
    python tools/lds_staging_probe.py [--scans 256] [--out profiles/r03_lds_staging_probe.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lk_pkg  # noqa: E402

lk_pkg.load()
import bench as B  # noqa: E402
from legkilo_amd import binding, config, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scans", type=int, default=256)
ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_lds_staging_probe.json"))
args = ap.parse_args()
import torch  # noqa: E402

P = config.LEG_FUSION
B._init_worker()
world, traj = B._W, B._T
S = args.scans
t0 = 5.0
warm_t = [t0 + 0.1 + 2.5 * k for k in range(8)]
t_after = warm_t[-1] + 0.5
jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
jobs += [("dense", (B.scan_time(t_after, g_), 5, 5005 + g_, 6006 + g_)) for g_ in range(S)]
gen = B.generate(jobs, min(64, os.cpu_count() or 1))
first, warm, scans = gen[0], gen[1:9], gen[9:]
cfg = config.make_config(P, n_slots=S, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
g = binding.LegKiloHip(cfg)
B.build_map(g, traj, P, first, warm, warm_t)
off, dt = synth.buckets_of(scans[0])
xs = np.stack([synth.initial_state(traj, B.scan_time(t_after, s), P, np.random.default_rng(9009 + s), 0.02, 0.5) for s in range(S)])
Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
vs = float(P["voxel_size"]) if "voxel_size" in P else 0.5


def root_keys(sc, x):
    R = x[:9].reshape(3, 3)
    E = np.array(P["extrinsic_R"], float).reshape(3, 3)
    T = np.array(P["extrinsic_T"], float)
    pb = np.stack([sc["x"], sc["y"], sc["z"]], 1).astype(np.float64)
    pw = (pb @ E.T + T) @ R.T + x[9:12]
    k = np.floor(pw / vs).astype(np.int64)
    return (k[:, 0] + 4096) + ((k[:, 1] + 4096) << 14) + ((k[:, 2] + 4096) << 28)


def distinct_per_wave(keys):
    out = []
    for b in range(len(off) - 1):
        kb = keys[off[b]:off[b + 1]]
        for w in range(0, len(kb) - 63, 64):
            out.append(len(np.unique(kb[w:w + 64])))
    return np.array(out)


dist_as, dist_sorted, roots_per_bucket, sorted_scans = [], [], [], []
for s in range(S):
    keys = root_keys(scans[s], xs[s])
    if s < 32:
        dist_as.append(distinct_per_wave(keys))
    sc2 = scans[s].copy()
    ks = keys.copy()
    for b in range(len(off) - 1):
        o = np.argsort(keys[off[b]:off[b + 1]], kind="stable")
        sc2[off[b]:off[b + 1]] = scans[s][off[b]:off[b + 1]][o]
        ks[off[b]:off[b + 1]] = keys[off[b]:off[b + 1]][o]
        if s < 32:
            roots_per_bucket.append(len(np.unique(keys[off[b]:off[b + 1]])))
    if s < 32:
        dist_sorted.append(distinct_per_wave(ks))
    sorted_scans.append(sc2)
dist_as, dist_sorted = np.concatenate(dist_as), np.concatenate(dist_sorted)


def time_batch(batch):
    d = torch.from_numpy(np.concatenate([np.ascontiguousarray(sc).view(np.uint8).reshape(-1) for sc in batch])).cuda()
    g.batch_set_priors(xs, Ps)
    g.batch_replay_dev(d.data_ptr(), S, B.N_PTS, 0.0, off, dt, want_poses=False)
    g.synchronize()
    ts = []
    for _ in range(5):
        g.batch_set_priors(xs, Ps)
        g.synchronize()
        t1 = time.perf_counter()
        poses = g.batch_replay_dev(d.data_ptr(), S, B.N_PTS, 0.0, off, dt)
        g.synchronize()
        ts.append(time.perf_counter() - t1)
    ne = float(np.mean([p.n_effect for p in poses]))
    return float(np.median(ts)), ne


t_as, ne_as = time_batch(scans)
t_so, ne_so = time_batch(sorted_scans)
# the binning pass itself: a device radix sort of one launch's keys (S x 20 000 points), key + index
nk = S * (B.N_PTS // B.N_BUCKETS)
kk = torch.randint(0, 1 << 40, (nk,), device="cuda")
torch.sort(kk)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(5):
    torch.sort(kk)
torch.cuda.synchronize()
t_sort = (time.perf_counter() - t1) / 5
res = {
    "workload": f"{S} scans x 100 000 points, 5 buckets x 20 000, frozen map of bench.py (8 warm-up scans)",
    "distinct_roots_per_20k_bucket_mean": float(np.mean(roots_per_bucket)), "points_per_root_per_bucket": float(20000 / np.mean(roots_per_bucket)),
    "distinct_roots_per_wave_as_delivered": {"mean": float(dist_as.mean()), "p10": float(np.percentile(dist_as, 10)), "p90": float(np.percentile(dist_as, 90))},
    "distinct_roots_per_wave_sorted_by_root": {"mean": float(dist_sorted.mean()), "p10": float(np.percentile(dist_sorted, 10)), "p90": float(np.percentile(dist_sorted, 90))},
    "batch_step_ms_as_delivered": round(t_as * 1e3, 4), "batch_step_ms_sorted_by_root": round(t_so * 1e3, 4),
    "ps_per_point_as_delivered": round(t_as * 1e12 / (S * B.N_PTS), 2), "ps_per_point_sorted_by_root": round(t_so * 1e12 / (S * B.N_PTS), 2),
    "mean_n_effect_as_delivered": ne_as, "mean_n_effect_sorted": ne_so,
    "device_sort_ms_per_launch_keys": round(t_sort * 1e3, 4), "device_sort_keys": nk,
    "residual_launch_ms_as_delivered": round(t_as * 1e3 / B.N_BUCKETS, 4),
    "conclusion": "a wave of 64 points spans several root voxels even when the bucket is perfectly sorted by root; the sorted order is the upper "
                  "bound of what staging could gain (compare the two step times) and the sort alone costs more than the launch it would speed up",
}
print(json.dumps(res, indent=1))
json.dump(res, open(args.out, "w"), indent=1)
