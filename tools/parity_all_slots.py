#!/usr/bin/env python3
"""All-slots parity record of the bench batch: the oracle replays EVERY scan of the 1 024-scan config-5 batch -
all frozen-map slots (lk_batch_replay_dev) and all slots of the batch WITH insert (lk_batch_replay_overlay_dev, each scan on a private
copy of the device's map blob, insert on) - not the 240 / 24 samples of the bench line.  Writes profiles/r05_parity_all_slots.json.

The GPU work runs in a child process (this one never touches HIP), so the oracle side can use a fork pool over the host's cores:
    python tools/parity_all_slots.py [--scans 1024] [--workers 64] [--out profiles/r05_parity_all_slots.json]
Exit code 3 when a slot violates the bench line's tolerances (poses 1e-7 m / rad, counts equal up to 2 % of the slots).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench  # noqa: E402
from legkilo_amd import abi as _abi  # noqa: E402
from legkilo_amd import config, synth  # noqa: E402

N_PTS, N_BUCKETS = bench.N_PTS, bench.N_BUCKETS


def make_cfg(S):
    return config.make_config(config.LEG_FUSION, device_id=0, n_slots=max(2, S), max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)


def child(d):
    """GPU side: map, frozen replay, overlay replay; results to files."""
    import torch  # noqa: F401  (device memory)

    from legkilo_amd import binding

    scans = np.load(os.path.join(d, "scans.npy"), mmap_mode="r")
    xs, Ps = np.load(os.path.join(d, "xs.npy")), np.load(os.path.join(d, "Ps.npy"))
    first, warm = np.load(os.path.join(d, "first.npy")), np.load(os.path.join(d, "warm.npy"))
    warm_t = list(np.load(os.path.join(d, "warm_t.npy")))
    S = len(scans)
    P = config.LEG_FUSION
    g = binding.LegKiloHip(make_cfg(S))
    bench.build_map(g, synth.Trajectory(), P, first, list(warm), warm_t)
    np.save(os.path.join(d, "blob.npy"), g.map_export())
    off, dt = synth.buckets_of(scans[0])
    dev = torch.device("cuda", 0)
    d_batch = torch.from_numpy(np.ascontiguousarray(scans).view(np.uint8).reshape(S, -1)).to(dev)
    g.batch_set_priors(xs, Ps)
    t0 = time.perf_counter()
    poses = g.batch_replay_dev(d_batch.data_ptr(), S, N_PTS, 0.0, off, dt)
    t_fr = time.perf_counter() - t0
    np.save(os.path.join(d, "poses_frozen.npy"), np.frombuffer(poses, dtype=_abi.pose_dtype()).copy())
    g.batch_set_priors(xs, Ps)
    g.batch_replay_overlay_dev(d_batch.data_ptr(), S, N_PTS, 0.0, off, dt, want_poses=False)   # sizes the pools
    g.batch_set_priors(xs, Ps)
    t0 = time.perf_counter()
    poses = g.batch_replay_overlay_dev(d_batch.data_ptr(), S, N_PTS, 0.0, off, dt)
    t_ov = time.perf_counter() - t0
    np.save(os.path.join(d, "poses_overlay.npy"), np.frombuffer(poses, dtype=_abi.pose_dtype()).copy())
    json.dump({"frozen_ms": t_fr * 1e3, "overlay_ms": t_ov * 1e3, "overlay_stats": g.overlay_stats(), "map_stats": g.map_stats()}, open(os.path.join(d, "gpu.json"), "w"))
    g.close()


_O = _BLOB = _SCANS = _XS = _PS = None


def _init(d, S):
    global _O, _BLOB, _SCANS, _XS, _PS
    import oracle_binding as ob

    _O = ob.Oracle(make_cfg(S), imu_mode_only=True)
    _O.init_process_cov_q()
    _O.set_acc_norm(9.81)
    _BLOB = np.load(os.path.join(d, "blob.npy"))
    _SCANS = np.load(os.path.join(d, "scans.npy"), mmap_mode="r")
    _XS, _PS = np.load(os.path.join(d, "xs.npy")), np.load(os.path.join(d, "Ps.npy"))
    _O.map_import(_BLOB)
    _O.set_map_insert(False)


def _row(pose):
    return [int(pose.n_buckets), int(pose.n_updates), int(pose.n_effect)] + list(pose.pos) + list(pose.rot)


def _frozen(chunk):
    out = []
    for s in chunk:
        _O.set_state(_XS[s], _PS[s])
        _O.set_times(0.0, 0.0)
        pose, _ = _O.process_scan(np.array(_SCANS[s]), 0.0, with_sort=True)
        out.append((s, _row(pose)))
    return out


def _overlay(chunk):
    out = []
    for s in chunk:
        _O.map_import(_BLOB)
        _O.set_map_insert(True)
        _O.set_state(_XS[s], _PS[s])
        _O.set_times(0.0, 0.0)
        pose, _ = _O.process_scan(np.array(_SCANS[s]), 0.0, with_sort=True)
        out.append((s, _row(pose)))
    return out


def compare(rows, dev):
    eq, dpos, drot, worst, bad = 0, 0.0, 0.0, None, []
    for s, r in rows:
        d = dev[s]
        same = (r[0], r[1], r[2]) == (int(d["n_buckets"]), int(d["n_updates"]), int(d["n_effect"]))
        eq += int(same)
        if not same:
            bad.append({"slot": int(s), "oracle": r[:3], "device": [int(d["n_buckets"]), int(d["n_updates"]), int(d["n_effect"])]})
        dp = float(np.abs(np.array(r[3:6]) - d["pos"]).max())
        dr = float(np.abs(np.array(r[6:15]) - d["rot"]).max())
        if dp > dpos:
            worst = int(s)
        dpos, drot = max(dpos, dp), max(drot, dr)
    n = len(rows)
    return {"n": n, "counts_equal": eq, "max_pos_delta_m": dpos, "max_rot_delta": drot, "worst_slot": worst, "count_mismatches": bad[:16], "tolerance_m": 1e-7,
            "ok": bool(dpos <= 1e-7 and drot <= 1e-7 and eq >= n - max(1, n // 50))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_parity_all_slots.json"))
    ap.add_argument("--child", default="")
    args = ap.parse_args()
    if args.child:
        return child(args.child)
    S = args.scans
    P = config.LEG_FUSION
    traj = synth.Trajectory()
    t0 = t_after = 5.0
    warm_t = [t0 + 3.0 * k for k in range(20)]
    ncpu = len(os.sched_getaffinity(0))
    workers = args.workers or max(1, min(64, ncpu))
    jobs = [("dense", (bench.scan_time(t_after, u), N_BUCKETS, 5005 + u, 1_000_003 + u)) for u in range(S)]
    jobs += [("first", (t0,))] + [("dense", (tb, N_BUCKETS, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
    tg = time.time()
    gen = bench.generate(jobs, workers)
    d = tempfile.mkdtemp(prefix="lk_allslots_")
    np.save(os.path.join(d, "scans.npy"), np.stack(gen[:S]))
    np.save(os.path.join(d, "first.npy"), gen[S])
    np.save(os.path.join(d, "warm.npy"), np.stack(gen[S + 1:]))
    np.save(os.path.join(d, "warm_t.npy"), np.array(warm_t))
    xs = np.stack([synth.initial_state(traj, bench.scan_time(t_after, s), P, np.random.default_rng(9009 + s), 0.02, 0.5) for s in range(S)])
    np.save(os.path.join(d, "xs.npy"), xs)
    np.save(os.path.join(d, "Ps.npy"), np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1)))
    del gen
    gen_s = time.time() - tg
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), "--child", d])
    if rc:
        sys.exit(f"GPU child failed ({rc})")
    import multiprocessing as mp

    chunks = [list(range(i, S, workers * 4)) for i in range(workers * 4)]
    chunks = [c for c in chunks if c]
    tc = time.time()
    with mp.get_context("fork").Pool(workers, initializer=_init, initargs=(d, S)) as pool:
        fr = [r for part in pool.map(_frozen, chunks) for r in part]
        t_fr = time.time() - tc
        ov = [r for part in pool.map(_overlay, chunks) for r in part]
    t_all = time.time() - tc
    pf = np.load(os.path.join(d, "poses_frozen.npy"))
    po = np.load(os.path.join(d, "poses_overlay.npy"))
    gpu = json.load(open(os.path.join(d, "gpu.json")))
    out = {"what": "every scan of the bench's config-5 batch (same seeds, poses, priors, map recipe as bench.py) replayed by the oracle on the device's map blob: "
                   "frozen map (insert off) against lk_batch_replay_dev, and insert ON on a private copy of the blob per scan against lk_batch_replay_overlay_dev",
           "scans": S, "points_per_scan": N_PTS, "buckets": N_BUCKETS, "frozen": compare(fr, pf), "overlay": compare(ov, po),
           "overlay_changes_counts_in_slots": int(np.sum(pf["n_effect"] != po["n_effect"])),
           "gpu": gpu, "oracle_workers": workers, "oracle_wall_s": {"frozen": round(t_fr, 1), "both": round(t_all, 1)}, "generate_s": round(gen_s, 1)}
    try:
        out["commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    except Exception:
        out["commit"] = None
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("frozen", "overlay", "overlay_changes_counts_in_slots", "gpu")}))
    import shutil

    shutil.rmtree(d, ignore_errors=True)
    if not (out["frozen"]["ok"] and out["overlay"]["ok"]):
        sys.exit(3)


if __name__ == "__main__":
    main()
