#!/usr/bin/env python3
"""CPU figure (i) of BASELINE.md section 2 / SURVEY 8(d), MEASURED: the reference's literal N x N `updateByPoints` (eskf.cc:105-112: HPH^T + R
assembled as a dense N x N matrix and inverted) on ONE 100 000-point scan with the reference's own 51 time buckets (2 ms bins,
lidar_processing.cc:48; N ~ 1 900 matched rows per bucket), through the oracle with `literal_max_n` raised so that every bucket takes the
literal branch, one pinned core.  CPU only (no GPU, no product code on the measured path); the same scan through the 6 x 6 information form
beside it, and the two posteriors compared.  Writes profiles/r05_cpu_literal_51.json, which bench.py quotes in
cpu_baseline.reference_literal (cached: a scan takes minutes, the bench line must not).

    python tools/cpu_literal_51.py [--warm 20] [--out profiles/r05_cpu_literal_51.json]
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import bench  # noqa: E402  (generation + map building of the bench line; importing it touches no GPU)
import oracle_binding as ob  # noqa: E402
from legkilo_amd import config, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--warm", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_cpu_literal_51.json"))
    args = ap.parse_args()
    try:
        os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[-1]})
    except Exception:
        pass
    P = config.LEG_FUSION
    traj = synth.Trajectory()
    t0, t_after = 5.0, 5.0
    warm_t = [t0 + 3.0 * k for k in range(args.warm)]
    jobs = [("first", (t0,))] + [("dense", (tb, 5, 2002 + k, 3003 + k)) for k, tb in enumerate(warm_t)]
    jobs += [("dense", (t_after + 0.1, 51, 8208, 8308))]
    gen = bench.generate(jobs, 1)
    first, warm, scan = gen[0], gen[1:1 + args.warm], gen[-1]
    cfg = config.make_config(P, device_id=0, n_slots=2, max_roots=1 << 15, max_nodes=1 << 16, max_point_blocks=1 << 17, max_scan_points=1 << 17)
    res = {}
    for name, lit in (("info6", 512), ("literal", 1 << 30)):
        o = ob.Oracle(cfg, imu_mode_only=True)
        bench.build_map(o, traj, P, first, warm, warm_t)
        o.set_literal_max_n(lit)
        o.set_state(synth.initial_state(traj, t_after + 0.1, P, np.random.default_rng(9009), 0.02, 0.5), 1e-4 * np.eye(30))
        o.set_times(t_after + 0.1, t_after + 0.1)
        tc = time.perf_counter()
        pose, _ = o.process_scan(scan, t_after + 0.1, with_sort=True)
        el = time.perf_counter() - tc
        x, Pc = o.get_state()
        res[name] = dict(s=el, n_effect=int(pose.n_effect), n_buckets=int(pose.n_buckets), n_updates=int(pose.n_updates), x=np.asarray(x).ravel(), P=np.asarray(Pc).ravel())
        print(name, f"{el:.2f} s", res[name]["n_effect"], flush=True)
        o.close()
    a, b = res["literal"], res["info6"]
    out = {
        "what": "ONE 100 000-point synthetic scan (bench scene, seeds 8208 / 8308), 51 time buckets, full path with insert (KILO.cc:367-396 incl. the sort), "
                "oracle port with literal_max_n raised: every bucket's updateByPoints is the reference's literal N x N inverse (eskf.cc:105-112; the oracle's "
                "partially pivoted elimination, -O3, no BLAS) - measured, not extrapolated",
        "host": platform.processor() or platform.machine(), "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
        "cores_used": 1, "warm_scans_in_map": args.warm,
        "literal_51_buckets_s_per_scan_measured": round(a["s"], 2), "info6_51_buckets_s_per_scan_measured": round(b["s"], 4),
        "literal_over_info6": round(a["s"] / b["s"], 1),
        "n_buckets": a["n_buckets"], "n_effect_literal": a["n_effect"], "n_effect_info6": b["n_effect"], "mean_matched_rows_per_bucket": round(a["n_effect"] / max(a["n_updates"], 1), 1),
        "max_state_delta_literal_vs_info6": float(np.abs(a["x"] - b["x"]).max()), "max_cov_delta_literal_vs_info6": float(np.abs(a["P"] - b["P"]).max()),
    }
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
