#!/usr/bin/env python
"""CPU-only (oracle): how far does ANOTHER LEGAL outcome of KILO.cc:369's unstable std::sort move the result?
The reference sorts a scan's points by curvature (= time offset) with std::sort; points of one time bucket compare equal, so any
permutation INSIDE a bucket is a legal result of that sort (the oracle, the reference build under oracle/_ref and the device all use
the stable instance: the input order).  Two copies of the oracle run the same scans with insert; copy B sees every bucket's points in a
random order.  Reported per scan: position / rotation difference of the posterior, difference of the matched-point counts, and at the end
the two maps compared (root keys, plane flags).  Two shapes: config-1 scans (hundreds of buckets of a dozen points) and config-3 scans
(100 000 points in 5 buckets).
Usage: sort_permutation_sensitivity.py [N_VLP_SCANS=12] [N_DENSE_SCANS=6] -> profiles/r05_sort_permutation.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lk_pkg  # noqa: E402

lk_pkg.load()
import oracle_binding as oracle_lib  # noqa: E402
import scenes  # noqa: E402
from legkilo_amd import abi, synth  # noqa: E402

oracle_lib.build()
n_vlp = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n_dense = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def permute_in_buckets(pts, rng):
    out = pts.copy()
    c = pts["curvature"]
    i = 0
    n = len(pts)
    while i < n:
        j = i + 1
        while j < n and c[j] == c[i]:
            j += 1
        if j - i > 1:
            out[i:j] = pts[i:j][rng.permutation(j - i)]
        i = j
    return out


def rot_angle(Ra, Rb):
    M = Ra.reshape(3, 3).T @ Rb.reshape(3, 3)
    return float(np.arccos(np.clip((np.trace(M) - 1.0) / 2.0, -1.0, 1.0)))


def map_summary(blob):
    b = abi.parse_blob(blob)
    cm = scenes.canon_map(blob)
    return set(cm), int((b["planes"]["flags"] & abi.LK_PLANE_IS_PLANE).astype(bool).sum()), len(b["nodes"])


def run(kind, n_scans, seed):
    scene = scenes.Scene()
    rng = np.random.default_rng(seed)
    a = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    b = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    t0 = 41.0
    for obj in (a, b):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, **({"dense": 60000} if kind == "dense" else {}))
    rows = []
    for k in range(n_scans):
        tb = t0 + 0.1 * k
        if kind == "vlp":
            pts = scenes.vlp_scan_input(scene, tb, k)
            kw = dict(imus=synth.imu_stream(scene.traj, tb, tb + 0.1, seed=3003 + k))
        else:
            pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=9100 + k, seed_noise=9200 + k)
            kw = {}
        pa, _ = a.process_scan(pts, tb, **kw)
        pb, _ = b.process_scan(permute_in_buckets(pts, rng), tb, **kw)
        xa, _ = a.get_state()
        xb, _ = b.get_state()
        rows.append({"scan": k, "points": int(len(pts)), "buckets": int(pa.n_buckets), "pos_diff_m": float(np.abs(xa[9:12] - xb[9:12]).max()),
                     "rot_diff_rad": rot_angle(xa[:9], xb[:9]), "n_effect_stable": int(pa.n_effect), "n_effect_permuted": int(pb.n_effect)})
        print(kind, rows[-1], flush=True)
    ka, pla, na = map_summary(a.map_export())
    kb, plb, nb_ = map_summary(b.map_export())
    out = {"scans": rows, "max_pos_diff_m": max(r["pos_diff_m"] for r in rows), "max_rot_diff_rad": max(r["rot_diff_rad"] for r in rows),
           "max_abs_n_effect_diff": max(abs(r["n_effect_stable"] - r["n_effect_permuted"]) for r in rows),
           "map": {"root_keys_stable": len(ka), "root_keys_permuted": len(kb), "root_keys_only_in_one": len(ka ^ kb), "nodes": [na, nb_],
                   "plane_nodes": [pla, plb]}}
    a.close()
    b.close()
    return out


res = {"what": "oracle vs oracle, copy B with every time bucket's points randomly permuted (a legal outcome of KILO.cc:369's std::sort); closed loop with insert",
       "config1_vlp": run("vlp", n_vlp, 777), "config3_dense_5x20k": run("dense", n_dense, 778)}
with open(os.path.join(ROOT, "profiles", "r05_sort_permutation.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps({k: {q: v[q] for q in ("max_pos_diff_m", "max_rot_diff_rad", "max_abs_n_effect_diff", "map")} for k, v in res.items() if isinstance(v, dict)}, indent=1))
