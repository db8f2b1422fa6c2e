#!/usr/bin/env python
"""Generates tests/golden/path_small.npz with the ORACLE (the reference ships no vectors of its own).
The oracle itself is pinned against the reference's own sources (oracle/_ref, tests/test_reference_pin.py);
make_golden_ref.py next to this file generates vectors with that reference build directly.

The file is self-contained: inputs (first-frame clouds, prior state, query / bucket points) AND the oracle's
outputs, so the GPU box needs neither /root/reference nor the synthetic generator to check against it.
Re-run only when the oracle changes on purpose:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lk_pkg  # noqa: E402

lk_pkg.load()
import oracle_binding as ob  # noqa: E402
import scenes  # noqa: E402
from legkilo_amd import synth  # noqa: E402


def main():
    sc = scenes.Scene(max_roots=1 << 14, max_nodes=1 << 15, max_point_blocks=1 << 14, max_scan_points=1 << 15)
    o = ob.Oracle(sc.cfg(), imu_mode_only=True)
    t0 = 1.0
    x0 = scenes.init_filter(o, sc, t0)
    raw = synth.vlp16_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, sc.P)[::4]
    xb = scenes.xyz_of(raw)
    xw = scenes.world_of(x0, xb, sc.P)
    o.map_build(xw, xb)
    # a few scans so that the map has refitted / frozen planes and octree children
    seq = []
    for k in range(4):
        tb = t0 + 0.1 * k
        ds = scenes.vlp_scan_input(sc, tb, k)
        imus = synth.imu_stream(sc.traj, tb, tb + 0.1, seed=3003 + k)
        pose, _ = o.process_scan(ds, tb, imus=imus)
        x, _ = o.get_state()
        seq.append(dict(pts=ds, imus=imus, tb=tb, x=x.copy(), n_effect=pose.n_effect, n_buckets=pose.n_buckets, n_updates=pose.n_updates))
    blob = o.map_export()
    xs, Ps = o.get_state()
    tp, tu = o.get_times()
    # residual rows at the current state (config 2 shape, small)
    q = synth.dense_scan(sc.world, scenes.Frozen(sc.traj, tp), tp, sc.P, n=3000, n_buckets=1, seed_scan=31)
    qb = scenes.xyz_of(q)
    h6, z, R, valid = o.residuals(qb)
    # one bucket through predictUpdatePoint
    bk = scenes.xyz_of(scenes.vlp_scan_input(sc, tp + 0.01, 99))[:1200]
    w, inten, ne = o.update_points(tp + 0.01, bk)
    x1, P1 = o.get_state()
    n_roots = o.map_stats()
    np.savez_compressed(
        os.path.join(HERE, "path_small.npz"),
        build_world=xw, build_body=xb, x0=x0, t0=t0,
        seq_pts=np.concatenate([s["pts"] for s in seq]), seq_len=np.array([len(s["pts"]) for s in seq]),
        seq_imus=np.concatenate([s["imus"] for s in seq]), seq_imu_len=np.array([len(s["imus"]) for s in seq]),
        seq_tb=np.array([s["tb"] for s in seq]), seq_x=np.array([s["x"] for s in seq]),
        seq_counts=np.array([[s["n_buckets"], s["n_updates"], s["n_effect"]] for s in seq]),
        map_blob=blob, xs=xs, Ps=Ps, times=np.array([tp, tu]),
        q_body=qb, q_h6=h6, q_z=z, q_R=R, q_valid=valid,
        bk_t=tp + 0.01, bk_body=bk, bk_world=w, bk_intensity=inten, bk_n_effect=ne, x1=x1, P1=P1, n_roots_after=n_roots,
    )
    sz = os.path.getsize(os.path.join(HERE, "path_small.npz"))
    print("wrote path_small.npz", sz, "bytes; valid", int(valid.sum()), "/", len(valid), "bucket n_effect", ne)


if __name__ == "__main__":
    main()
