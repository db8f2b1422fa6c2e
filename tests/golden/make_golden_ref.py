#!/usr/bin/env python
"""Generates tests/golden/ref_kilo_small.npz with the REFERENCE ITSELF: oracle/_ref = the reference's own eskf.cc,
voxel_map.cc and KILO.cc compiled unmodified against oracle/shim (oracle/Makefile, target `ref`), driven through
KILO::process by oracle_binding.ReferenceKilo.

Runs only where /root/reference exists (this container).  The file is self-contained - inputs AND the reference's
outputs - so the oracle (CPU) and the HIP path (GPU box, no reference tree) can be checked against what the reference's
code computed:  python tests/golden/make_golden_ref.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import lk_pkg  # noqa: E402

lk_pkg.load()
import oracle_binding as ob  # noqa: E402
import scenes  # noqa: E402
from legkilo_amd import config, synth  # noqa: E402

CAPS = dict(max_roots=1 << 14, max_nodes=1 << 15, max_point_blocks=1 << 14, max_scan_points=1 << 15)


def run(mode, n_scans, tmp):
    use_kin = mode == "kin"
    sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3) if use_kin else None, **CAPS)
    k = ob.ReferenceKilo(sc.P, not use_kin, os.path.join(tmp, f"{mode}.yaml"))
    t0 = 1.0
    x0 = scenes.init_filter(k, sc, t0)
    raw = synth.vlp16_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, sc.P)[::4]
    xb = scenes.xyz_of(raw)
    xw = scenes.world_of(x0, xb, sc.P)
    k.map_build(xw, xb)
    out = {f"{mode}_x0": x0, f"{mode}_t0": t0, f"{mode}_build_world": xw, f"{mode}_build_body": xb}
    pts, aux, xs, ne = [], [], [], []
    for s in range(n_scans):
        tb = t0 + 0.1 * s
        ds = scenes.vlp_scan_input(sc, tb, s)
        a = synth.kin_stream(sc.traj, tb, tb + 0.1, sc.P, seed=3003 + s) if use_kin else synth.imu_stream(sc.traj, tb, tb + 0.1, seed=3003 + s)
        pose, _ = k.process_scan(ds, tb, kins=a) if use_kin else k.process_scan(ds, tb, imus=a)
        x, _ = k.get_state()
        pts.append(ds), aux.append(a), xs.append(x.copy()), ne.append(pose.n_effect)
    _, P = k.get_state()
    out.update({
        f"{mode}_pts": np.concatenate(pts), f"{mode}_len": np.array([len(p) for p in pts]),
        f"{mode}_aux": np.concatenate(aux), f"{mode}_aux_len": np.array([len(a) for a in aux]),
        f"{mode}_tb": np.array([t0 + 0.1 * s for s in range(n_scans)]), f"{mode}_x": np.array(xs),
        f"{mode}_n_effect": np.array(ne, dtype=np.int64), f"{mode}_P": P, f"{mode}_times": np.array(k.get_times()),
    })
    if mode == "imu":  # one map is enough to pin the insert path; it dominates the file size
        out[f"{mode}_map_blob"] = np.asarray(k.map_export(), dtype=np.uint8)
    k.close()
    return out


def main():
    assert ob.build_ref() is not None and os.path.exists("/root/reference"), "needs the reference tree"
    with tempfile.TemporaryDirectory() as tmp:
        d = run("imu", 3, tmp)
        d.update(run("kin", 2, tmp))
    path = os.path.join(HERE, "ref_kilo_small.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path), "bytes; n_effect imu", d["imu_n_effect"], "kin", d["kin_n_effect"])


if __name__ == "__main__":
    main()
