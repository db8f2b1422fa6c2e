"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the two steps just before the path:

  pcl::VoxelGrid<PointXYZINormal>::filter as used at KILO.cc:356-360 (PCL 1.8, third-party, absent here;
      voxel_grid.hpp applyFilter: inverse_leaf_size = 1/leaf in float, ijk = floor(p * inverse_leaf_size) - min_b
      in float arithmetic, idx = ijk0 + ijk1*div0 + ijk2*div0*div1, points grouped by idx, centroid of ALL
      fields incl. `curvature` accumulated in float, divided by the count)
  std::sort by curvature at KILO.cc:369-370.

PARITY: decode() below is PINNED against the reference's own lidar_processing.cc (compiled into oracle/_ref with a
restated pcl::fromROSMsg; tests/test_reference_pin.py::test_decode_matches_the_reference: points bit for bit).  The
voxel-grid filter stays UNPINNED by the reference - it is PCL's code, not the reference's: PCL's result order inside a cell (std::sort of (idx, point) pairs is unstable) and its output
order are not defined by the reference; this restatement fixes them: points of a cell are summed sequentially in
input order (float32), cells are emitted in ascending idx, the time sort is stable.
"""
import numpy as np


def voxel_grid_centroid(pts, leaf):
    x = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = np.floor(x.min(0) * inv).astype(np.int64)
    mx = np.floor(x.max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = np.floor(x * inv).astype(np.int64) - mn
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    ids = idx[order]
    starts = np.flatnonzero(np.r_[True, ids[1:] != ids[:-1]])
    cnt = np.diff(np.r_[starts, len(ids)])
    out = np.zeros(len(starts), dtype=pts.dtype)
    for name in ("x", "y", "z", "curvature"):
        v = pts[name][order].astype(np.float32)
        acc = np.zeros(len(starts), dtype=np.float32)
        for k in range(int(cnt.max())):  # k-th point of every cell that has one: sequential float32 sums per cell
            m = cnt > k
            acc[m] = acc[m] + v[starts[m] + k]
        out[name] = acc / cnt.astype(np.float32)
    return out


def sort_by_time(pts):
    return pts[np.argsort(pts["curvature"], kind="stable")]


def preprocess(pts, leaf):
    return sort_by_time(voxel_grid_centroid(pts, leaf))


def voxel_grid_centroid_loops(pts, leaf):
    """Same definition with plain Python loops (small inputs only) — cross-check of the vectorised form."""
    inv = np.float32(1.0) / np.float32(leaf)
    xs = [np.array([p["x"], p["y"], p["z"]], dtype=np.float32) for p in pts]
    mn = np.floor(np.min(xs, axis=0) * inv).astype(np.int64)
    mx = np.floor(np.max(xs, axis=0) * inv).astype(np.int64)
    div = mx - mn + 1
    cells = {}
    for p, x in zip(pts, xs):
        ijk = np.floor(x * inv).astype(np.int64) - mn
        cells.setdefault(int(ijk[0] + ijk[1] * div[0] + ijk[2] * div[0] * div[1]), []).append(p)
    out = np.zeros(len(cells), dtype=pts.dtype)
    for o, key in enumerate(sorted(cells)):
        for name in ("x", "y", "z", "curvature"):
            acc = np.float32(0)
            for p in cells[key]:
                acc = np.float32(acc + np.float32(p[name]))
            out[name][o] = acc / np.float32(len(cells[key]))
    return out


# ------------------------------------------------------------------ sensor decode (lidar_processing.cc:25-108)
VELODYNE_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "time", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<f4", "<u2"],
                           "offsets": [0, 4, 8, 12, 16, 20], "itemsize": 22})
OUSTER_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "t", "reflectivity", "ring", "ambient", "range"],
                         "formats": ["<f4", "<f4", "<f4", "<f4", "<u4", "<u2", "u1", "<u2", "<u4"],
                         "offsets": [0, 4, 8, 16, 20, 24, 26, 28, 32], "itemsize": 48})
HESAI_DTYPE = np.dtype({"names": ["x", "y", "z", "intensity", "timestamp", "ring"], "formats": ["<f4", "<f4", "<f4", "<f4", "<f8", "<u2"],
                        "offsets": [0, 4, 8, 16, 24, 32], "itemsize": 40})
OUT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("curvature", "<f4")])


def decode(raw, lidar_type, time_scale, filter_num, blind, header_stamp=0.0):
    """velodyneHandler (:25-52) / ousterHander (:54-80) / hesaiHandler (:82-108): plain-loop restatement."""
    tname = {1: "time", 2: "t", 3: "timestamp"}[lidar_type]
    n = len(raw)
    blind = np.float32(blind)
    out = []
    if lidar_type == 3:
        first = np.float64(time_scale) * np.float64(raw[tname][0])
        last = np.float64(time_scale) * np.float64(raw[tname][-1])
        begin, end = float(first), float(last)
    else:
        first = np.float32(np.float64(time_scale) * np.float64(raw[tname][0]))
        last = np.float32(np.float64(time_scale) * np.float64(raw[tname][-1]))
        begin, end = header_stamp + float(first), header_stamp + float(last)
    for i in range(n):
        x, y, z = np.float32(raw["x"][i]), np.float32(raw["y"][i]), np.float32(raw["z"][i])
        if (i % filter_num) or (blind * blind > x * x + y * y + z * z):
            continue
        if lidar_type == 3:
            cur = np.float64(time_scale) * np.float64(raw[tname][i])
            v = (cur - first) * np.float64(np.float32(500.0))
            curv = np.float32(np.floor(np.abs(v) + 0.5) * np.sign(v) / np.float64(np.float32(500.0)))  # std::round: half away from zero
        else:
            cur = np.float32(np.float64(time_scale) * np.float64(raw[tname][i]))
            v = np.float32(np.float32(cur - first) * np.float32(500.0))
            # std::round: half away from zero (np.round is half-to-even); |v| + 0.5 is exact in float64
            r = np.float32(np.floor(np.abs(np.float64(v)) + 0.5) * np.sign(np.float64(v)))
            curv = np.float32(r / np.float32(500.0))
        out.append((x, y, z, curv))
    return np.array(out, dtype=OUT_DTYPE), begin, end


def decode_vec(raw, lidar_type, time_scale, filter_num, blind, header_stamp=0.0):
    """decode() for the Velodyne / Ouster handlers (float arithmetic, lidar_processing.cc:25-80) as array expressions - the same
    float32 / float64 steps in the same order; tests/test_preprocess.py checks it against the loop version.  Needed where the
    loop is too slow (a 65 536-point Ouster scan per 0.1 s of a config-4 run)."""
    assert lidar_type in (1, 2)
    tname = {1: "time", 2: "t"}[lidar_type]
    blind = np.float32(blind)
    tt = (np.float64(time_scale) * raw[tname].astype(np.float64)).astype(np.float32)
    first, last = tt[0], tt[-1]
    x, y, z = raw["x"].astype(np.float32), raw["y"].astype(np.float32), raw["z"].astype(np.float32)
    keep = (np.arange(len(raw)) % filter_num == 0) & ~(blind * blind > x * x + y * y + z * z)
    v = ((tt - first).astype(np.float32) * np.float32(500.0)).astype(np.float32)
    r = (np.floor(np.abs(v.astype(np.float64)) + 0.5) * np.sign(v.astype(np.float64))).astype(np.float32)
    out = np.zeros(int(keep.sum()), dtype=OUT_DTYPE)
    out["x"], out["y"], out["z"] = x[keep], y[keep], z[keep]
    out["curvature"] = (r / np.float32(500.0)).astype(np.float32)[keep]
    return out, header_stamp + float(first), header_stamp + float(last)
