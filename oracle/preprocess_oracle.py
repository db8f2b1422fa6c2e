"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of the two steps just before the path:

  pcl::VoxelGrid<PointXYZINormal>::filter as used at KILO.cc:356-360 (PCL 1.8, third-party, absent here;
      voxel_grid.hpp applyFilter: inverse_leaf_size = 1/leaf in float, ijk = floor(p * inverse_leaf_size) - min_b
      in float arithmetic, idx = ijk0 + ijk1*div0 + ijk2*div0*div1, points grouped by idx, centroid of ALL
      fields incl. `curvature` accumulated in float, divided by the count)
  std::sort by curvature at KILO.cc:369-370.

PARITY UNPINNED: PCL's result order inside a cell (std::sort of (idx, point) pairs is unstable) and its output
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
