"""Shared scene construction and comparison helpers for the parity tests.

`Oracle` (oracle/oracle_binding.py) and `LegKiloHip` (leg-kilo_amd/binding.py) expose the same
method names, so every helper here drives either side with the same call sequence.
"""
import numpy as np

from legkilo_amd import abi, config, synth


class Frozen:
    """A trajectory frozen at time t (static sensor), for the first frame."""

    def __init__(self, tr, t):
        self.tr, self.t = tr, t

    def rot(self, tt):
        return self.tr.rot(np.full(np.shape(tt), self.t))

    def pos(self, tt):
        return self.tr.pos(np.full(np.shape(tt), self.t))


class Scene:
    def __init__(self, params=None, world_seed=1001, traj_seed=4004, **caps):
        self.P = dict(config.LEG_FUSION if params is None else params)
        self.world = synth.World(world_seed)
        self.traj = synth.Trajectory(traj_seed)
        self.caps = caps

    def cfg(self, **over):
        caps = dict(self.caps)
        caps.update(over)
        return config.make_config(self.P, **caps)


def xyz_of(pts):
    return np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)


def world_of(x36, xyz_body, P):
    """cloudLidarToWorld (KILO.cc:89-106): f64 transform, f32 result."""
    R = x36[:9].reshape(3, 3)
    p = x36[9:12]
    E = np.array(P["extrinsic_R"], float).reshape(3, 3)
    T = np.array(P["extrinsic_T"], float)
    b = xyz_body.astype(np.float64)
    return ((b @ E.T + T) @ R.T + p).astype(np.float32)


def init_filter(obj, scene, t0, P0=1e-6):
    x0 = synth.initial_state(scene.traj, t0, scene.P)
    obj.set_state(x0, P0 * np.eye(30))
    obj.init_process_cov_q()
    obj.set_acc_norm(9.81)
    obj.set_times(t0, t0)
    return x0


def first_frame(obj, scene, t0, x0, dense=0):
    """First-frame map build from a static scan at t0 (raw VLP-16 cloud, or a dense cloud)."""
    if dense:
        raw = synth.dense_scan(scene.world, Frozen(scene.traj, t0), t0, scene.P, n=dense, n_buckets=1, seed_scan=777)
    else:
        raw = synth.vlp16_scan(scene.world, Frozen(scene.traj, t0), t0, scene.P)
    xb = xyz_of(raw)
    xw = world_of(x0, xb, scene.P)
    obj.map_build(xw, xb)
    return xw, xb


def vlp_scan_input(scene, tb, k):
    """Config-1 style path input of scan k: raw -> preprocess -> voxel-grid centroid -> time sort."""
    P = scene.P
    raw = synth.vlp16_scan(scene.world, scene.traj, tb, P, seed_noise=3003 + k)
    pre = synth.preprocess_velodyne(raw, P["filter_num"], P["blind"])
    ds = synth.sort_by_time(synth.voxel_grid_centroid(pre, P["voxel_grid_resolution"]))
    return ds


def replay_vlp(obj, scene, t0, n_scans, use_kin=False, start=0, collect=None):  # noqa: C901
    """Replay n_scans 10 Hz scans through obj.process_scan; returns list of (pose, x36)."""
    out = []
    for k in range(start, start + n_scans):
        tb = t0 + 0.1 * k
        ds = vlp_scan_input(scene, tb, k)
        if use_kin:
            kins = synth.kin_stream(scene.traj, tb, tb + 0.1, scene.P, seed=3003 + k)
            pose, w = obj.process_scan(ds, tb, kins=kins, want_world=collect is not None)
        else:
            imus = synth.imu_stream(scene.traj, tb, tb + 0.1, seed=3003 + k)
            pose, w = obj.process_scan(ds, tb, imus=imus, want_world=collect is not None)
        x, _ = obj.get_state()
        out.append((pose, x.copy()))
        if collect is not None:
            collect.append(w)
    return out


def corner_clutter(rng, n_cells=60, per_cell=80, origin=(5.0, 5.0, 1.0)):
    """Points clustered at the 8 corners of random 0.25 m cells: per-axis variance 0.1125^2 > min_eigen_value
    (0.01), so layer-0 AND layer-1 nodes are non-planar and the octree is cut down to layer 2."""
    cells = rng.integers(0, 8, size=(n_cells, 3))
    pts = []
    for c in cells:
        centre = np.asarray(origin) + (c + 0.5) * 0.25
        sg = rng.choice([-1.0, 1.0], size=(per_cell, 3))
        pts.append(centre + sg * 0.1125 + rng.normal(0, 0.004, (per_cell, 3)))
    pts = np.concatenate(pts)
    rng.shuffle(pts)
    return pts


# ----------------------------------------------------------------------------- map comparison
def canon_map(blob_bytes):
    """blob -> {key tuple: nested node dict} (children by octant index)."""
    b = abi.parse_blob(blob_bytes)
    nodes, planes, blocks = b["nodes"], b["planes"], b["blocks"]

    def node(i):
        n, p = nodes[i], planes[i]
        d = dict(
            layer=int(n["layer"]), npts=int(n["npts"]), new_points=int(n["new_points"]), state=int(n["state"]),
            center=np.array(n["voxel_center"]), quater=float(n["quater_length"]),
            is_plane=bool(p["flags"] & abi.LK_PLANE_IS_PLANE), plane=p, pts=None, children={},
        )
        if n["block"] >= 0 and n["npts"] > 0:
            d["pts"] = blocks[int(n["block"])]["pts"][: int(n["npts"])]
        for o in range(8):
            c = int(n["child"][o])
            if c >= 0:
                d["children"][o] = node(c)
        return d

    return {tuple(int(v) for v in r["key"]): node(int(r["node"])) for r in b["roots"]}


def _expand21(v):
    M = np.zeros((6, 6))
    k = 0
    for r in range(6):
        for c in range(r, 6):
            M[r, c] = M[c, r] = v[k]
            k += 1
    return M


def compare_planes(pa, pb, rtol, where, ptol=1e-9):
    """ptol = tolerance on stored world points (the accumulated state delta of the run).  A plane fit divides by
    eigenvalue gaps, so normal / plane_var deltas are ptol times a conditioning factor (bounded here at 1e3/1e4)."""
    ntol = min(max(1e-7, 1e3 * ptol), 1e-3)
    vtol = min(max(rtol, 1e4 * ptol), 1e-2)
    assert np.allclose(pa["center"], pb["center"], rtol=0, atol=ptol), (where, pa["center"], pb["center"])
    s = 1.0 if np.dot(pa["normal"], pb["normal"]) > 0 else -1.0
    assert np.allclose(pa["normal"], s * pb["normal"], rtol=0, atol=ntol), (where, pa["normal"], pb["normal"])
    assert abs(pa["d"] - s * pb["d"]) <= max(1e-5, 40 * ntol) * max(1.0, abs(pa["d"])), (where, pa["d"], pb["d"])
    assert abs(pa["radius"] - pb["radius"]) <= 1e-5 * max(1.0, abs(pa["radius"])), where
    assert pa["points_size"] == pb["points_size"], where
    A, B = _expand21(pa["plane_var"]), _expand21(pb["plane_var"])
    B[:3, 3:] *= s
    B[3:, :3] *= s
    scale = np.abs(A).max() + 1e-300
    assert np.abs(A - B).max() <= vtol * scale, (where, np.abs(A - B).max() / scale)


def compare_nodes(a, b, where, rtol=1e-6, stats=None, ptol=1e-9):
    for f in ("layer", "npts", "new_points", "is_plane"):
        assert a[f] == b[f], (where, f, a[f], b[f])
    keep = abi.LK_NODE_INIT_OCTO | abi.LK_NODE_UPDATE_ENABLE | abi.LK_NODE_OCTO_STATE | abi.LK_NODE_PTS_DROPPED
    assert (a["state"] & keep) == (b["state"] & keep), (where, "state", a["state"], b["state"])
    assert np.array_equal(a["center"], b["center"]) and a["quater"] == b["quater"], (where, "geometry")
    if a["is_plane"]:
        compare_planes(a["plane"], b["plane"], rtol, where, ptol)
    assert (a["pts"] is None) == (b["pts"] is None), (where, "points presence", a["npts"], a["state"], b["state"])
    if a["pts"] is not None:
        dpw = np.abs(a["pts"]["pw"] - b["pts"]["pw"]).max()
        assert dpw <= ptol, (where, "pw", dpw)
        sc = np.abs(a["pts"]["var"]).max() + 1e-300
        dvar = np.abs(a["pts"]["var"] - b["pts"]["var"]).max() / sc
        assert dvar <= max(1e-7, 100 * ptol), (where, "var", dvar)
    assert set(a["children"]) == set(b["children"]), (where, "children", set(a["children"]), set(b["children"]))
    if stats is not None:
        stats["nodes"] = stats.get("nodes", 0) + 1
        stats["planes"] = stats.get("planes", 0) + int(a["is_plane"])
    for o in a["children"]:
        compare_nodes(a["children"][o], b["children"][o], where + (o,), rtol, stats, ptol)


def compare_maps(blob_a, blob_b, rtol=1e-6, ptol=1e-9):
    A, B = canon_map(blob_a), canon_map(blob_b)
    assert set(A) == set(B), ("root key sets differ", len(A), len(B), list(set(A) ^ set(B))[:5])
    stats = {}
    for k in A:
        compare_nodes(A[k], B[k], (k,), rtol, stats, ptol)
    stats["roots"] = len(A)
    return stats


def maps_identical(blob_a, blob_b):
    """Two exports hold the SAME map bit for bit (node ids may differ: roots are created by racing threads): same voxels, same
    tree shape, counters, state bits, plane records and stored points."""
    A, B = canon_map(blob_a), canon_map(blob_b)
    assert set(A) == set(B), ("root key sets differ", len(A), len(B))

    def same(a, b, where):
        for f in ("layer", "npts", "new_points", "is_plane", "state", "quater"):
            assert a[f] == b[f], (where, f, a[f], b[f])
        assert np.array_equal(a["center"], b["center"]), (where, "center")
        if a["is_plane"]:
            for f in ("center", "normal", "d", "radius", "flags", "points_size", "plane_var", "min_ev", "mid_ev", "max_ev"):
                assert np.array_equal(a["plane"][f], b["plane"][f]), (where, "plane", f)
        assert (a["pts"] is None) == (b["pts"] is None), (where, "points presence")
        if a["pts"] is not None:
            assert np.array_equal(a["pts"]["pw"], b["pts"]["pw"]) and np.array_equal(a["pts"]["var"], b["pts"]["var"]), (where, "points")
        assert set(a["children"]) == set(b["children"]), (where, "children")
        for o in a["children"]:
            same(a["children"][o], b["children"][o], where + (o,))

    for k in A:
        same(A[k], B[k], (k,))
    return len(A)


def subtree_sig(node):
    """Hashable digest of a canon_map node and everything below it: counters, state bits, stored points, tree shape."""
    return (node["layer"], node["npts"], node["new_points"], node["state"], node["is_plane"],
            None if node["pts"] is None else node["pts"].tobytes(),
            bytes(node["plane"]["plane_var"].tobytes()) if node["is_plane"] else None,
            tuple((k, subtree_sig(v)) for k, v in sorted(node["children"].items())))


def compare_overlay(overlay_blob, base_canon, after_canon, where, rtol=1e-6, ptol=1e-9):
    """One scan's insert overlay (lk_overlay_export) against the map a checker holds AFTER replaying that scan alone on a private
    copy of the base map: every private voxel equals the checker's voxel of that key (tree shape, counters, state bits exactly;
    planes and points to the tolerances of compare_nodes), and every voxel the checker changed or created is private."""
    ov = canon_map(overlay_blob)
    stats = {}
    for key, node in ov.items():
        assert key in after_canon, (where, "private voxel the checker does not have", key)
        compare_nodes(node, after_canon[key], (where, key), rtol, stats, ptol)
    changed = 0
    for key, node in after_canon.items():
        if key in base_canon and subtree_sig(node) == subtree_sig(base_canon[key]):
            continue
        changed += 1
        assert key in ov, (where, "voxel changed by the checker's insert but not private on the device", key)
    stats["private_roots"] = len(ov)
    stats["changed_roots"] = changed
    return stats


def rows_close(h6a, za, Ra, h6b, zb, Rb, valid, rtol=1e-9):
    """Compare observation rows up to the per-row sign of the plane normal."""
    v = valid.astype(bool)
    s = np.sign(np.sum(h6a[v, 3:] * h6b[v, 3:], axis=1))
    ha, hb = h6a[v], h6b[v] * s[:, None]
    assert np.allclose(ha, hb, rtol=rtol, atol=1e-9 * max(1.0, np.abs(ha).max())), np.abs(ha - hb).max()
    assert np.allclose(za[v], zb[v] * s, rtol=1e-6, atol=1e-7), np.abs(za[v] - zb[v] * s).max()
    assert np.allclose(Ra[v], Rb[v], rtol=1e-7, atol=0), np.abs(Ra[v] / Rb[v] - 1).max()


def ate(pos_a, pos_b):
    d = np.asarray(pos_a) - np.asarray(pos_b)
    return float(np.sqrt((d ** 2).sum(1).mean()))
