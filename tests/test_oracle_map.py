"""Pins the oracle's voxel map (oracle/oracle_voxel_map.cc) against numpy re-derivations of
voxel_map.cc:42-117 / :363-427 and checks the insert state machine (:185-241) and the parity traps of
SURVEY.md 8a on hand-built cases."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes
from legkilo_amd import abi, config


def expand21(v):
    M = np.zeros((6, 6))
    k = 0
    for r in range(6):
        for c in range(r, 6):
            M[r, c] = M[c, r] = v[k]
            k += 1
    return M


def init_plane_numpy(pw, var):
    """voxel_map.cc:42-117 with LAPACK eigh instead of EigenSolver."""
    n = len(pw)
    c = pw.mean(0)
    cov = (pw.T @ pw) / n - np.outer(c, c)
    w, V = np.linalg.eigh(cov)  # ascending
    imin, imid, imax = 0, 1, 2
    PV = np.zeros((6, 6))
    for i in range(n):
        F = np.zeros((3, 3))
        for m in (imid, imax):
            F[m] = (pw[i] - c) / (n * (w[imin] - w[m])) @ (np.outer(V[:, m], V[:, imin]) + np.outer(V[:, imin], V[:, m]))
        J = np.vstack([V @ F, np.eye(3) / n])
        PV += J @ var[i] @ J.T
    return c, w, V[:, imin], PV


def planar_cloud(rng, n, noise=0.01):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    uv = rng.uniform(-0.25, 0.25, (n, 2))
    pts = uv @ q[:, :2].T + rng.normal(0, noise, (n, 1)) * q[:, 2] + rng.normal(size=3) * 5
    A = rng.normal(size=(n, 3, 3)) * 0.01
    var = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    return pts, var


def test_init_plane_matches_numpy():
    rng = np.random.default_rng(0)
    for n in (6, 11, 50, 300):
        pw, var = planar_cloud(rng, n)
        rec, pv = ob.init_plane(pw, var.reshape(-1, 9))
        c, w, nrm, PV = init_plane_numpy(pw, var)
        assert rec["flags"] & abi.LK_PLANE_IS_PLANE
        assert np.allclose(rec["center"], c, atol=1e-12)
        s = np.sign(np.dot(rec["normal"], nrm))
        assert np.allclose(rec["normal"], s * nrm, atol=1e-9)
        assert np.isclose(rec["d"], np.float32(-(rec["normal"] @ c)), rtol=1e-6)
        assert np.isclose(rec["radius"], np.float32(np.sqrt(w[2])), rtol=1e-6)
        assert np.isclose(rec["min_ev"], np.float32(w[0]), rtol=1e-4) and rec["points_size"] == n
        PVs = PV.copy()
        PVs[:3, 3:] *= s
        PVs[3:, :3] *= s
        assert np.abs(pv - PVs).max() <= 1e-7 * np.abs(PV).max(), np.abs(pv - PVs).max() / np.abs(PV).max()
        assert np.allclose(expand21(rec["plane_var"]), pv, rtol=1e-12, atol=1e-300)  # packed upper triangle == full matrix


def test_init_plane_rejects_non_planar():
    rng = np.random.default_rng(1)
    pw = rng.uniform(-0.25, 0.25, (40, 3)) * [1, 1, 1]  # variance 0.0208 per axis > min_eigen_value 0.01
    rec, _ = ob.init_plane(pw, np.tile(np.eye(3).reshape(-1) * 1e-4, (40, 1)))
    assert not (rec["flags"] & abi.LK_PLANE_IS_PLANE)


def test_sigma_is_eigenvector_sign_invariant():
    """every downstream quantity is invariant to the sign of the plane normal."""
    rng = np.random.default_rng(2)
    pw, var = planar_cloud(rng, 30)
    _, pv = ob.init_plane(pw, var.reshape(-1, 9))
    rec, _ = ob.init_plane(pw, var.reshape(-1, 9))
    p = pw[0] + rng.normal(0, 0.02, 3)
    for s in (1.0, -1.0):
        nrm = s * rec["normal"]
        PVs = pv.copy()
        PVs[:3, 3:] *= s
        PVs[3:, :3] *= s
        J = np.r_[p - rec["center"], -nrm]
        sig = J @ PVs @ J
        if s > 0:
            ref = sig
    assert np.isclose(sig, ref, rtol=1e-12)


@pytest.fixture()
def o():
    h = ob.Oracle(config.make_config())
    yield h
    h.close()


def node_of(o, key):
    return scenes.canon_map(o.map_export())[key]


def test_insert_state_machine_plane_path(o):
    """voxel_map.cc:185-204: un-initialised until the 6th point, refit every 6th new point, freeze at 50."""
    rng = np.random.default_rng(3)
    base = np.array([0.25, 0.25, 0.25])
    var = np.tile((np.eye(3) * 1e-4).reshape(-1), (1, 1))
    seen = []
    for i in range(60):
        p = base + np.r_[rng.uniform(-0.2, 0.2, 2), rng.normal(0, 0.005)]
        o.map_update(p[None, :], var)
        n = node_of(o, (0, 0, 0))
        seen.append((n["npts"], n["new_points"], n["state"] & 7, n["is_plane"], int(n["plane"]["points_size"])))
    INIT, UPD = abi.LK_NODE_INIT_OCTO, abi.LK_NODE_UPDATE_ENABLE
    assert seen[4] == (5, 5, UPD, False, 0)            # 5 points: not initialised
    assert seen[5] == (6, 0, INIT | UPD, True, 6)      # 6th point: init_octo_tree -> plane
    assert seen[10] == (11, 5, INIT | UPD, True, 6)    # 5 new points: no refit yet
    assert seen[11] == (12, 0, INIT | UPD, True, 12)   # 6th new point: refit over 12
    assert seen[48][0] == 49 and seen[48][2] == INIT | UPD
    assert seen[49] == (0, 0, INIT, True, 48)          # size >= 50: frozen, points released, last refit at 48
    assert seen[59] == seen[49]                        # frozen plane ignores further points


def test_insert_cut_routes_by_strict_greater(o):
    """6 points at the corners of a 0.5 m voxel are not planar (lambda_min = 0.04/3 > 0.01): init_octo_tree cuts
    the root (voxel_map.cc:139-161); octant = 4*(x>cx) + 2*(y>cy) + (z>cz) with a STRICT '>' (a coordinate equal
    to the centre goes to the lower octant); children with <= 5 points stay un-initialised."""
    sg = np.array([[1, 1, 1], [-1, -1, -1], [1, -1, 1], [-1, 1, -1], [1, 1, -1], [-1, -1, 1]], float)
    pts = 0.25 + 0.24 * sg
    pts[5, 2] = 0.25  # exactly the centre z: NOT '>' -> z bit 0
    var = np.tile((np.eye(3) * 1e-5).reshape(-1), (6, 1))
    o.map_update(pts, var)
    root = node_of(o, (0, 0, 0))
    assert not root["is_plane"] and root["pts"] is None and root["npts"] == 6  # dead after the cut
    assert root["state"] & abi.LK_NODE_INIT_OCTO and root["state"] & abi.LK_NODE_OCTO_STATE
    assert sorted(root["children"]) == [0, 2, 5, 6, 7]  # (-,-,-) and (-,-,centre) share octant 0
    ch = root["children"]
    assert ch[0]["npts"] == 2 and np.array_equal(ch[0]["pts"]["pw"], pts[[1, 5]])  # input order kept
    for k, c in ch.items():
        assert c["layer"] == 1 and not (c["state"] & abi.LK_NODE_INIT_OCTO) and c["new_points"] == c["npts"]
        assert c["quater"] == np.float32(0.0625)
        exp = 0.25 + 0.125 * (np.array([(k >> 2) & 1, (k >> 1) & 1, k & 1]) * 2 - 1)
        assert np.array_equal(c["center"], exp)


def test_max_layer_leaf_freezes_above_50(o):
    """voxel_map.cc:224-237: a non-planar leaf at max_layer keeps refitting and freezes when size > 50 (not >=)."""
    sc = config.make_config()
    sc.max_layer = 0  # the root itself is the max layer
    h = ob.Oracle(sc)
    rng = np.random.default_rng(6)
    pts = rng.uniform(0.02, 0.48, (60, 3))  # non-planar at every refit
    var = np.tile((np.eye(3) * 1e-5).reshape(-1), (1, 1))
    hist = []
    for p in pts:
        h.map_update(p[None, :], var)
        n = node_of(h, (0, 0, 0))
        hist.append((n["npts"], bool(n["state"] & abi.LK_NODE_UPDATE_ENABLE), n["is_plane"]))
    assert hist[49] == (50, True, False) and hist[50] == (0, False, False) and hist[59] == (0, False, False)
    h.close()


def test_key_quirks():
    """(int)-after-minus-one float key on the residual side vs floor() on the insert side."""
    def key_trunc(p, vs=0.5):
        out = []
        for v in p:
            l = np.float32(v / vs)
            if l < 0:
                l = np.float32(np.float64(l) - 1.0)
            out.append(int(l))
        return tuple(out)

    def key_floor(p, vs=0.5):
        return tuple(int(np.floor(v / np.float64(np.float32(vs)))) for v in p)

    assert key_trunc([0.3, -0.3, 1.7]) == key_floor([0.3, -0.3, 1.7]) == (0, -1, 3)
    # exactly on a negative voxel boundary the two keys differ (the float path gives one voxel lower)
    assert key_trunc([-1.0, 0.0, 0.0]) == (-3, 0, 0) and key_floor([-1.0, 0.0, 0.0]) == (-2, 0, 0)


def test_residual_gates_and_neighbour_retry(o):
    """A plane in voxel (2,2,0); a query point whose own voxel (2,2,1) exists but holds no plane only matches
    through the one-neighbour retry of KILO.cc:156-178 when the unit-mismatch rule happens to pick (.,.,-1)..."""
    rng = np.random.default_rng(5)
    # floor plane z ~ 0.48 inside voxel (2,2,0); 60 points -> frozen plane
    pl = np.c_[rng.uniform(1.02, 1.48, 60), rng.uniform(1.02, 1.48, 60), 0.48 + rng.normal(0, 0.003, 60)]
    var = np.tile((np.eye(3) * 1e-5).reshape(-1), (len(pl), 1))
    o.map_update(pl, var)
    # make voxel (2,2,1) exist with too few points for a plane
    o.map_update(np.array([[1.25, 1.25, 0.75]]), var[:1])
    x = np.zeros(36)
    x[[0, 4, 8]] = 1.0
    o.set_state(x, 1e-8 * np.eye(30))
    T = np.array(config.LEG_FUSION["extrinsic_T"])
    inside = np.array([[1.25, 1.25, 0.47]]) - T    # in voxel (2,2,0): direct match
    above = np.array([[1.25, 1.25, 0.51]]) - T     # in voxel (2,2,1): root has no plane -> neighbour retry
    h, z, R, v = o.residuals(np.r_[inside, above].astype(np.float32))
    assert v[0] == 1 and abs(abs(z[0]) - 0.01) < 5e-3 and abs(abs(h[0, 5]) - 1) < 1e-3
    # loc = (2.5, 2.5, 1.02) voxel units vs centre +- quater in METRES (1.25 +- 0.125, 0.75 +- 0.125):
    # x: 2.5 > 1.375 -> +1 ; y: +1 ; z: 1.02 > 0.875 -> +1  => neighbour (3,3,2) does not exist -> no match
    assert v[1] == 0
    assert R[0] > 0


def test_map_sliding_semantics(o):
    """voxel_map.cc:552-594: threshold on the travelled distance since the last slide (initially the origin), box in
    KEY space centred on floor(position / max_voxel_size), strict comparisons, whole root voxels deleted."""
    vs = 0.5
    keys = [(x, 0, 0) for x in range(-8, 9)] + [(0, y, 0) for y in (-5, -4, 4, 5)] + [(0, 0, 3), (0, 0, -3)]
    var = (np.eye(3) * 1e-4).reshape(1, 9)
    for k in keys:
        o.map_update(((np.array(k) + 0.5) * vs)[None, :], var)
    assert set(scenes.canon_map(o.map_export())) == set(keys)
    # travelled less than sliding_thresh since (0,0,0): nothing happens, last_slide_position keeps its value
    assert o.map_slide([0.9, 0.0, 0.0], sliding_thresh=1.0, half_map_size=2) == (False, 0)
    assert np.array_equal(o.get_last_slide_position(), [0.0, 0.0, 0.0])
    # exactly at the threshold: `<` is false -> it slides.  k = floor(-1.0 / 0.5) = -2: box x in [-6, 2], y, z in [-4, 4]
    slid, nrem = o.map_slide([-1.0, 0.0, 0.0], sliding_thresh=1.0, half_map_size=4)
    left = set(scenes.canon_map(o.map_export()))
    gone = set(keys) - left
    assert slid and nrem == len(gone)
    assert gone == {(-8, 0, 0), (-7, 0, 0)} | {(x, 0, 0) for x in range(3, 9)} | {(0, -5, 0), (0, 5, 0)}
    assert (-6, 0, 0) in left and (2, 0, 0) in left and (0, 4, 0) in left and (0, -4, 0) in left   # on the faces: kept
    assert np.array_equal(o.get_last_slide_position(), [-1.0, 0.0, 0.0])
    # the distance is now measured from the new last_slide_position
    assert o.map_slide([-1.5, 0.0, 0.0], sliding_thresh=1.0, half_map_size=1) == (False, 0)
    # clearMemOutOfMap on its own, asymmetric box
    n = o.map_clear_outside(1, -1, 0, 0, 3, 0)
    left2 = set(scenes.canon_map(o.map_export()))
    assert left2 == {(-1, 0, 0), (0, 0, 0), (1, 0, 0), (0, 0, 3)} and n == len(left) - len(left2)
    # a surviving voxel keeps accepting points (the tree object was not touched)
    o.map_update((np.array([0.25, 0.25, 0.25]))[None, :], var)
    assert node_of(o, (0, 0, 0))["npts"] == 2


def test_blob_import_round_trip_and_continuation():
    """lko_map_import (oracle_capi.cc) rebuilds the octrees of an exported blob: the re-export is byte-identical, and two
    more scans WITH insert through the rebuilt map give the same poses and the same map as the original handle.  This is
    what lets bench.py's checker replay against exactly the map the device holds."""
    sc = scenes.Scene()
    t0 = 5.0
    a = ob.Oracle(sc.cfg(), imu_mode_only=True)
    x0 = scenes.init_filter(a, sc, t0)
    scenes.first_frame(a, sc, t0, x0)
    scenes.replay_vlp(a, sc, t0, 2)
    blob = a.map_export()
    xa, Pa = a.get_state()
    b = ob.Oracle(sc.cfg(), imu_mode_only=True)
    b.init_process_cov_q()
    b.set_acc_norm(9.81)
    b.map_import(blob)
    assert np.array_equal(b.map_export(), blob)
    b.set_state(xa, Pa)
    b.set_times(*a.get_times())
    ra = scenes.replay_vlp(a, sc, t0, 2, start=2)
    rb = scenes.replay_vlp(b, sc, t0, 2, start=2)
    for (pa, xa_), (pb, xb_) in zip(ra, rb):
        assert (pa.n_effect, pa.n_buckets, pa.n_updates) == (pb.n_effect, pb.n_buckets, pb.n_updates)
        # the blob keeps the upper triangles of plane_var / var only: the rebuilt matrices are exactly symmetric where the
        # original accumulations are symmetric to rounding, hence last-bit differences
        assert np.abs(xa_ - xb_).max() < 1e-9, np.abs(xa_ - xb_).max()
    scenes.compare_maps(a.map_export(), b.map_export(), rtol=1e-6, ptol=1e-9)
    with pytest.raises(AssertionError):
        b.map_import(blob[:-8])
    a.close(), b.close()
