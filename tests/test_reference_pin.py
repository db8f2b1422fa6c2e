"""Pins the oracle against the REFERENCE'S OWN eskf.cc / voxel_map.cc.

oracle/_ref/liblegkilo_ref.so is those two reference sources compiled unmodified from /root/reference (oracle/Makefile,
target `ref`) against stand-in headers for the absent third-party libraries (oracle/shim: an eager mini-Eigen, field-only
PCL / ROS message shells) plus the thin C API of oracle/ref_capi.cc.  Every arithmetic routine of the hot path that
lives in those two files is compared here, on the same seeded inputs, with the restatement under oracle/ that all other
parity tests use as their checker:

  ESKF (eskf.cc)          initProcessCovQ, getFx, getFunctionf, predict, State (+) / (-), updateByPoints (N = 1 and the
                          literal N x N form), updateByImu, updateByKinImu
  math (math_utils.hpp)   Exp (both thresholds), Log
  keys (eigen_types.hpp)  hash_vec<3>, voxelKeyFloor
  map (voxel_map.cc)      calcBodyCov, init_plane, BuildVoxelMap, UpdateVoxelMap / UpdateOctoTree (incl. cuts to layer 2,
                          refits, freezes), build_single_residual, mapSliding / clearMemOutOfMap

What is restated on the reference side is third-party arithmetic only (Eigen's inverse() and EigenSolver: oracle/shim
header comment); KILO.cc's glue cannot be built here and stays pinned by the oracle's own tests.  The file is skipped
where neither /root/reference nor a prebuilt oracle/_ref exists.
"""
import numpy as np
import pytest

import oracle_binding as ob
import scenes
from legkilo_amd import abi, config, synth

pytestmark = pytest.mark.skipif(ob.build_ref() is None, reason="oracle/_ref not built and /root/reference absent")


def rand_state(rng, big_rot=True):
    x = np.zeros(36)
    a, _, _ = ob.exp_log(rng.normal(0, 0.8 if big_rot else 1e-3, 3))
    x[:9] = a.reshape(9)
    x[9:] = rng.normal(0, 1.0, 27)
    x[9 + 12:9 + 15] = [0.0, 0.0, -9.81]
    return x


def rand_cov(rng, scale=1e-3):
    A = rng.normal(size=(30, 30))
    return scale * (A @ A.T) / 30 + 1e-6 * np.eye(30)


@pytest.fixture()
def pair():
    cfg = config.make_config()
    o, r = ob.Oracle(cfg), ob.Reference(cfg)
    o.set_literal_max_n(1 << 30)   # the reference only has the literal N x N form
    yield o, r
    o.close()
    r.close()


def both(pair, fn):
    return fn(pair[0]), fn(pair[1])


# ----------------------------------------------------------------------------- ESKF
def test_process_noise_fx_f_predict(pair):
    rng = np.random.default_rng(11)
    o, r = pair
    for obj in pair:
        obj.init_process_cov_q()
    assert np.array_equal(o.get_Q(), r.get_Q())
    for trial in range(20):
        x, P = rand_state(rng), rand_cov(rng)
        dt = float(rng.uniform(1e-4, 0.05))
        for obj in pair:
            obj.set_state(x, P)
        # association of scalar * matrix * matrix may differ by one rounding between the two sides
        assert np.allclose(o.get_fx(dt), r.get_fx(dt), rtol=1e-15, atol=1e-18), trial
        assert np.allclose(o.get_function_f(dt), r.get_function_f(dt), rtol=1e-15, atol=1e-18), trial
        for obj in pair:
            obj.predict(dt, True, False)
            obj.predict(dt * 1.7, False, True)
        (xo, Po), (xr, Pr) = o.get_state(), r.get_state()
        # same formulas, same operation order: equal to the last bit except for re-association inside the 30x30 products
        assert np.allclose(xo, xr, rtol=1e-15, atol=1e-15), np.abs(xo - xr).max()
        assert np.allclose(Po, Pr, rtol=1e-13, atol=1e-18), np.abs(Po - Pr).max()


def test_exp_log_and_boxminus(pair):
    rng = np.random.default_rng(12)
    for v in [np.zeros(3), np.array([3e-6, 0, 0]), np.array([2e-5, 1e-5, 0]), np.array([5e-8, 0, 0]), np.array([0.0, 1e-7, 1e-7]),
              np.array([0.3, -0.2, 0.9]), np.array([3.0, 0.1, -0.1])] + [rng.normal(0, 1, 3) for _ in range(10)]:
        ao, bo, lo = ob.exp_log(v)
        ar, br, lr = ob.exp_log(v, "ref")
        # both thresholds (1e-5 and 1e-7); (1 - cos) * K * K may associate either way (Eigen version dependent): 1 ulp
        assert np.allclose(ao, ar, rtol=0, atol=3e-16) and np.allclose(bo, br, rtol=0, atol=3e-16), v
        if np.linalg.norm(v) <= 1e-7:
            assert np.array_equal(ao, np.eye(3)) and np.array_equal(ar, np.eye(3)) and np.array_equal(br, np.eye(3))
        # Log amplifies the 1-ulp difference of R by 1 / sin(theta) (theta ~ 3 rad in one of the cases)
        assert np.allclose(lo, lr, rtol=1e-12, atol=1e-15), (v, lo, lr)
    for _ in range(10):
        xa, xb = rand_state(rng), rand_state(rng)
        assert np.allclose(ob.state_minus(xa, xb), ob.state_minus(xa, xb, "ref"), rtol=1e-14, atol=1e-16)
        xs = xa.copy()
        xs[9:] += 1e-5
        assert np.allclose(ob.state_minus(xs, xa), ob.state_minus(xs, xa, "ref"), rtol=1e-12, atol=1e-18)  # |theta| < 1e-3 branch


@pytest.mark.parametrize("N", [1, 2, 7, 40, 200])
def test_update_by_points_literal_form(pair, N):
    """eskf.cc:91-113: N == 1 scalar branch with the +1e-4, otherwise K = P H^T (H P H^T + R)^-1 with the N x N inverse.
    The two sides invert with different (both partially pivoted) eliminations: agreement to ~1e-9 relative."""
    rng = np.random.default_rng(100 + N)
    o, r = pair
    x, P = rand_state(rng), rand_cov(rng, 1e-4)
    h6 = rng.normal(0, 1, (N, 6))
    h6[:, 3:] /= np.linalg.norm(h6[:, 3:], axis=1, keepdims=True)
    z = rng.normal(0, 0.02, N)
    R = rng.uniform(1e-4, 1e-3, N)
    for obj in pair:
        obj.set_state(x, P)
        obj.update_by_points(h6, z, R)
    (xo, Po), (xr, Pr) = o.get_state(), r.get_state()
    assert np.allclose(xo, xr, rtol=1e-9, atol=1e-12), np.abs(xo - xr).max()
    assert np.allclose(Po, Pr, rtol=1e-7, atol=1e-13), np.abs(Po - Pr).max()
    assert np.abs(xo - x).max() > 1e-6   # the update did something


def test_update_by_imu_and_kin_imu(pair):
    rng = np.random.default_rng(13)
    o, r = pair
    for trial in range(5):
        x, P = rand_state(rng), rand_cov(rng, 1e-4)
        z6, R6 = rng.normal(0, 0.1, 6), rng.uniform(1e-3, 1e-2, 6)
        M = int(rng.integers(7, 19))
        kh = np.zeros((M, 30))
        kh[:, rng.integers(0, 30, M)] = 1.0
        kh += rng.normal(0, 0.2, (M, 30)) * (rng.random((M, 30)) < 0.2)
        kz, kR = rng.normal(0, 0.05, M), rng.uniform(1e-3, 1e-2, M)
        for obj in pair:
            obj.set_state(x, P)
            obj.update_by_imu(z6, R6)
            obj.update_by_kin_imu(kh, kz, kR)
        (xo, Po), (xr, Pr) = o.get_state(), r.get_state()
        assert np.allclose(xo, xr, rtol=1e-9, atol=1e-12), (trial, np.abs(xo - xr).max())
        assert np.allclose(Po, Pr, rtol=1e-7, atol=1e-13), (trial, np.abs(Po - Pr).max())


# ----------------------------------------------------------------------------- keys
def test_hash_and_floor_key():
    rng = np.random.default_rng(14)
    for k in rng.integers(-3000, 3000, (200, 3)):
        assert ob.hash_vec3(*map(int, k)) == ob.hash_vec3(*map(int, k), which="ref")
    for p in list(rng.normal(0, 20, (200, 3))) + [np.array([-1.0, 0.0, 0.5]), np.array([-0.5, -1e-12, 1e-12])]:
        for vs in (0.5, float(np.float32(0.3)), 0.3):
            assert ob.key_floor(p, vs) == ob.key_floor(p, vs, "ref")


# ----------------------------------------------------------------------------- map arithmetic
def test_calc_body_cov():
    rng = np.random.default_rng(15)
    pts = list(rng.normal(0, 10, (300, 3))) + [np.array([1.0, 2.0, 0.0]), np.array([0.0, 0.0, 0.0]), np.array([5.0, -5.0, 1e-9])]
    for pb in pts:
        co, cr = ob.calc_body_cov(pb, 0.02, 0.1), ob.calc_body_cov(pb, 0.02, 0.1, "ref")
        assert np.allclose(co, cr, rtol=1e-13, atol=1e-20), (pb, np.abs(co - cr).max())


def planar(rng, n, noise):
    R = ob.exp_log(rng.normal(0, 1, 3))[0]
    uv = rng.uniform(-0.25, 0.25, (n, 2))
    p = np.c_[uv, rng.normal(0, noise, n)] @ R.T + rng.normal(0, 3, 3)
    var = np.zeros((n, 9))
    for i in range(n):
        A = rng.normal(0, 1, (3, 3))
        var[i] = (1e-4 * (A @ A.T) / 3 + 1e-6 * np.eye(3)).reshape(9)
    return p, var


@pytest.mark.parametrize("noise", [0.002, 0.02, 0.2])
def test_init_plane(noise):
    """voxel_map.cc:42-117.  The two sides use different symmetric eigen-solvers (both Jacobi-type stand-ins for Eigen's
    EigenSolver): eigenvalues / normal agree to ~1e-12, the eigenvector derivative blows that up by 1 / eigen-gap in
    plane_var."""
    rng = np.random.default_rng(int(noise * 1e4))
    for trial in range(10):
        n = int(rng.integers(6, 50))
        p, var = planar(rng, n, noise)
        (ro, vo), (rr, vr) = ob.init_plane(p, var), ob.init_plane(p, var, which="ref")
        assert (ro["flags"] & abi.LK_PLANE_IS_PLANE) == (rr["flags"] & abi.LK_PLANE_IS_PLANE), trial
        assert ro["points_size"] == rr["points_size"] == n
        assert np.allclose(ro["center"], rr["center"], rtol=1e-14)
        if ro["flags"] & abi.LK_PLANE_IS_PLANE:
            s = np.sign(np.dot(ro["normal"], rr["normal"]))
            assert np.allclose(ro["normal"], s * rr["normal"], atol=1e-10), trial
            assert abs(ro["d"] - s * rr["d"]) <= 1e-6 * max(1.0, abs(ro["d"]))
            assert np.isclose(ro["radius"], rr["radius"], rtol=1e-6)
            for k in ("min_ev", "mid_ev", "max_ev"):
                assert np.isclose(ro[k], rr[k], rtol=1e-5, atol=1e-12), (trial, k)
            S = np.diag([1, 1, 1, s, s, s]).astype(float)   # sign of the normal flips the n-rows / n-columns
            assert np.allclose(vo, S @ vr @ S, rtol=1e-6, atol=1e-9 * np.abs(vo).max()), (trial, np.abs(vo - S @ vr @ S).max())


def test_build_and_update_voxel_map_match_the_reference():
    """BuildVoxelMap (voxel_map.cc:287-334) on a first frame, then UpdateVoxelMap (:336-361) fed point by point and in
    batches with clutter that cuts voxels down to layer 2: identical tree shape, counters, state bits, stored points."""
    sc = scenes.Scene()
    cfg = sc.cfg()
    o, r = ob.Oracle(cfg), ob.Reference(cfg)
    t0 = 1.0
    for obj in (o, r):
        x0 = scenes.init_filter(obj, sc, t0)
        scenes.first_frame(obj, sc, t0, x0)
    st = scenes.compare_maps(o.map_export(), r.map_export(), rtol=1e-6, ptol=1e-9)
    assert st["roots"] > 500
    rng = np.random.default_rng(16)
    pts = scenes.corner_clutter(rng, n_cells=40, per_cell=70)
    walls = np.c_[rng.uniform(-3, 3, 4000), np.full(4000, 7.3) + rng.normal(0, 0.01, 4000), rng.uniform(0, 2.5, 4000)]
    allp = np.concatenate([pts, walls])
    rng.shuffle(allp)
    var = np.tile((np.eye(3) * 4e-4).reshape(1, 9), (len(allp), 1))
    var[:, [1, 3]] = 1e-5
    k = 0
    for chunk in (1, 1, 1, 5, 17, 200, 1000, len(allp)):
        a, b = k, min(len(allp), k + chunk)
        if a >= b:
            break
        for obj in (o, r):
            obj.map_update(allp[a:b], var[a:b])
        k = b
        scenes.compare_maps(o.map_export(), r.map_export(), rtol=1e-6, ptol=1e-9)
    cm = scenes.canon_map(r.map_export())

    def depth(n):
        return 1 + max([depth(c) for c in n["children"].values()], default=0)

    assert max(depth(n) for n in cm.values()) >= 3, "the clutter must cut voxels down to layer 2"
    assert o.map_stats() == r.map_stats()
    # ---- build_single_residual (voxel_map.cc:363-427) on home and neighbouring voxels
    n_ok = 0
    q = np.concatenate([allp[:1500] + rng.normal(0, 0.02, (1500, 3)), rng.uniform(-4, 9, (500, 3))])
    for p in q:
        key = ob.key_floor(p, float(np.float32(0.5)))
        for dk in ((0, 0, 0), (1, 0, 0), (0, -1, 0)):
            kk = tuple(int(a + b) for a, b in zip(key, dk))
            V = (np.eye(3) * 1e-4 + 2e-5).reshape(9)
            mo, mr = o.match_voxel(kk, p, V), r.match_voxel(kk, p, V)
            assert (mo["found"], mo["success"], mo["layer"] if mo["success"] else -1) == \
                   (mr["found"], mr["success"], mr["layer"] if mr["success"] else -1), (p, kk)
            if mo["success"]:
                n_ok += 1
                s = np.sign(np.dot(mo["normal"], mr["normal"]))
                assert np.allclose(mo["normal"], s * mr["normal"], atol=1e-9)
                assert np.isclose(mo["dis_to_plane"], s * mr["dis_to_plane"], rtol=1e-5, atol=1e-7)
                assert np.isclose(mo["prob"], mr["prob"], rtol=1e-6)
    assert n_ok > 500
    # ---- sliding
    assert o.map_slide([5.2, 5.4, 1.0], 1.0, 6) == r.map_slide([5.2, 5.4, 1.0], 1.0, 6)
    assert np.array_equal(o.get_last_slide_position(), r.get_last_slide_position())
    assert o.map_slide([5.3, 5.4, 1.0], 1.0, 6) == r.map_slide([5.3, 5.4, 1.0], 1.0, 6) == (False, 0)
    assert o.map_clear_outside(12, 8, 20, 9, 4, 1) == r.map_clear_outside(12, 8, 20, 9, 4, 1)
    scenes.compare_maps(o.map_export(), r.map_export(), rtol=1e-6, ptol=1e-9)
    assert 0 < o.map_stats() == r.map_stats()
    o.close()
    r.close()


def test_dense_first_frame_map_matches_the_reference():
    """The 100 k-point first frame the benches use: ~20 k root voxels through init_octo_tree / cut_octo_tree."""
    sc = scenes.Scene()
    cfg = sc.cfg()
    o, r = ob.Oracle(cfg), ob.Reference(cfg)
    t0 = 5.0
    for obj in (o, r):
        x0 = scenes.init_filter(obj, sc, t0)
        scenes.first_frame(obj, sc, t0, x0, dense=30000)
    st = scenes.compare_maps(o.map_export(), r.map_export(), rtol=1e-6, ptol=1e-9)
    assert st["roots"] > 3000
    o.close()
    r.close()


# ----------------------------------------------------------------------------- the path itself: KILO::process
def kilo_pair(sc, imu_only, tmp_path):
    o = ob.Oracle(sc.cfg(), imu_mode_only=imu_only)
    k = ob.ReferenceKilo(sc.P, imu_only, tmp_path / "ref.yaml")
    return o, k


@pytest.mark.parametrize("use_kin", [False, True])
def test_kilo_process_matches_the_reference(tmp_path, use_kin):
    """The reference's own KILO::process (KILO.cc:316-399) - time sort, bucket loop with interleaved IMU or
    kinematic+IMU updates, predictUpdatePoint (:108-233: transform, covariances, float-truncated key, one-neighbour
    retry with its unit mismatch, observation rows, literal N x N update, re-projection, map insert) - replays the same
    config-1 style scans as the oracle: identical match counts every scan, states to 1e-8, identical map."""
    sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3) if use_kin else None)
    o, k = kilo_pair(sc, not use_kin, tmp_path)
    t0 = 1.0
    for obj in (o, k):
        x0 = scenes.init_filter(obj, sc, t0)
        scenes.first_frame(obj, sc, t0, x0)
    scenes.compare_maps(o.map_export(), k.map_export(), rtol=1e-6, ptol=1e-9)
    n_scans = 4
    ro = scenes.replay_vlp(o, sc, t0, n_scans, use_kin=use_kin)
    rk = scenes.replay_vlp(k, sc, t0, n_scans, use_kin=use_kin)
    for s, ((po, xo), (pk, xk)) in enumerate(zip(ro, rk)):
        assert po.n_effect == pk.n_effect > 500, (s, po.n_effect, pk.n_effect)
        assert np.allclose(xo, xk, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xk).max())
    (_, Po), (_, Pk) = o.get_state(), k.get_state()
    assert np.allclose(Po, Pk, rtol=1e-6, atol=1e-12), np.abs(Po - Pk).max()
    assert o.get_times() == k.get_times()
    st = scenes.compare_maps(o.map_export(), k.map_export(), rtol=1e-6, ptol=1e-7)
    assert st["roots"] > 500
    o.close()
    k.close()


@pytest.mark.parametrize("imu_only", [True, False])
def test_kilo_first_frame_initialisation(tmp_path, imu_only):
    """KILO.cc:332-353 + state_initial.hpp: gravity / gyro-bias from the IMU (or kinematic) samples of the first packet,
    P0, Q, acc_norm, both time stamps, map from the raw first cloud."""
    sc = scenes.Scene()
    o, k = kilo_pair(sc, imu_only, tmp_path)
    t0 = 2.0
    raw = synth.vlp16_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, sc.P)
    imus = synth.imu_stream(sc.traj, t0 - 0.1, t0, seed=77)
    kins = synth.kin_stream(sc.traj, t0 - 0.1, t0, sc.P, seed=77)
    for obj in (o, k):
        if imu_only:
            obj.first_frame(raw, t0, imus=imus)
        else:
            obj.first_frame(raw, t0, kins=kins)
    (xo, Po), (xk, Pk) = o.get_state(), k.get_state()
    assert np.allclose(xo, xk, rtol=1e-14, atol=1e-15), np.abs(xo - xk).max()
    assert np.array_equal(Po, Pk) and np.array_equal(o.get_Q(), k.get_Q())
    assert np.isclose(o.get_acc_norm(), k.get_acc_norm(), rtol=1e-15) and o.get_times() == k.get_times() == (t0, t0)
    scenes.compare_maps(o.map_export(), k.map_export(), rtol=1e-6, ptol=1e-9)
    o.close()
    k.close()


# ----------------------------------------------------------------------------- in front of the path: the sensor decode
@pytest.mark.parametrize("lidar_type", [1, 2, 3])
def test_decode_matches_the_reference(lidar_type):
    """The reference's own LidarProcessing::{velodyne,ouster,hesai}Handler (lidar_processing.cc:25-108: every
    filter_num-th point outside the blind radius, curvature = round((t - t_first) * 500) / 500 in the handler's own
    arithmetic type, begin / end times) on a PointCloud2 payload vs oracle/preprocess_oracle.decode - bit for bit.
    (The device decode lk_decode_scan is bit-exact against that oracle: tests/test_preprocess.py.)"""
    import preprocess_oracle as po
    import test_preprocess as tp

    sc = scenes.Scene()
    for t, stamp, fnum, blind in ((1.0, 50.0, 3, 1.5), (2.3, 1234.5, 1, 0.5), (3.1, 0.0, 4, 4.0)):
        raw, layout, scale = tp.raw_message(sc, t, lidar_type)
        want, wb, we = po.decode(raw, lidar_type, scale, fnum, blind, header_stamp=stamp)
        got, gb, ge = ob.ref_decode(raw, layout, scale, fnum, blind, header_stamp=stamp)
        assert len(got) == len(want) > 300, (len(got), len(want))
        for f in ("x", "y", "z", "curvature"):
            assert np.array_equal(got[f], want[f]), (lidar_type, f, int((got[f] != want[f]).sum()))
        # begin / end go through a `float` temporary in the reference (lidar_processing.cc:30-34,59-63); g++ -O3 keeps it in
        # double for the Ouster handler (-O0..-O2 round it like the oracle does), so the time stamps are compared to float
        # precision only.  They stamp the first frame; no point arithmetic depends on them.
        assert abs(gb - wb) <= 1e-7 * max(1.0, abs(wb)) and abs(ge - we) <= 1e-7 * max(1.0, abs(we)), (gb, wb, ge, we)


# ----------------------------------------------------------------------------- around the path: the TUM trajectory writer
def test_tum_writer_matches_the_reference(tmp_path):
    """TrajectorySaver::write (trajectory_saver.hpp:43-50: `t tx ty tz qx qy qz qw`, fixed, 9 decimals, quaternion from the
    rotation matrix) vs legkilo_amd.tum.write_tum - the same text, line for line, over all four branches of the
    matrix-to-quaternion conversion."""
    from legkilo_amd import tum

    rng = np.random.default_rng(21)
    rots = [np.eye(3)]
    for ax in range(3):                       # rotations by ~pi about each axis: trace < 0, each diagonal branch
        v = np.zeros(3)
        v[ax] = 3.1
        rots.append(ob.exp_log(v + rng.normal(0, 0.02, 3))[0])
    rots += [ob.exp_log(rng.normal(0, 1.5, 3))[0] for _ in range(40)]
    stamps = 1.7e9 + np.cumsum(rng.uniform(0.05, 0.15, len(rots)))
    poss = rng.normal(0, 50, (len(rots), 3))
    want = ob.ref_write_tum(stamps, np.array(rots), poss)
    tum.write_tum(tmp_path / "a.tum", stamps, rots, poss)
    got = open(tmp_path / "a.tum").read()
    assert got.count("\n") == len(rots)
    assert got == want
