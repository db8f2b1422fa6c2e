"""-m gpu parity tests: the HIP path (through the C-ABI) against the CPU oracle on identical seeded
inputs.  Integer / decision outputs (valid masks, tree shape, counters) must match exactly; fp64
quantities to the stated tolerances (different summation order only).  Target of the north star:
trajectory ATE delta < 1 mm; asserted here at 1e-6 m or tighter.
"""
import os

import numpy as np
import pytest

import scenes
from legkilo_amd import abi, config, synth

pytestmark = pytest.mark.gpu

CAPS = dict(max_roots=1 << 16, max_nodes=1 << 17, max_point_blocks=1 << 16, max_scan_points=1 << 17)


def rand_spd(rng, scale=1e-4):
    A = rng.normal(size=(30, 30))
    return scale * (A @ A.T / 30 + 0.1 * np.eye(30))


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300))


@pytest.fixture(scope="module")
def scene():
    return scenes.Scene(**CAPS)


@pytest.fixture()
def pair(scene, oracle_lib, hip_lib):
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    yield o, g
    g.close()
    o.close()


def both(pair, fn):
    return fn(pair[0]), fn(pair[1])


# ----------------------------------------------------------------------------- ESKF class surface
def test_eskf_predict_and_fx(pair, scene):
    o, g = pair
    rng = np.random.default_rng(1)
    x0 = synth.initial_state(scene.traj, 3.3, scene.P)
    x0[15:21] = rng.normal(0, 0.01, 6)
    P0 = rand_spd(rng)
    for obj in pair:
        obj.set_state(x0, P0)
        obj.init_process_cov_q()
    assert np.array_equal(o.get_Q(), g.get_Q())
    for dt in (0.002, 0.0371):
        assert rel_err(g.get_fx(dt), o.get_fx(dt)) < 1e-14
        assert np.allclose(g.get_function_f(dt), o.get_function_f(dt), rtol=1e-14, atol=1e-16)
    for dt, ps, pc in ((0.002, 0, 1), (0.004, 1, 0), (0.01, 1, 1)):
        for obj in pair:
            obj.predict(dt, ps, pc)
        xo, Po = o.get_state()
        xg, Pg = g.get_state()
        assert np.allclose(xg, xo, rtol=1e-13, atol=1e-14), np.abs(xg - xo).max()
        assert rel_err(Pg, Po) < 1e-12


def test_device_exp_on_both_sides_of_the_thresholds(scene, oracle_lib, hip_lib):
    """math_utils.hpp:19-32 (Exp(vec&&), identity below 1e-7: the (0,0) block of Fx, eskf.cc:74) and :54-68 (Exp(v1,v2,v3),
    identity below 1e-5: the rotation part of x (+) dx, eskf.cc:19) ON THE DEVICE (expv_1e7 / exp3_1e5, lk_device.h), against the
    reference's own two functions (oracle/_ref) - or the oracle's pinned restatement where _ref is not built."""
    which = "ref" if oracle_lib.ref_lib() is not None else "oracle"
    g = hip_lib.LegKiloHip(scene.cfg())
    rng = np.random.default_rng(5)
    R0 = oracle_lib.exp_log(np.array([0.3, -0.5, 0.8]))[0]
    dirs = [np.array([1.0, 0, 0]), np.array([0.6, -0.8, 0.0]), rng.normal(size=3)]
    mags = [0.0, 5e-8, 9.9e-8, 1.01e-7, 3e-6, 9.9e-6, 1.01e-5, 2e-5, 1e-3, 0.3, 3.0]
    for d in dirs:
        d = d / np.linalg.norm(d)
        for m in mags:
            v = m * d
            x = np.zeros(36)
            x[:9] = R0.reshape(9)
            x[9:12] = [1.0, 2.0, 3.0]
            x[27:30] = v                      # imu_w; dt = 1 -> f[0:3] = v, Fx(0:3,0:3) = Exp(-v)
            g.set_state(x, 1e-4 * np.eye(30))
            Fx = g.get_fx(1.0)
            want_fx = oracle_lib.exp_log(-v, which)[1]
            assert np.allclose(Fx[:3, :3], want_fx, rtol=0, atol=4e-16), (v, np.abs(Fx[:3, :3] - want_fx).max())
            if np.linalg.norm(-v) <= 1e-7:
                assert np.array_equal(Fx[:3, :3], np.eye(3)), v
            elif m >= 1.01e-7:
                assert not np.array_equal(Fx[:3, :3], np.eye(3)), v
            g.predict(1.0, True, False)
            xn, _ = g.get_state()
            want_R = R0 @ oracle_lib.exp_log(v, which)[0]
            assert np.allclose(xn[:9].reshape(3, 3), want_R, rtol=0, atol=1e-15), (v, np.abs(xn[:9].reshape(3, 3) - want_R).max())
            if np.linalg.norm(v) <= 1e-5:
                assert np.array_equal(xn[:9], x[:9]), v       # below the threshold Exp is exactly I: the rotation keeps its bits
            elif m >= 1.01e-5:
                assert not np.array_equal(xn[:9], x[:9]), v
    g.close()


@pytest.mark.parametrize("N", [1, 2, 7, 300, 5000])
def test_update_by_points(pair, scene, N):
    o, g = pair
    rng = np.random.default_rng(10 + N)
    x0 = synth.initial_state(scene.traj, 1.0, scene.P)
    P0 = rand_spd(rng)
    h6 = rng.normal(size=(N, 6))
    h6[:, 3:] /= np.linalg.norm(h6[:, 3:], axis=1, keepdims=True)
    z = rng.normal(0, 0.02, N)
    R = rng.uniform(0.001, 0.01, N)
    for obj in pair:
        obj.set_state(x0, P0)
        obj.update_by_points(h6, z, R)
    xo, Po = o.get_state()
    xg, Pg = g.get_state()
    assert np.allclose(xg, xo, rtol=1e-9, atol=1e-11), np.abs(xg - xo).max()
    assert rel_err(Pg, Po) < 1e-8


def test_update_by_imu_and_kin(pair, scene):
    o, g = pair
    rng = np.random.default_rng(5)
    x0 = synth.initial_state(scene.traj, 1.0, scene.P)
    P0 = rand_spd(rng)
    z6 = rng.normal(0, 0.1, 6)
    R6 = np.array([0.1, 0.1, 1.0, 0.01, 0.01, 0.01])
    M = 15
    ki_h = np.zeros((M, 30))
    ki_h[:6, 9:15] = np.eye(6)
    ki_h[:6, 18:24] = np.eye(6)
    ki_h[6:, :] = rng.normal(size=(M - 6, 30)) * (rng.random((M - 6, 30)) < 0.3)
    ki_z = rng.normal(0, 0.1, M)
    ki_R = rng.uniform(0.01, 0.2, M)
    for obj in pair:
        obj.set_state(x0, P0)
        obj.update_by_imu(z6, R6)
    xo, Po = o.get_state()
    xg, Pg = g.get_state()
    assert np.allclose(xg, xo, rtol=1e-10, atol=1e-12) and rel_err(Pg, Po) < 1e-10
    for obj in pair:
        obj.set_state(x0, P0)
        obj.update_by_kin_imu(ki_h, ki_z, ki_R)
    xo, Po = o.get_state()
    xg, Pg = g.get_state()
    assert np.allclose(xg, xo, rtol=1e-9, atol=1e-11), np.abs(xg - xo).max()
    assert rel_err(Pg, Po) < 1e-9


# ----------------------------------------------------------------------------- voxel map
def test_map_build_parity(pair, scene):
    o, g = pair
    t0 = 1.0
    for obj in pair:
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, dense=60000)  # dense: exercises cut_octo_tree and >50-pt leaves
    stats = scenes.compare_maps(o.map_export(), g.map_export())
    assert stats["roots"] > 1000 and stats["planes"] > 500 and stats["nodes"] > stats["roots"]


def test_map_build_clutter(pair):
    """BuildVoxelMap on planes + volumetric clutter: deep cut_octo_tree recursion, frozen >50-pt leaves,
    max-layer non-plane leaves (kept / count-only), un-initialised children."""
    o, g = pair
    rng = np.random.default_rng(11)
    n = 30000
    pw = np.concatenate([
        np.c_[rng.uniform(-3, 3, n // 3), rng.uniform(-3, 3, n // 3), rng.normal(0, 0.01, n // 3)],
        np.c_[rng.uniform(-3, 3, n // 3), rng.normal(1.26, 0.01, n // 3), rng.uniform(0, 3, n // 3)],
        rng.uniform(-1.5, 1.5, (n - 2 * (n // 3), 3)) + [-5, 5, 1.5],
        scenes.corner_clutter(rng),
    ]).astype(np.float32)
    rng.shuffle(pw)
    x0 = np.zeros(36)
    x0[[0, 4, 8]] = 1.0
    E = np.array(config.LEG_FUSION["extrinsic_R"], float).reshape(3, 3)
    T = np.array(config.LEG_FUSION["extrinsic_T"], float)
    body = ((pw.astype(np.float64) - T) @ E).astype(np.float32)  # so that world = E body + T at the identity pose
    for obj in pair:
        obj.set_state(x0, 1e-6 * np.eye(30))
        obj.map_build(pw, body)
    stats = scenes.compare_maps(o.map_export(), g.map_export())
    b = abi.parse_blob(g.map_export())
    layers = np.bincount(b["nodes"]["layer"], minlength=3)
    assert layers[1] > 50 and layers[2] > 50, layers
    assert int(((b["nodes"]["state"] & abi.LK_NODE_PTS_DROPPED) > 0).sum()) >= 0
    # then keep inserting into the same map through UpdateVoxelMap
    extra = np.concatenate([rng.uniform(-1.5, 1.5, (3000, 3)) + [-5, 5, 1.5], scenes.corner_clutter(rng, 40, 30)])
    A = rng.normal(size=(len(extra), 3, 3)) * 0.01
    var = (A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)).reshape(-1, 9)
    for obj in pair:
        obj.map_update(extra, var)
    scenes.compare_maps(o.map_export(), g.map_export())


def test_map_import_export_roundtrip(pair, scene):
    o, g = pair
    t0 = 1.0
    x0 = scenes.init_filter(o, scene, t0)
    scenes.first_frame(o, scene, t0, x0)
    scenes.replay_vlp(o, scene, t0, 5)
    blob = o.map_export()
    g.map_import(blob)
    scenes.compare_maps(blob, g.map_export(), rtol=0.0)
    nr, nn, nb = g.map_stats()
    assert nr == o.map_stats()


def test_map_update_surface(pair, scene):
    """VoxelMapManager::UpdateVoxelMap on caller-supplied pointWithVar, incl. ordering inside a voxel."""
    o, g = pair
    rng = np.random.default_rng(3)
    n = 4000
    # points on three planes + clutter, many per voxel so that init / refit / freeze / cut all trigger
    pw = np.concatenate([
        np.c_[rng.uniform(0, 4, n // 4), rng.uniform(0, 4, n // 4), rng.normal(0, 0.01, n // 4)],
        np.c_[rng.uniform(0, 4, n // 4), rng.normal(2.0, 0.01, n // 4), rng.uniform(0, 2, n // 4)],
        np.c_[rng.normal(-1.0, 0.01, n // 4), rng.uniform(-3, 1, n // 4), rng.uniform(0, 2, n // 4)],
        rng.uniform(-2, 2, (n // 4, 3)) * [0.5, 0.5, 0.5] + [-6, 6, 1],
        scenes.corner_clutter(rng, 50, 40),
    ])
    rng.shuffle(pw)
    A = rng.normal(size=(len(pw), 3, 3)) * 0.01
    var = A @ A.transpose(0, 2, 1) + 1e-5 * np.eye(3)
    for i in range(0, len(pw), 1000):
        for obj in pair:
            obj.map_update(pw[i:i + 1000], var[i:i + 1000].reshape(-1, 9))
        scenes.compare_maps(o.map_export(), g.map_export())


# ----------------------------------------------------------------------------- residuals (config 2, reduced)
def mature_oracle_map(o, scene, t0, n_scans=10):
    x0 = scenes.init_filter(o, scene, t0)
    scenes.first_frame(o, scene, t0, x0)
    scenes.replay_vlp(o, scene, t0, n_scans)
    return o.map_export()


def test_residuals_on_imported_map(pair, scene):
    o, g = pair
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    g.map_import(blob)
    xs, Ps = o.get_state()
    g.set_state(xs, Ps)
    ts = t0 + 1.0
    pts = synth.dense_scan(scene.world, scenes.Frozen(scene.traj, ts), ts, scene.P, n=20000, n_buckets=1)
    xb = scenes.xyz_of(pts)
    ho, zo, Ro, vo = o.residuals(xb)
    hg, zg, Rg, vg = g.residuals(xb)
    assert vo.sum() > 2000, vo.sum()
    mism = int((vo != vg).sum())
    assert mism == 0, f"valid mask differs at {mism} of {len(vo)} points"
    scenes.rows_close(hg, zg, Rg, ho, zo, Ro, vo)


def test_build_single_residual_per_point(pair, scene, oracle_lib):
    """lk_match_points == VoxelMapManager::build_single_residual (voxel_map.cc:363-427) of the oracle, point by point, on the same map:
    caller-held world points with caller-held covariances, on their home voxel and two neighbouring keys (the searches of KILO.cc:149-185),
    keys without a root voxel, covariances small (most candidates fail the 3-sigma gate) and large (several planes of a cut voxel pass and
    the probability decides)."""
    o, g = pair
    t0 = 1.0
    mature_oracle_map(o, scene, t0)
    rng = np.random.default_rng(41)
    clutter = scenes.corner_clutter(rng, n_cells=40, per_cell=70)   # cuts voxels down to layer 2: candidates below the root, several per voxel
    cvar = np.tile((np.eye(3) * 4e-4).reshape(1, 9), (len(clutter), 1))
    o.map_update(clutter, cvar)
    blob = o.map_export()
    g.map_import(blob)
    ts = t0 + 1.0
    pts = synth.dense_scan(scene.world, scenes.Frozen(scene.traj, ts), ts, scene.P, n=3000, n_buckets=1)
    xs, _ = o.get_state()
    Rw, tw = np.asarray(xs[:9]).reshape(3, 3), np.asarray(xs[9:12])
    pw = (scenes.xyz_of(pts).astype(np.float64) + np.asarray(scene.P["extrinsic_T"])) @ Rw.T + tw + rng.normal(0, 0.03, (len(pts), 3))
    pw = np.concatenate([pw, clutter[:1200] + rng.normal(0, 0.01, (1200, 3)), rng.uniform(-30, 30, (300, 3))])
    vs = float(scene.P["voxel_size"])
    keys, P, V = [], [], []
    for i, p in enumerate(pw):
        k0 = oracle_lib.key_floor(p, vs)
        A = rng.normal(size=(3, 3))
        var = (A @ A.T) * (1e-5 if i % 3 else 4e-3) + np.eye(3) * 1e-6
        for dk in ((0, 0, 0), (1, 0, 0), (0, -1, 0), (0, 0, 1)):
            keys.append([a + b for a, b in zip(k0, dk)]), P.append(p), V.append(var)
    keys, P, V = np.array(keys, dtype=np.int32), np.array(P), np.array(V)
    mg = g.match_points(keys, P, V)
    n_found = n_ok = n_deep = 0
    for i in range(len(keys)):
        mo = o.match_voxel(keys[i], P[i], V[i].reshape(9))
        assert (mo["found"], mo["success"]) == (bool(mg["found"][i]), bool(mg["success"][i])), (i, keys[i], mo, {k: v[i] for k, v in mg.items()})
        n_found += mo["found"]
        if mo["success"]:
            n_ok += 1
            n_deep += mo["layer"] > 0
            assert mo["layer"] == mg["layer"][i]
            assert np.array_equal(mo["normal"], mg["normal"][i]) and np.array_equal(mo["center"], mg["center"][i])   # the same plane of the same map
            assert mo["d"] == mg["d"][i]
            assert np.isclose(mo["dis_to_plane"], mg["dis_to_plane"][i], rtol=1e-6, atol=1e-9)
            assert np.isclose(mo["prob"], mg["prob"][i], rtol=1e-9), (mo["prob"], mg["prob"][i])
        else:
            assert mg["layer"][i] == -1 and mg["prob"][i] == 0.0
    assert n_found > 3000 and n_ok > 1500 and n_deep > 20, (n_found, n_ok, n_deep)
    assert n_found < len(keys)
    scenes.compare_maps(blob, g.map_export())   # the query left the map alone


def test_update_points_bucket_and_insert(pair, scene):
    o, g = pair
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    g.map_import(blob)
    xs, Ps = o.get_state()
    g.set_state(xs, Ps)
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    tp, tu = o.get_times()
    g.set_times(tp, tu)
    ts = tp + 0.01
    ds = scenes.vlp_scan_input(scene, ts, 77)
    xb = scenes.xyz_of(ds)[:1500]
    wo, io_, neo = o.update_points(ts, xb)
    wg, ig, neg = g.update_points(ts, xb)
    assert neo == neg and neo > 100, (neo, neg)
    assert np.array_equal(io_, ig)
    assert np.abs(wo - wg).max() < 1e-5
    xo, Po = o.get_state()
    xg, Pg = g.get_state()
    assert np.allclose(xg, xo, rtol=1e-9, atol=1e-10), np.abs(xg - xo).max()
    assert rel_err(Pg, Po) < 1e-7
    assert o.get_times() == g.get_times()
    scenes.compare_maps(o.map_export(), g.map_export())


def test_update_points_edge_cases(pair, scene):
    """Buckets the reference's loop can meet: nothing matches (no update; the covariance is then propagated again from
    the LAST update time on the next bucket - KILO.cc:110-111 - and every point still goes into the map), exactly one
    match (the +1e-4 branch of eskf.cc:98-104), sizes around the wave width, and more than 64 points piled into one
    voxel (the long-list replay of the insert).  Same state, covariance, times, counts and map as the oracle each time;
    empty and oversized inputs are loud errors."""
    o, g = pair
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    g.map_import(blob)
    xs, Ps = o.get_state()
    g.set_state(xs, Ps)
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    tp, tu = o.get_times()
    g.set_times(tp, tu)
    rng = np.random.default_rng(4242)
    ds = scenes.xyz_of(scenes.vlp_scan_input(scene, tp + 0.01, 55))
    far = (rng.uniform(-1, 1, (40, 3)) + np.array([300.0, -200.0, 50.0])).astype(np.float32)      # no voxel anywhere near
    pile = (np.array([310.0, -210.0, 40.0]) + rng.uniform(0.02, 0.23, (90, 3))).astype(np.float32)  # 90 points, one new voxel
    cases = [("no match", far), ("after no match", ds[:700]), ("pile > 64 into one voxel", pile), ("one point", ds[700:701]),
             ("63", ds[701:764]), ("64", ds[764:828]), ("65", ds[828:893]), ("257", ds[893:1150])]
    cases.insert(2, ("single match", None))   # built below from the state at that moment
    t = tp
    seen_zero = seen_one = False
    for name, xb in cases:
        t += 0.004
        if xb is None:  # exactly ONE match: a point the oracle's matcher accepts right now, plus points that hit nothing
            valid = o.residuals(ds[1150:])[3]
            k = 1150 + int(np.flatnonzero(valid)[0])
            xb = np.concatenate([ds[k:k + 1], far[:5] + 7.0])
        wo, io_, neo = o.update_points(t, xb)
        wg, ig, neg = g.update_points(t, xb)
        assert neo == neg, (name, neo, neg)
        seen_zero |= neo == 0
        seen_one |= neo == 1
        assert np.array_equal(io_, ig), name
        assert np.abs(wo - wg).max() < 2e-4, (name, np.abs(wo - wg).max())   # f32 world points at |x| ~ 300 m
        (xo, Po), (xg, Pg) = o.get_state(), g.get_state()
        assert np.allclose(xg, xo, rtol=1e-9, atol=1e-9), (name, np.abs(xg - xo).max())
        assert rel_err(Pg, Po) < 1e-7, name
        assert o.get_times() == g.get_times(), name
    assert seen_zero and seen_one, "the no-match and the single-match branch must both have been exercised"
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-7)
    cm = scenes.canon_map(g.map_export())
    assert any(k[0] > 500 for k in cm), "the far points must have created voxels"
    from legkilo_amd import binding

    dss = scenes.vlp_scan_input(scene, tp + 0.5, 56)
    with pytest.raises(binding.LegKiloError):
        g.process_scan(dss[:0], t + 0.1)                                   # empty scan
    with pytest.raises(binding.LegKiloError):
        g.update_points(t + 0.1, np.zeros((g.cfg.max_scan_points + 1, 3), dtype=np.float32))   # bucket larger than max_scan_points


# ----------------------------------------------------------------------------- full sequences
@pytest.mark.parametrize("literal", [True, False])
def test_sequence_imu_mode(pair, scene, literal, tmp_path):
    """Config 1 shape: first-frame build + 12 real-like scans (hundreds of small buckets each), IMU-only mode.
    literal=True : the oracle runs the reference's literal N x N inverse (eskf.cc:105-112) for every bucket;
    literal=False: the oracle uses the algebraically identical 6 x 6 form -> only summation order differs,
                   so the tolerance is two orders tighter."""
    o, g = pair
    o.set_literal_max_n(512 if literal else 0)
    tol = 1e-6 if literal else 1e-7
    t0 = 1.0
    for obj in pair:
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    ro = scenes.replay_vlp(o, scene, t0, 12)
    rg = scenes.replay_vlp(g, scene, t0, 12)
    worst = 0.0
    for k, ((po, xo), (pg, xg)) in enumerate(zip(ro, rg)):
        assert (po.n_buckets, po.n_updates, po.n_effect) == (pg.n_buckets, pg.n_updates, pg.n_effect), k
        worst = max(worst, np.abs(xo[9:12] - xg[9:12]).max())
    assert worst < tol, worst
    ate = scenes.ate([x[9:12] for _, x in ro], [x[9:12] for _, x in rg])
    assert ate < tol, ate  # north star: ATE delta < 1 mm
    print(f"literal={literal}: worst position delta {worst:.3e} m, ATE delta {ate:.3e} m")
    # the same statement through the on-disk format the reference writes (trajectory_saver.hpp:43-50)
    from legkilo_amd import tum
    stamps = [t0 + 0.1 * (k + 1) for k in range(len(ro))]
    tum.write_tum(tmp_path / "cpu.txt", stamps, [x[:9] for _, x in ro], [x[9:12] for _, x in ro])
    tum.write_tum(tmp_path / "gpu.txt", stamps, [x[:9] for _, x in rg], [x[9:12] for _, x in rg])
    e, n = tum.ate_files(tmp_path / "cpu.txt", tmp_path / "gpu.txt")
    assert n == len(ro) and e < 1e-3, e  # north star: trajectory ATE delta < 1 mm (files carry 9 decimals)
    # the stored points / plane centres carry the accumulated state delta
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=10 * tol)


def test_sequence_kin_mode(scene, oracle_lib, hip_lib):
    """Config 4 shape (reduced): diter.yaml parameters, 500 Hz kinematic+IMU observations, leg factors on."""
    sc = scenes.Scene(params=config.DITER, **CAPS)
    o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
    g = hip_lib.LegKiloHip(sc.cfg())
    t0 = 2.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, sc, t0)
        scenes.first_frame(obj, sc, t0, x0)
    ro = scenes.replay_vlp(o, sc, t0, 6, use_kin=True)
    rg = scenes.replay_vlp(g, sc, t0, 6, use_kin=True)
    for k, ((po, xo), (pg, xg)) in enumerate(zip(ro, rg)):
        assert (po.n_buckets, po.n_updates, po.n_effect) == (pg.n_buckets, pg.n_updates, pg.n_effect), k
        assert np.allclose(xo, xg, rtol=1e-7, atol=1e-8), (k, np.abs(xo - xg).max())
    g.close()
    o.close()


def ouster_message(sc, tb, k):
    """One OS1-64-like sensor message of the config-4 run: 64 x 1024 rays, Ouster PointCloud2 point layout, `t` in ns."""
    import preprocess_oracle as po

    pts, t_ns = synth.ouster_scan(sc.world, sc.traj, tb, sc.P, seed_noise=4000 + k)
    raw = np.zeros(len(pts), dtype=po.OUSTER_DTYPE)
    raw["x"], raw["y"], raw["z"], raw["t"] = pts["x"], pts["y"], pts["z"], t_ns
    raw["intensity"] = 10.0
    dt = po.OUSTER_DTYPE
    layout = dict(point_step=dt.itemsize, off_x=dt.fields["x"][1], off_y=dt.fields["y"][1], off_z=dt.fields["z"][1],
                  off_time=dt.fields["t"][1], lidar_type=2)
    return raw, layout


N_CONFIG4_SCANS = 100


def test_config4_ouster_leg_fusion_run(oracle_lib, hip_lib, tmp_path):
    """Config 4 at the shape SURVEY.md 8(d) states: diter.yaml, an Ouster-like 64 x 1024 scan every 0.1 s (`t` in ns, time_scale
    1e-9, filter_num 3, voxel grid 0.5 m), 500 Hz kinematic + IMU messages with a trot contact pattern, only_imu_use: false, a
    10 s segment (100 scans) of the 60 s figure-eight.  GPU chain = the product's own: lk_decode_scan -> lk_process_raw_scan
    (voxel grid + time sort + bucket loop with predictUpdateKinImu between the buckets + map insert); CPU chain = decode oracle
    -> preprocess oracle -> the oracle's KILO::process.  Every scan: decode bit-exact, identical bucket / update / match counts;
    trajectory: ATE delta through the TUM files < 1e-6 m (north star: < 1 mm)."""
    import preprocess_oracle as po
    from legkilo_amd import tum

    sc = scenes.Scene(params=config.DITER, **CAPS)
    P = sc.P
    assert P["time_scale"] == 1e-9 and P["lidar_type"] == 2 and not P["only_imu_use"] and P["voxel_grid_resolution"] == 0.5
    o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
    g = hip_lib.LegKiloHip(sc.cfg())
    t0 = 3.0
    raw0, layout = ouster_message(sc, t0, 999)
    first, _, _ = po.decode_vec(raw0, 2, P["time_scale"], P["filter_num"], P["blind"], header_stamp=t0)
    # first frame from a static sensor: both sides build the map from the same decoded cloud (BuildVoxelMap, KILO.cc:332-353)
    raw_static, _ = synth.ouster_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, P, seed_noise=3999)
    xb = scenes.xyz_of(raw_static[::3])
    for obj in (o, g):
        x0 = scenes.init_filter(obj, sc, t0)
        obj.map_build(scenes.world_of(x0, xb, P), xb)
    stamps, rows_o, rows_g = [], [], []
    n_msgs = 0
    worst = 0.0
    for k in range(N_CONFIG4_SCANS):
        tb = t0 + 0.1 * k
        raw, layout = ouster_message(sc, tb, k)
        kins = synth.kin_stream(sc.traj, tb, tb + 0.1, P, seed=5000 + k)
        n_msgs += len(kins)
        dec_o, b_o, e_o = po.decode_vec(raw, 2, P["time_scale"], P["filter_num"], P["blind"], header_stamp=tb)
        dec_g, b_g, e_g = g.decode_scan(raw.tobytes(), len(raw), layout, P["time_scale"], P["filter_num"], P["blind"], header_stamp=tb)
        assert (b_o, e_o) == (b_g, e_g) and len(dec_o) == len(dec_g)
        for f in dec_o.dtype.names:
            assert np.array_equal(dec_o[f], dec_g[f]), (k, f)
        ds = po.preprocess(dec_o, P["voxel_grid_resolution"])
        pose_o, _ = o.process_scan(ds, b_o, kins=kins)
        pose_g, nd = g.process_raw_scan(dec_g, P["voxel_grid_resolution"], b_g, kins=kins)
        assert nd == len(ds), (k, nd, len(ds))
        assert (pose_o.n_buckets, pose_o.n_updates, pose_o.n_effect) == (pose_g.n_buckets, pose_g.n_updates, pose_g.n_effect), \
            (k, pose_o.n_buckets, pose_o.n_updates, pose_o.n_effect, pose_g.n_buckets, pose_g.n_updates, pose_g.n_effect)
        assert pose_o.n_buckets > 300 and pose_o.n_effect > 500, (k, pose_o.n_buckets, pose_o.n_effect)
        stamps.append(e_o)
        rows_o.append((np.array(pose_o.rot), np.array(pose_o.pos)))
        rows_g.append((np.array(pose_g.rot), np.array(pose_g.pos)))
        worst = max(worst, float(np.abs(rows_o[-1][1] - rows_g[-1][1]).max()))
    assert n_msgs >= 49 * N_CONFIG4_SCANS
    tum.write_tum(tmp_path / "cpu.txt", stamps, [r for r, _ in rows_o], [p_ for _, p_ in rows_o])
    tum.write_tum(tmp_path / "gpu.txt", stamps, [r for r, _ in rows_g], [p_ for _, p_ in rows_g])
    e, n = tum.ate_files(tmp_path / "cpu.txt", tmp_path / "gpu.txt")
    assert n == N_CONFIG4_SCANS
    ate_state = scenes.ate([p_ for _, p_ in rows_o], [p_ for _, p_ in rows_g])
    truth = sc.traj.pos(np.array(stamps))
    ate_truth_o = scenes.ate([p_ for _, p_ in rows_o], truth)
    ate_truth_g = scenes.ate([p_ for _, p_ in rows_g], truth)
    print(f"config 4: {N_CONFIG4_SCANS} scans, worst position delta {worst:.3e} m, ATE delta {ate_state:.3e} m (TUM files: {e:.3e}), "
          f"ATE vs ground truth cpu {ate_truth_o:.4f} m / gpu {ate_truth_g:.4f} m")
    assert ate_state < 1e-6 and e < 1e-6 + 2e-9, (ate_state, e)     # the files carry 9 decimals
    assert abs(ate_truth_o - ate_truth_g) < 1e-6
    (xo, _), (xg, _) = o.get_state(), g.get_state()
    assert np.allclose(xo[:12], xg[:12], rtol=0, atol=1e-6), np.abs(xo[:12] - xg[:12]).max()
    so, sg = scenes.canon_map(o.map_export()), scenes.canon_map(g.map_export())
    assert set(so) == set(sg)
    g.close()
    o.close()


# ----------------------------------------------------------------------------- 100k-point configs
@pytest.fixture(scope="module")
def big(scene, oracle_lib, hip_lib):
    """Config 3 at full size: map from a dense first frame, then 100k-pt scans in 5 buckets of 20k."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    t0 = 5.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, dense=100000)
    yield scene, o, g, t0
    g.close()
    o.close()


def test_config3_full_size(big):
    scene, o, g, t0 = big
    for k in range(2):
        tb = t0 + 0.1 * k
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=2002 + k,
                               seed_noise=3003 + k)
        po, _ = o.process_scan(pts, tb)
        pg, wg = g.process_scan(pts, tb, want_world=True)
        assert (po.n_buckets, po.n_updates) == (pg.n_buckets, pg.n_updates) == (5, 5)
        # decision parity at full size: match counts are integer work - exact
        assert int(po.n_effect) == int(pg.n_effect), (po.n_effect, pg.n_effect)
        xo, Po = o.get_state()
        xg, Pg = g.get_state()
        assert np.abs(xo[9:12] - xg[9:12]).max() < 1e-6, np.abs(xo[9:12] - xg[9:12]).max()
        assert np.allclose(xo, xg, rtol=1e-6, atol=1e-7)
        # size-independent property: re-projected world points = R (E p + T) + pos of the bucket's posterior
        assert np.isfinite(wg).all()
    # the whole map at full size: tree shape, counters and state bits exactly, planes and stored points to tolerance - the insert's
    # apply / fallback / long-list paths all run at this size (voxel_map.cc:185-241)
    st = scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-7)
    assert st["roots"] > 10000, st


def test_config3_51_buckets_full_size(big):
    """SURVEY 8(d) config 3, the reference's own time quantisation: a 100 000-point scan in 51 two-ms bins (lidar_processing.cc:48)
    = 51 predict / residual / update / re-project / insert cycles of ~2 000 points each (the variant bench.py times as
    extra.stream51_*), against the oracle's bucket loop: identical bucket / update / match counts, all 36 state entries to 1e-6,
    the same set of voxels afterwards."""
    scene, o, g, t0 = big
    for k in range(2):
        tb = t0 + 0.3 + 0.1 * k
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=51, seed_scan=8208 + k, seed_noise=8308 + k)
        off, _ = synth.buckets_of(pts)
        assert len(off) - 1 == 51
        for obj in (o, g):
            obj.set_state(synth.initial_state(scene.traj, tb, scene.P), 1e-6 * np.eye(30))
            obj.set_times(tb, tb)
        po, _ = o.process_scan(pts, tb)
        pg, _ = g.process_scan(pts, tb)
        assert (po.n_buckets, po.n_updates) == (pg.n_buckets, pg.n_updates) == (51, 51), (po.n_buckets, po.n_updates, pg.n_buckets, pg.n_updates)
        assert int(po.n_effect) == int(pg.n_effect), (po.n_effect, pg.n_effect)
        assert po.n_effect > 30000
        xo, _ = o.get_state()
        xg, _ = g.get_state()
        # every block of the state - rotation, position, velocity, biases, gravity, IMU states, kinematic states - at 1e-6
        assert np.abs(xo - xg).max() < 1e-6, (np.abs(xo - xg).max(), int(np.abs(xo - xg).argmax()))
    st = scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-7)   # not only the key set: every voxel's tree, counters, state bits, planes, points
    assert st["roots"] > 10000, st


def test_config3_soak_full_size(scene, oracle_lib, hip_lib):
    """16 consecutive 100 k-point scans (5 buckets each) with map insert on a moving trajectory: the map keeps growing,
    leaves refit / freeze / get cut, point blocks are recycled, the generic insert fallback and the long-list replay get
    their share.  Identical match counts on every scan, positions to 1e-6 (measured: 1e-8), the same map at the end, no pool
    overflow."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    t0 = 8.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, dense=60000)
    worst = 0.0
    for k in range(16):
        tb = t0 + 0.1 * k
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=4100 + k, seed_noise=4200 + k)
        po, _ = o.process_scan(pts, tb)
        pg, _ = g.process_scan(pts, tb)
        assert (po.n_buckets, po.n_updates) == (pg.n_buckets, pg.n_updates) == (5, 5), k
        assert int(po.n_effect) == int(pg.n_effect), (k, po.n_effect, pg.n_effect)
        xo, _ = o.get_state()
        xg, _ = g.get_state()
        worst = max(worst, float(np.abs(xo[9:12] - xg[9:12]).max()))
        assert np.allclose(xo[:12], xg[:12], rtol=1e-6, atol=1e-6), (k, np.abs(xo[:12] - xg[:12]).max())   # rotation, position
        # velocity / bias / IMU states are driven by large process noise and amplify 1e-9 differences over the 16-scan closed
        # loop (DESIGN "Closed-loop sensitivity"): measured <= 1.5e-6, asserted at 1e-5
        assert np.abs(xo - xg).max() < 1e-5, (k, np.abs(xo - xg).max(), int(np.abs(xo - xg).argmax()))
    assert worst < 1e-6, worst
    # after 16 scans x 100 000 points: the maps voxel by voxel (tree shape, counters, state bits exact; planes 1e-5; stored points to the
    # closed loop's own position tolerance, 1e-6 - measured 1e-8)
    st = scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-6)
    assert st["roots"] > 20000, st
    roots, nodes, blocks = g.map_stats()
    assert nodes >= roots > 20000 and blocks > 1000
    g.close()
    o.close()


@pytest.mark.parametrize("use_kin", [False, True])
def test_scan_resident_kernel_equals_per_bucket_launches(scene, oracle_lib, hip_lib, use_kin):
    """The scan-resident stream kernel (one launch per scan: the whole bucket loop with its messages and the map insert in one
    resident workgroup, lk_scan_stream_kernel) against the per-bucket launches of the same library (lk_stream_resident(0)) on
    config-1 scans with IMU (only_imu_use) or kinematic + IMU (leg fusion) messages, over a young map that is still growing (roots
    created, planes initialised / refitted / cut while the scans run): state, covariance, re-projected cloud and map bit for bit on
    every scan, and both equal to the oracle (counts exact, state 1e-7)."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=not use_kin)
    g = hip_lib.LegKiloHip(scene.cfg())
    g_pb = hip_lib.LegKiloHip(scene.cfg())
    g_pb.stream_resident(False)
    t0 = 21.0
    for obj in (o, g, g_pb):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    for k in range(4):
        tb = t0 + 0.1 * k
        ds = scenes.vlp_scan_input(scene, tb, k)
        kw = dict(kins=synth.kin_stream(scene.traj, tb, tb + 0.1, scene.P, seed=3003 + k)) if use_kin else \
            dict(imus=synth.imu_stream(scene.traj, tb, tb + 0.1, seed=3003 + k))
        po, _ = o.process_scan(ds, tb, **kw)
        pg, wg = g.process_scan(ds, tb, want_world=True, **kw)
        pp, wp = g_pb.process_scan(ds, tb, want_world=True, **kw)
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (pg.n_buckets, pg.n_updates, int(pg.n_effect)) == \
            (pp.n_buckets, pp.n_updates, int(pp.n_effect)), (k, po.n_effect, pg.n_effect, pp.n_effect)
        assert po.n_buckets > 100
        (xo, _), (xg, Pg), (xp, Pp) = o.get_state(), g.get_state(), g_pb.get_state()
        assert np.abs(xo - xg).max() < 1e-7, (k, np.abs(xo - xg).max())
        assert np.array_equal(xg, xp) and np.array_equal(Pg, Pp), (k, np.abs(xg - xp).max())
        assert np.array_equal(wg, wp)
    scenes.maps_identical(g.map_export(), g_pb.map_export())
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-7)
    print("scan-resident kernel: scans, launches beyond one per scan:", g.stream_resident_stats())
    for obj in (g, g_pb, o):
        obj.close()


def test_scan_resident_kernel_edge_buckets(scene, oracle_lib, hip_lib):
    """The scan-resident kernel on bucket shapes a real scan rarely has: 1, 2, 63, 64, 65, 257, 300 and 512 points (one to eight
    tiles evaluated one after the other by the filter wave, re-evaluated together on a conflict), IMU messages before the first
    bucket, between buckets and after the last one, on a map young enough that most buckets insert: bit-identical to the per-bucket
    launches, counts identical to the oracle, state 1e-7."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    g_pb = hip_lib.LegKiloHip(scene.cfg())
    g_pb.stream_resident(False)
    t0 = 31.0
    for obj in (o, g, g_pb):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    sizes = [1, 63, 2, 64, 65, 257, 1, 300, 512, 7]
    for k in range(3):
        tb = t0 + 0.1 * k
        src = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=sum(sizes), n_buckets=1, seed_scan=7300 + k, seed_noise=7400 + k)
        pts = src.copy()
        curv = np.concatenate([np.full(n_, np.float32(0.002 * (i + 1))) for i, n_ in enumerate(sizes)])
        pts["curvature"] = curv
        # generic fallback items inside the scan: 90 points piled into ONE new voxel (a root with more than 64 queued points) in the 257-point
        # bucket, volumetric clutter (voxels that are cut, init_octo_tree / cut_octo_tree) in the 300-point one
        rngk = np.random.default_rng(8100 + k)
        o257, o300 = sum(sizes[:5]), sum(sizes[:7])
        pile = np.array([60.0 + 3.0 * k, -45.0, 12.0]) + rngk.uniform(0.02, 0.23, (90, 3))
        clutter = np.array([14.0, 9.0 + 2.0 * k, 2.0]) + rngk.uniform(-0.7, 0.7, (300, 3))
        for c, ax in enumerate("xyz"):
            pts[ax][o257:o257 + 90] = pile[:, c].astype(np.float32)
            pts[ax][o300:o300 + 300] = clutter[:, c].astype(np.float32)
        imus = synth.imu_stream(scene.traj, tb - 0.004, tb + 0.1, seed=5003 + k)   # some stamped before the first bucket, some after the last
        po, _ = o.process_scan(pts, tb, imus=imus)
        pg, wg = g.process_scan(pts, tb, imus=imus, want_world=True)
        pp, wp = g_pb.process_scan(pts, tb, imus=imus, want_world=True)
        assert po.n_buckets == len(sizes)
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (pg.n_buckets, pg.n_updates, int(pg.n_effect)) == \
            (pp.n_buckets, pp.n_updates, int(pp.n_effect)), (k, po.n_effect, pg.n_effect, pp.n_effect)
        (xo, _), (xg, Pg), (xp, Pp) = o.get_state(), g.get_state(), g_pb.get_state()
        assert np.abs(xo - xg).max() < 1e-7, (k, np.abs(xo - xg).max())
        assert np.array_equal(xg, xp) and np.array_equal(Pg, Pp) and np.array_equal(wg, wp), k
    scenes.maps_identical(g.map_export(), g_pb.map_export())
    # buckets of hundreds of points on a young map queue more than 64 points on a root and cut voxels: generic fallback items, which end a
    # launch of the resident kernel, run as a launch of their own and have the kernel launched again from where it stopped (LkResume)
    n_scans, n_relaunch = g.stream_resident_stats()
    print(f"scan-resident kernel: {n_scans} scans, {n_relaunch} launches beyond one per scan (fallback rounds)")
    assert n_scans == 3 and n_relaunch >= 1, (n_scans, n_relaunch)
    # a one-bucket, one-point scan
    one = src[:1].copy()
    one["curvature"] = 0.0
    for obj in (g, g_pb):
        obj.process_scan(one, t0 + 0.5)
    assert np.array_equal(g.get_state()[0], g_pb.get_state()[0])
    for obj in (g, g_pb, o):
        obj.close()


@pytest.mark.parametrize("kind", ["scan-resident", "grid-resident"])
def test_resident_kernel_timeout_restores_the_filter(scene, hip_lib, kind):
    """The error contract of the resident stream kernels (include/legkilo_hip.h, LK_ERR_TIMEOUT): a bounded device-side wait that is given up
    fails the call, the FILTER keeps its pre-scan state (the scan started with a copy that is put back), the status is not sticky, and the handle
    goes on once its map has been restored.  lk_test_stall injects the fault: one role stops answering at the scan's fourth bucket, every wait is
    bounded by 30 ms.  The same scan through an untouched second handle is the reference for "goes on"."""
    from legkilo_amd import binding

    g = hip_lib.LegKiloHip(scene.cfg())
    g_ref = hip_lib.LegKiloHip(scene.cfg())
    t0 = 51.0
    for obj in (g, g_ref):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, **({"dense": 20000} if kind == "grid-resident" else {}))
        if kind == "grid-resident":
            obj.stream_grid(2)

    def scan_of(k):
        tb = t0 + 0.1 * k
        if kind == "scan-resident":
            return scenes.vlp_scan_input(scene, tb, k), tb
        return synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=51, seed_scan=9700 + k, seed_noise=9800 + k), tb

    pts, tb = scan_of(0)
    g.process_scan(pts, tb)
    g_ref.process_scan(pts, tb)
    blob = g.map_export()
    x_pre, P_pre = g.get_state()
    t_pre = g.get_times()
    pts, tb = scan_of(1)
    g.test_stall(30)
    with pytest.raises(binding.LegKiloError, match="error -6"):   # LK_ERR_TIMEOUT
        g.process_scan(pts, tb)
    g.test_stall(0)
    x_after, P_after = g.get_state()
    assert np.array_equal(x_pre, x_after) and np.array_equal(P_pre, P_after), "the filter must keep its pre-scan state"
    g.map_import(blob)       # the map may hold a partial insert: restore it ...
    g.set_times(*t_pre)
    pg, _ = g.process_scan(pts, tb)   # ... and replay the scan
    pr, _ = g_ref.process_scan(pts, tb)
    assert (pg.n_buckets, pg.n_updates, int(pg.n_effect)) == (pr.n_buckets, pr.n_updates, int(pr.n_effect))
    (xg, Pg), (xr, Pr) = g.get_state(), g_ref.get_state()
    assert np.array_equal(xg, xr) and np.array_equal(Pg, Pr)
    scenes.maps_identical(g.map_export(), g_ref.map_export())
    for obj in (g, g_ref):
        obj.close()


@pytest.mark.parametrize("nb", [5, 51, -51])
def test_scan_grid_kernel_equals_per_bucket_launches(scene, oracle_lib, hip_lib, nb):
    """The grid-resident stream kernel (lk_scan_grid_kernel: the whole bucket loop of a scan of LARGE buckets as one launch of
    co-resident workgroups, grid barriers instead of kernel boundaries, lk_stream_grid) against the per-bucket launches on a second
    handle: state, covariance, re-projected cloud and map bit for bit over three consecutive 100 000-point scans (5 buckets of
    20 000 / 51 two-ms bins of ~1 960) on a young map - inits, refits, cuts, emitted leaf groups and fallback items all occur - and
    both equal to the oracle (counts exact, state 1e-6).  nb = -51: the 51 buckets are a RANDOM partition of the scan instead of azimuth
    sectors, so every bucket's insert changes planes the very next bucket matches all over the scene - whatever a barrier of the
    resident kernel failed to hand over would show up in the next bucket's counts."""
    scattered = nb < 0
    nb = abs(nb)
    rng = np.random.default_rng(424242)
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    g_seq = hip_lib.LegKiloHip(scene.cfg())
    g.stream_grid(2)       # any bucket size (the default takes buckets up to 4 096 points)
    g_seq.stream_grid(0)
    t0 = 13.0
    for obj in (o, g, g_seq):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, dense=20000)
    for k in range(3):
        tb = t0 + 0.1 * k
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=nb, seed_scan=9300 + k, seed_noise=9400 + k)
        if scattered:
            curv = pts["curvature"].copy()
            pts = pts[rng.permutation(len(pts))]
            pts["curvature"] = curv
        po, _ = o.process_scan(pts, tb)
        pg, wg = g.process_scan(pts, tb, want_world=True)
        ps, ws = g_seq.process_scan(pts, tb, want_world=True)
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (pg.n_buckets, pg.n_updates, int(pg.n_effect)) == (ps.n_buckets, ps.n_updates, int(ps.n_effect)), \
            (k, po.n_effect, pg.n_effect, ps.n_effect)
        xo, _ = o.get_state()
        xg, Pg = g.get_state()
        xs, Ps = g_seq.get_state()
        assert np.abs(xo - xg).max() < 1e-6, (k, np.abs(xo - xg).max())
        assert np.array_equal(xg, xs) and np.array_equal(Pg, Ps), (k, np.abs(xg - xs).max())
        assert np.array_equal(wg, ws), k
    scenes.maps_identical(g.map_export(), g_seq.map_export())
    assert set(scenes.canon_map(o.map_export())) == set(scenes.canon_map(g.map_export()))
    # generic fallback items are not part of the resident kernel: a bucket that produces them ends the launch, they run as a launch of their own
    # and the kernel is launched again at the next bucket (LkResume) - on this young map that happens in every scan
    n_scans, n_relaunch = g.stream_resident_stats()
    print(f"grid-resident kernel, {nb} buckets{' (scattered)' if scattered else ''}: {n_scans} scans, {n_relaunch} launches beyond one per scan (fallback rounds)")
    assert n_scans == 3 and n_relaunch >= 1, (n_scans, n_relaunch)
    if nb == 51:   # at most 32 workgroups: launched as every eighth block of 8 G - where they ran is reported, whatever it was the bits above are equal
        mask = g.stream_grid_placement()
        assert mask != 0, "the one-XCD launch of the grid-resident kernel did not report its XCC ids"
        print(f"grid-resident kernel, 51 buckets: XCC ids of the working blocks {mask:#x} ({'one XCD: barriers without the L2 write-back' if mask & (mask - 1) == 0 else 'several XCDs: full barriers'})")
    for obj in (g, g_seq, o):
        obj.close()


def test_stream_pipeline_forced_conflicts(scene, oracle_lib, hip_lib):
    """The pipelined stream path (insert of bucket k on its own stream beside predict + residual of bucket k+1, verify pass,
    lk_stream.hip `enqueue_bucket_spec`) with the conflicts FORCED: the five 20 000-point buckets of each scan are not azimuth
    sectors but a random partition of the whole scan, so every bucket's insert initialises / refits / cuts planes of root voxels
    that the next bucket's points match - on a young map (sparse first frame), where most leaves are still live.  The verify
    pass must find those tiles and evaluate them again after the insert: counts identical to the oracle on every scan, every state
    entry to 1e-6, the same map - and bit for bit the state and map of the same library with the pipeline switched off."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    g_seq = hip_lib.LegKiloHip(scene.cfg())
    g.stream_pipeline(True)        # off by default (the sequential order measures faster)
    g_seq.stream_pipeline(False)
    t0 = 11.0
    for obj in (o, g, g_seq):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0, dense=20000)
    rng = np.random.default_rng(424242)
    for k in range(3):
        tb = t0 + 0.1 * k
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=9100 + k, seed_noise=9200 + k)
        curv = pts["curvature"].copy()
        pts = pts[rng.permutation(len(pts))]     # the same five time stamps, now scattered over the whole scan
        pts["curvature"] = curv
        po, _ = o.process_scan(pts, tb)
        pg, _ = g.process_scan(pts, tb)
        ps, _ = g_seq.process_scan(pts, tb)
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (pg.n_buckets, pg.n_updates, int(pg.n_effect)) == (5, 5, int(ps.n_effect)), \
            (k, po.n_effect, pg.n_effect, ps.n_effect)
        xo, _ = o.get_state()
        xg, Pg = g.get_state()
        xs, Ps = g_seq.get_state()
        assert np.abs(xo - xg).max() < 1e-6, (k, np.abs(xo - xg).max())
        assert np.array_equal(xg, xs) and np.array_equal(Pg, Ps), (k, np.abs(xg - xs).max())
    n_buckets, n_tiles, n_redo = g.stream_stats()
    assert n_buckets == 15 and n_tiles == 3 * 4 * 313, (n_buckets, n_tiles)   # the first bucket of a scan has nothing to verify
    assert n_redo > 100, n_redo           # the conflicts were really there ...
    assert g_seq.stream_stats()[0] == 0   # ... and the reference run really was sequential
    assert set(scenes.canon_map(o.map_export())) == set(scenes.canon_map(g.map_export()))
    scenes.maps_identical(g.map_export(), g_seq.map_export())
    print("forced conflicts: tiles verified", n_tiles, "evaluated again", n_redo)
    for obj in (g, g_seq, o):
        obj.close()


def test_config2_full_size_residuals(big, hip_lib):
    """Config 2: 100 000 points, one state, residual rows only, against the map the ORACLE built (SURVEY.md 8d:
    'map pre-built by replaying warm-up scans through the oracle') - isolates K1+K2 at full size."""
    scene, o, g, t0 = big
    g2 = hip_lib.LegKiloHip(scene.cfg())
    g2.map_import(o.map_export())
    ts = t0 + 0.3
    pts = synth.dense_scan(scene.world, scenes.Frozen(scene.traj, ts), ts, scene.P, n=100000, n_buckets=1, seed_scan=99)
    xs = synth.initial_state(scene.traj, ts, scene.P)
    _, Ps = o.get_state()
    for obj in (o, g2):
        obj.set_state(xs, Ps)
    xb = scenes.xyz_of(pts)
    ho, zo, Ro, vo = o.residuals(xb)
    hg, zg, Rg, vg = g2.residuals(xb)
    assert vo.sum() > 20000
    # decisions are exact.  The ONE allowance: a point whose gate sits within rounding of its threshold (the device
    # evaluates the 3-sigma gate squared and sigma_l in closed form - DESIGN section 4 - so d vs sigma_num * sqrt(sigma_l) can fall
    # on the other side when the two are equal to ~1e-12).  Such a point is named, its margins on the oracle's side are printed, and
    # the margin must be at rounding level; anything else fails.
    flips = np.flatnonzero(vo != vg)
    for i in flips:
        v, marg = o.residual_margins(xb[i])
        print(f"config2 flip: point {int(i)} oracle valid {int(vo[i])} hip valid {int(vg[i])} margins range/sigma/key {marg}")
        assert min(marg[0], marg[1]) < 1e-9 or marg[2] < 1e-9, (int(i), marg)
    assert len(flips) <= 1, flips
    both_v = (vo & vg).astype(np.uint8)
    scenes.rows_close(hg, zg, Rg, ho, zo, Ro, both_v, rtol=1e-9)
    g2.close()


def test_config2_batch_rows_resident(big, oracle_lib, hip_lib):
    """Config 2 for a device-resident batch (lk_batch_residuals_dev): 12 slots x 100 000 points, every slot under its own state, rows
    materialised in HBM - against the oracle's residual build per slot on the DEVICE's map blob (valid mask exact up to a named
    rounding-level gate flip, rows 1e-9), and against the host entry lk_residuals on the same points (identical bits: the same tile code)."""
    scene, o_big, _, t0 = big
    S, U, n_pts = 12, 4, 100000
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    g.map_import(o_big.map_export())
    g.init_process_cov_q()
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    o.init_process_cov_q()
    o.map_import(g.map_export())
    rng = np.random.default_rng(2202)
    tbs = [t0 + 0.7 + 0.21 * u for u in range(U)]
    scans = [synth.dense_scan(scene.world, scenes.Frozen(scene.traj, tbs[u]), tbs[u], scene.P, n=n_pts, n_buckets=1, seed_scan=2302 + u) for u in range(U)]
    tile = np.arange(S) % U
    xs = np.stack([synth.initial_state(scene.traj, tbs[tile[s]], scene.P, rng, 0.02, 0.5) for s in range(S)])
    Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
    allpts = np.concatenate([scans[u] for u in tile])
    N = S * n_pts
    d_pts = g.device_malloc(allpts.nbytes)
    d_rows, d_v = g.device_malloc(N * 64), g.device_malloc(N)
    g.h2d(d_pts, allpts)
    g.batch_set_priors(xs, Ps)
    g.batch_residuals_dev(d_pts, S, n_pts, d_rows, d_v)
    g.synchronize()
    rows8, v = np.zeros((N, 8)), np.zeros(N, dtype=np.uint8)
    g.d2h(rows8, d_rows)
    g.d2h(v, d_v)
    h6, z, R = np.ascontiguousarray(rows8[:, :6]), np.ascontiguousarray(rows8[:, 6]), np.ascontiguousarray(rows8[:, 7])
    n_flips = 0
    for s in range(S):
        a, b = s * n_pts, (s + 1) * n_pts
        xb = scenes.xyz_of(scans[tile[s]])
        o.set_state(xs[s], Ps[s].reshape(30, 30))
        ho, zo, Ro, vo = o.residuals(xb)
        assert vo.sum() > 20000
        flips = np.flatnonzero(vo != v[a:b])
        for i in flips:
            _, marg = o.residual_margins(xb[i])
            print(f"config2 batch flip: slot {s} point {int(i)} oracle valid {int(vo[i])} hip valid {int(v[a + i])} margins range/sigma/key {marg}")
            assert min(marg[0], marg[1]) < 1e-9 or marg[2] < 1e-9, (s, int(i), marg)
        n_flips += len(flips)
        scenes.rows_close(h6[a:b], z[a:b], R[a:b], ho, zo, Ro, (vo & v[a:b]).astype(np.uint8), rtol=1e-9)
        unm = v[a:b] == 0
        assert not h6[a:b][unm].any() and not z[a:b][unm].any() and not R[a:b][unm].any()   # rows of unmatched points are zero
        if s < 2:   # the host entry on slot 0's state: same tile code, same bits
            g.set_state(xs[s], Ps[s].reshape(30, 30), slot=0)
            hh, zh, Rh, vh = g.residuals(xb)
            assert np.array_equal(vh, v[a:b]) and np.array_equal(hh, h6[a:b]) and np.array_equal(zh, z[a:b]) and np.array_equal(Rh, R[a:b])
            g.batch_set_priors(xs, Ps)
    assert n_flips <= 2, n_flips
    # the same launch with the roots looked up through the hash table instead of the frozen-map grid (LEGKILO_GRID=0: the <true, 0, false> instantiation): the same bits
    os.environ["LEGKILO_GRID"] = "0"
    try:
        g0 = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    finally:
        del os.environ["LEGKILO_GRID"]
    g0.map_import(g.map_export())
    g0.init_process_cov_q()
    d_pts0, d_rows0, d_v0 = g0.device_malloc(allpts.nbytes), g0.device_malloc(N * 64), g0.device_malloc(N)
    g0.h2d(d_pts0, allpts)
    g0.batch_set_priors(xs, Ps)
    g0.batch_residuals_dev(d_pts0, S, n_pts, d_rows0, d_v0)
    g0.synchronize()
    rows0, v0 = np.zeros((N, 8)), np.zeros(N, dtype=np.uint8)
    g0.d2h(rows0, d_rows0)
    g0.d2h(v0, d_v0)
    assert np.array_equal(v0, v) and np.array_equal(rows0, rows8)
    for d in (d_pts0, d_rows0, d_v0):
        g0.device_free(d)
    g0.close()
    for d in (d_pts, d_rows, d_v):
        g.device_free(d)
    g.close()
    o.close()


# ----------------------------------------------------------------------------- batch replay (config 5, reduced)
def test_batch_sort_by_voxel(scene, oracle_lib, hip_lib):
    """lk_batch_sort_by_voxel_dev: every time bucket of every scan of a device-resident batch put into root-voxel order under the slots' prior
    poses.  The buckets come in RANDOM order; afterwards every bucket holds exactly its own points (a permutation: curvature untouched), the 64
    points of a residual tile lie in far fewer root voxels than before, and the replay of the sorted batch equals the oracle's replay of the SAME
    sorted scans (counts exact, state 1e-8) - the order inside a bucket is one of the legal outcomes of the reference's sort (KILO.cc:369)."""
    S, n_pts, nb = 6, 8000, 5
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    g.map_import(blob)
    g.init_process_cov_q()
    o.set_map_insert(False)
    rng = np.random.default_rng(6116)
    xs, Ps, scans = [], [], []
    for s in range(S):
        tb = t0 + 1.2 + 0.37 * s
        sc = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=6116 + s, seed_noise=7227 + s)
        off, dt = synth.buckets_of(sc)
        for b in range(nb):   # random order inside every bucket
            a, e = int(off[b]), int(off[b + 1])
            sc[a:e] = sc[a:e][rng.permutation(e - a)]
        scans.append(sc)
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
    off, dt = synth.buckets_of(scans[0])
    allpts = np.ascontiguousarray(np.concatenate(scans))
    d_in, d_out = g.device_malloc(allpts.nbytes), g.device_malloc(allpts.nbytes)
    g.h2d(d_in, allpts)
    g.batch_set_priors(np.array(xs), np.array(Ps))
    g.batch_sort_by_voxel_dev(d_in, d_out, S, n_pts, off)
    srt = np.empty_like(allpts)
    g.d2h(srt, d_out)
    vs = float(scene.P["voxel_size"])
    E = np.array(scene.P["extrinsic_R"], float).reshape(3, 3)
    T = np.array(scene.P["extrinsic_T"], float)

    def voxels_per_tile(sc_, x_):   # mean number of distinct root voxels among the 64 points of a residual tile
        R, p = x_[:9].reshape(3, 3), x_[9:12]
        cnt = []
        for b in range(nb):
            a, e = int(off[b]), int(off[b + 1])
            xyz = np.stack([sc_["x"][a:e], sc_["y"][a:e], sc_["z"][a:e]], 1).astype(np.float64)
            key = np.floor(((xyz @ E.T + T) @ R.T + p) / vs).astype(np.int64)
            lin = (key[:, 2] * 4096 + key[:, 1]) * 4096 + key[:, 0]
            cnt += [len(np.unique(lin[i:i + 64])) for i in range(0, e - a, 64)]
        return float(np.mean(cnt))

    for s in range(S):
        a_in, a_out = allpts[s * n_pts:(s + 1) * n_pts], srt[s * n_pts:(s + 1) * n_pts]
        for b in range(nb):
            a, e = int(off[b]), int(off[b + 1])
            u_in, u_out = (np.ascontiguousarray(q[a:e]).view(np.uint64).reshape(-1, 2) for q in (a_in, a_out))   # 16-byte records as two words
            assert np.array_equal(u_in[np.lexsort((u_in[:, 1], u_in[:, 0]))], u_out[np.lexsort((u_out[:, 1], u_out[:, 0]))]), (s, b)
        v_in, v_out = voxels_per_tile(a_in, xs[s]), voxels_per_tile(a_out, xs[s])
        assert v_out < 0.7 * v_in, (s, v_in, v_out)   # (1 600 points per bucket: ~2 per voxel, so ~32 voxels per tile is the floor here)
    print(f"distinct root voxels per 64-point tile: {voxels_per_tile(allpts[:n_pts], xs[0]):.1f} in random order, {voxels_per_tile(srt[:n_pts], xs[0]):.1f} sorted")
    poses = g.batch_replay_dev(d_out, S, n_pts, 0.0, off, dt)
    for s in range(S):
        o.set_state(xs[s], Ps[s])
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(srt[s * n_pts:(s + 1) * n_pts], 0.0)
        xo, _ = o.get_state()
        xg, _ = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect), s
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
    # ---- the implicit path (lk_batch_order, default LK_BATCH_ORDER_AUTO): the batch entries themselves keep a voxel-ordered copy of a batch that comes back
    def replays(buf, x_, P_, n):
        out = []
        for _ in range(n):
            g.batch_set_priors(x_, P_)
            p_ = g.batch_replay_dev(buf, S, n_pts, 0.0, off, dt)
            X_, _ = g.batch_get_states(0, S)
            out.append((p_, X_))
        return out

    X_sorted, _ = g.batch_get_states(0, S)
    xa, Pa = np.array(xs), np.array(Ps)
    assert g.batch_order_stats() == (0, 0, 0)          # one replay of d_out so far: only its stamp was taken
    for p_, X_ in replays(d_out, xa, Pa, 3):             # replayed unchanged twice -> examined at the third: already in voxel order, no copy
        assert np.array_equal(X_, X_sorted)
    assert g.batch_order_stats() == (1, 0, 0), g.batch_order_stats()
    r_in = replays(d_in, xa, Pa, 4)                      # the RANDOM-order buffer: as given twice, then sorted once into the library's copy ...
    assert g.batch_order_stats() == (2, 1, 0), g.batch_order_stats()
    assert np.array_equal(r_in[0][1], r_in[1][1]) and not np.array_equal(r_in[0][1], X_sorted)
    for p_, X_ in r_in[2:]:   # ... which is the order the explicit call produced (same keys, same priors, stable sort): the same bits
        assert np.array_equal(X_, X_sorted), np.abs(X_ - X_sorted).max()
        for s in range(S):
            assert (p_[s].n_buckets, p_[s].n_updates, p_[s].n_effect) == (poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect), s
    assert np.abs(r_in[0][1] - X_sorted).max() < 1e-9     # the same scans in another legal order: equal up to the order of sums
    chk = np.empty_like(allpts)
    g.d2h(chk, d_in)
    assert np.array_equal(chk.view(np.uint8), allpts.view(np.uint8))     # the caller's buffer is never written
    # new CONTENT in the sorted batch's buffer (the scans in reverse slot order): noticed on the device - that replay reads the buffer as given, correct at
    # once -, counted again, sorted again at the third replay
    rev = np.ascontiguousarray(np.concatenate(scans[::-1]))
    g.h2d(d_in, rev)
    xr, Pr = np.array(xs[::-1]), np.array(Ps[::-1])
    want = []
    for s in range(S):
        o.set_state(xr[s], Pr[s])
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(rev[s * n_pts:(s + 1) * n_pts], 0.0)
        want.append(((po.n_buckets, po.n_updates, po.n_effect), o.get_state()[0]))
    for rnd, (p_, X_) in enumerate(replays(d_in, xr, Pr, 4)):
        for s in range(S):
            assert want[s][0] == (p_[s].n_buckets, p_[s].n_updates, p_[s].n_effect), (rnd, s)
            assert np.allclose(want[s][1], X_[s], rtol=1e-8, atol=1e-9), (rnd, s, np.abs(want[s][1] - X_[s]).max())
    assert g.batch_order_stats() == (3, 2, 1), g.batch_order_stats()
    # LK_BATCH_ORDER_AS_GIVEN: nothing is looked at
    g.batch_order(0)
    replays(d_in, xr, Pr, 1)
    g.batch_changed()
    assert g.batch_order_stats() == (3, 2, 1)
    g.batch_order(1)
    # lk_batch_prepare_dev: a caller who knows the batch will come back has the copy made at once (works in either mode); the very next replay reads it
    g.h2d(d_in, allpts)
    g.batch_set_priors(xa, Pa)
    g.batch_prepare_dev(d_in, S, n_pts, off)
    assert g.batch_order_stats() == (4, 3, 1), g.batch_order_stats()
    (p_, X_), = replays(d_in, xa, Pa, 1)
    assert np.array_equal(X_, X_sorted)
    g.batch_prepare_dev(d_in, S, n_pts, off)            # prepared already: nothing to do
    assert g.batch_order_stats() == (4, 3, 1)
    g.device_free(d_in)
    g.device_free(d_out)
    with pytest.raises(hip_lib.LegKiloError):   # buckets must cover the scan
        g.batch_sort_by_voxel_dev(d_in, d_out, S, n_pts, off[:-1])
    g.close()
    o.close()


@pytest.mark.parametrize("S", [6, 7, 3])   # 6/7: three slot groups on three HIP streams (even / ragged split); 3: single stream
def test_batch_replay_frozen_map(scene, oracle_lib, hip_lib, S):
    n_pts, nb = 8000, 5
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    g.map_import(blob)
    g.init_process_cov_q()
    o.set_map_insert(False)
    rng = np.random.default_rng(5005)
    xs, Ps, scans, tbs = [], [], [], []
    for s in range(S):
        tb = t0 + 1.2 + 0.37 * s
        scans.append(synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=5005 + s,
                                      seed_noise=6006 + s))
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
        tbs.append(tb)
    off, dt = synth.buckets_of(scans[0])
    for sc in scans:
        o2, d2 = synth.buckets_of(sc)
        assert np.array_equal(o2, off) and np.array_equal(d2, dt)
    # GPU: all scans share t_begin semantics (times are relative): replay each at its own t via separate calls
    # is not needed because only dt matters; use t_begin = 0 on both sides.
    allpts = np.concatenate(scans)
    d_pts = g.device_malloc(allpts.nbytes)
    g.h2d(d_pts, allpts)
    if S == 7:   # priors already resident in HBM (lk_batch_set_priors_dev)
        hx, hP = np.ascontiguousarray(np.array(xs)), np.ascontiguousarray(np.array(Ps).reshape(S, 900))
        d_x, d_P = g.device_malloc(hx.nbytes), g.device_malloc(hP.nbytes)
        g.h2d(d_x, hx)
        g.h2d(d_P, hP)
        g.batch_set_priors_dev(d_x, d_P, S)
    else:
        g.batch_set_priors(np.array(xs), np.array(Ps))
    poses = g.batch_replay_dev(d_pts, S, n_pts, 0.0, off, dt)
    g.device_free(d_pts)
    if S == 7:
        g.device_free(d_x)
        g.device_free(d_P)
    # bulk read-out with covariance (lk_batch_get_states: one gather kernel) == the per-slot calls, bit for bit
    Xall, Pall = g.batch_get_states(0, S)
    for s in range(S):
        o.set_state(xs[s], Ps[s])
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(scans[s], 0.0)
        xo, Po = o.get_state()
        xg, Pg = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect), s
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
        assert np.allclose(np.array(poses[s].pos), xo[9:12], atol=1e-8)
        assert np.array_equal(Xall[s], xg) and np.array_equal(Pall[s], Pg), s
        # the covariance a caller gets per scan (SURVEY 8e record; getRotCov / getPosCov / getVelCov are its diagonal blocks)
        assert np.abs(Pg - Po).max() <= 1e-6 * np.abs(Po).max(), (s, np.abs(Pg - Po).max() / np.abs(Po).max())
        for b0 in (0, 3, 6):
            blk_o, blk_g = Po[b0:b0 + 3, b0:b0 + 3], Pall[s][b0:b0 + 3, b0:b0 + 3]
            assert np.abs(blk_g - blk_o).max() <= 1e-6 * np.abs(blk_o).max(), (s, b0)
    if S == 6:
        # the asynchronous, double-buffered entry: the same batch on slots [0, S) and on [S, 2S) (the second stream),
        # priors armed on the batch's own stream, poses copied to host memory on the stream; nothing synchronises until
        # synchronize().  Both halves must reproduce the synchronous result bit for bit.
        g2 = hip_lib.LegKiloHip(scene.cfg(n_slots=2 * S))
        g2.map_import(blob)
        g2.init_process_cov_q()
        g2.batch_order(0)   # bit for bit against the synchronous replay above, which read the buffer as given (a batch that keeps coming back is otherwise replayed from its voxel-ordered copy)
        hx, hP = np.ascontiguousarray(np.array(xs)), np.ascontiguousarray(np.array(Ps).reshape(S, 900))
        d_x, d_P, d_pts2 = g2.device_malloc(hx.nbytes), g2.device_malloc(hP.nbytes), g2.device_malloc(allpts.nbytes)
        g2.h2d(d_x, hx)
        g2.h2d(d_P, hP)
        g2.h2d(d_pts2, allpts)
        outs = [np.zeros(S, dtype=abi.pose_dtype()) for _ in range(4)]
        for k in range(4):
            g2.batch_replay_async_dev(d_pts2, (k & 1) * S, S, n_pts, 0.0, off, dt, d_x36=d_x, d_P900=d_P, host_out_ptr=outs[k].ctypes.data)
        g2.synchronize()
        ref = np.frombuffer(poses, dtype=abi.pose_dtype())
        for k in range(4):
            for f in ("rot", "pos", "vel", "n_effect", "n_buckets", "n_updates"):
                assert np.array_equal(outs[k][f], ref[f]), (k, f)
        for s in range(S):
            xa, _ = g2.get_state(slot=s)
            xb, _ = g2.get_state(slot=S + s)
            xg, _ = g.get_state(slot=s)
            assert np.array_equal(xa, xg) and np.array_equal(xb, xg), s
        for d in (d_x, d_P, d_pts2):
            g2.device_free(d)
        g2.close()
    g.close()
    o.close()


def test_config5_full_size_batch(big, oracle_lib, hip_lib):
    """Config 5 at its stated per-scan size through the graded entry: 64 filter slots x 100 000 points x 5 buckets of 20 000 on the
    asynchronous batch entry bench.py times (two batches in flight on two slot ranges), against the oracle replaying every slot on
    the DEVICE's map blob: counts exact, positions and rotations to 1e-9.  (16 distinct scans, each replayed from four different
    priors - generating a 100 000-point ray-cast scan costs 0.4 s of host time.)"""
    scene, o_big, _, t0 = big
    S, U, n_pts = 64, 16, 100000
    blob = o_big.map_export()
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=2 * S))
    g.map_import(blob)
    g.init_process_cov_q()
    dev_blob = g.map_export()        # what the device holds (ids may differ from the oracle's export; the voxels are the same)
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    o.init_process_cov_q()
    o.map_import(dev_blob)
    o.set_map_insert(False)
    rng = np.random.default_rng(646464)
    tbs = [t0 + 0.9 + 0.13 * u for u in range(U)]
    scans = [synth.dense_scan(scene.world, scene.traj, tbs[u], scene.P, n=n_pts, n_buckets=5, seed_scan=6405 + u, seed_noise=6506 + u) for u in range(U)]
    off, dt = synth.buckets_of(scans[0])
    tile = np.arange(S) % U
    xs = np.stack([synth.initial_state(scene.traj, tbs[tile[s]], scene.P, rng, 0.02, 0.5) for s in range(S)])
    Ps = np.tile((1e-4 * np.eye(30)).reshape(1, 900), (S, 1))
    allpts = np.concatenate([scans[u] for u in tile])
    d_pts, d_x, d_P = g.device_malloc(allpts.nbytes), g.device_malloc(xs.nbytes), g.device_malloc(Ps.nbytes)
    g.h2d(d_pts, allpts)
    g.h2d(d_x, np.ascontiguousarray(xs))
    g.h2d(d_P, np.ascontiguousarray(Ps))
    outs = [np.zeros(S, dtype=abi.pose_dtype()) for _ in range(2)]
    for k in range(2):
        g.batch_replay_async_dev(d_pts, k * S, S, n_pts, 0.0, off, dt, d_x36=d_x, d_P900=d_P, host_out_ptr=outs[k].ctypes.data)
    g.synchronize()
    for f in ("rot", "pos", "vel", "n_effect", "n_buckets", "n_updates"):
        assert np.array_equal(outs[0][f], outs[1][f]), f
    worst = 0.0
    for s in range(S):
        o.set_state(xs[s], Ps[s].reshape(30, 30))
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(scans[tile[s]], 0.0)
        r = outs[0][s]
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (int(r["n_buckets"]), int(r["n_updates"]), int(r["n_effect"])), (s, po.n_effect, r["n_effect"])
        d = max(float(np.abs(np.array(po.pos) - r["pos"]).max()), float(np.abs(np.array(po.rot) - r["rot"]).max()))
        worst = max(worst, d)
        assert d < 1e-9, (s, d)
    print("config 5 at full size: 64 slots, counts exact, worst pose delta", worst)
    for d in (d_pts, d_x, d_P):
        g.device_free(d)
    g.close()
    o.close()


@pytest.mark.gpu
def test_batch_replay_overlay_follows_the_map(scene, oracle_lib, hip_lib):
    """The overlay replay reads derived structures of the shared map - the frozen-map grid, and a bitmap of the base voxels that are
    frozen leaves (their points are dropped after one bit test) - and both must follow the map: replay a batch, then let the HANDLE'S map
    change through the stream path (three scans with insert: new voxels, leaves that fill up and freeze), then replay the same batch
    again.  Each time every scan must match the oracle on a private copy of the map AS IT IS THEN (counts exact, state 1e-6), and the
    second replay must differ from the first (the map really changed under it)."""
    S, n_pts, nb = 3, 20000, 4
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    t0 = 31.0
    x0 = scenes.init_filter(o, scene, t0)
    scenes.first_frame(o, scene, t0, x0, dense=20000)
    g.map_import(o.map_export())
    g.init_process_cov_q()
    o.map_import(o.map_export())   # both sides from the re-imported form (test_batch_replay_overlay)
    rng = np.random.default_rng(626262)
    xs, Ps, scans = [], [], []
    for s in range(S):
        tb = t0 + 0.4 + 0.17 * s
        scans.append(synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=8105 + s, seed_noise=8206 + s))
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
    off, dt = synth.buckets_of(scans[0])
    allpts = np.concatenate(scans)
    d_pts = g.device_malloc(allpts.nbytes)
    g.h2d(d_pts, allpts)
    g.overlay_reserve(16384, 32768, 16384)

    def replay_and_check(tag):
        blob = g.map_export()
        g.batch_set_priors(np.array(xs), np.array(Ps))
        poses = g.batch_replay_overlay_dev(d_pts, S, n_pts, 0.0, off, dt)
        X, _ = g.batch_get_states(0, S)
        counts = []
        for s in range(S):
            o.map_import(blob)
            o.set_map_insert(True)
            o.set_state(xs[s], Ps[s])
            o.set_times(0.0, 0.0)
            po, _ = o.process_scan(scans[s], 0.0)
            xo, _ = o.get_state()
            assert (po.n_buckets, int(po.n_effect)) == (poses[s].n_buckets, int(poses[s].n_effect)), (tag, s, po.n_effect, poses[s].n_effect)
            assert np.abs(xo - X[s]).max() < 1e-6, (tag, s, np.abs(xo - X[s]).max())
            counts.append(int(po.n_effect))
        return counts

    first = replay_and_check("young map")
    assert len(g.overlay_export(0)) > 0   # an overlay refers to the map of its replay: exportable while that map stands ...
    # the handle's own map moves on: dense scans with insert through the stream path (and through the oracle, only to keep its filter in step)
    for k in range(3):
        tb = t0 + 0.1 * (k + 1)
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=8300 + k, seed_noise=8400 + k)
        g.set_state(synth.initial_state(scene.traj, tb, scene.P), 1e-6 * np.eye(30))
        g.set_times(tb, tb)
        g.process_scan(pts, tb)
        if k == 0:   # ... and refused (LK_ERR_STATE), not served from recycled base blocks, once the map has changed under it
            with pytest.raises(RuntimeError, match="map has changed since the overlay replay"):
                g.overlay_export(0)
    second = replay_and_check("after three more scans")
    assert len(g.overlay_export(0)) > 0
    assert first != second, (first, second)
    print(f"overlay follows the map: n_effect {first} on the young map, {second} after three more scans with insert")
    g.device_free(d_pts)
    g.close()
    o.close()


@pytest.mark.parametrize("case", ["scattered", "sectors", "tiny", "groups"])
def test_batch_replay_overlay(scene, oracle_lib, hip_lib, case):
    """Batch replay WITH the map insert (lk_batch_replay_overlay_dev, SURVEY 8d config 5 "scan-local insert overlay"): every scan
    of the batch runs KILO::process's whole bucket loop - predict, residual, update, re-projection + UpdateVoxelMap per bucket
    (KILO.cc:108-233, :375-395) - on its own copy-on-write overlay of the shared map.  Per slot the checker is the oracle on a
    private copy of that map (lko_map_import of the same blob, insert ON): identical counts, every state entry to 1e-6, and the
    slot's private voxels (lk_overlay_export) equal the oracle's voxels of those keys - tree shape, counters and state bits
    exactly - while every voxel the oracle changed or created is private on the device.  The handle's own map stays untouched.
      scattered: the buckets are a random partition of the scan (every bucket's insert refits / creates planes the next bucket
                 matches) on a young map - the overlay lookup path of the residual pass is what decides the counts;
      sectors:   config 5's shape at full size (100 000 points, 5 azimuth sectors of 20 000);
      tiny:      12 buckets of 40..90 points, one scan of a single bucket's worth of new voxels;
      groups:    200 such scans - from 192 on the replay splits its slots into three groups on three HIP streams (each group sees the
                 pools offset to its first slot); the slots on both sides of the seams and at both ends are checked."""
    if case == "sectors":
        S, n_pts, nb, young = 3, 100000, 5, False
    elif case == "scattered":
        S, n_pts, nb, young = 4, 30000, 5, True
    elif case == "groups":
        S, n_pts, nb, young = 200, 800, 12, True
    else:
        S, n_pts, nb, young = 5, 800, 12, True
    check_slots = list(range(S)) if S <= 8 else [0, S // 3 - 1, S // 3, 2 * S // 3 - 1, 2 * S // 3, S - 1]   # both sides of the seams of three slot groups
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    t0 = 21.0
    x0 = scenes.init_filter(o, scene, t0)
    scenes.first_frame(o, scene, t0, x0, dense=20000 if young else 100000)
    if not young:
        for k in range(3):   # a few full scans with insert: frozen leaves, refitted planes, cut voxels
            tb = t0 + 0.1 * (k + 1)
            o.set_state(synth.initial_state(scene.traj, tb, scene.P), 1e-6 * np.eye(30))
            o.set_times(tb, tb)
            o.process_scan(synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=100000, n_buckets=5, seed_scan=3100 + k, seed_noise=3200 + k), tb)
    # the shared map in the form a blob round trip leaves it in: the oracle's import does not restore the DEAD points of a cut voxel's
    # inner node (never read again; the live oracle still counts them in npts), so the checker, which starts every slot from
    # map_import(blob), and the device, which copies the blob's records, must both start from the re-imported form
    o.map_import(o.map_export())
    blob = o.map_export()
    g.map_import(blob)
    g.init_process_cov_q()
    base = scenes.canon_map(blob)
    rng = np.random.default_rng(515151)
    xs, Ps, scans = [], [], []
    for s in range(S):
        tb = t0 + 0.5 + 0.21 * s
        pts = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=7005 + s, seed_noise=7106 + s)
        if case != "sectors":
            curv = pts["curvature"].copy()
            pts = pts[rng.permutation(len(pts))]
            pts["curvature"] = curv
        scans.append(pts)
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
    off, dt = synth.buckets_of(scans[0])
    for sc in scans:
        o2, d2 = synth.buckets_of(sc)
        assert np.array_equal(o2, off) and np.array_equal(d2, dt)
    allpts = np.concatenate(scans)
    d_pts = g.device_malloc(allpts.nbytes)
    g.h2d(d_pts, allpts)
    g.batch_set_priors(np.array(xs), np.array(Ps))
    frozen = g.batch_replay_dev(d_pts, S, n_pts, 0.0, off, dt)
    n_eff_frozen = [int(p.n_effect) for p in frozen]
    if case == "scattered":
        # pools too small for what a scan touches: a loud LK_ERR_CAPACITY that names the slot and the sizes, never a fault; the
        # handle stays usable.  The fresh pools are poisoned (LEGKILO_POISON_POOLS): a kernel that follows a record which the overflowing
        # slot never got to write must meet garbage here, not the zeros a young process's allocator happens to hand out (round 5: the
        # thread-per-point geometry pass did exactly that and faulted in ONE test order only)
        poison_before = os.environ.get("LEGKILO_POISON_POOLS")
        os.environ["LEGKILO_POISON_POOLS"] = "1"
        g.overlay_reserve(64, 128, 64)
        g.batch_set_priors(np.array(xs), np.array(Ps))
        try:
            with pytest.raises(hip_lib.LegKiloError, match="overlay pool overflow"):
                g.batch_replay_overlay_dev(d_pts, S, n_pts, 0.0, off, dt)
        finally:
            if poison_before is None:
                os.environ.pop("LEGKILO_POISON_POOLS", None)
        g.overlay_reserve(16384, 32768, 16384)   # a young map: a scattered 30 000-point scan touches most of its voxels
    if case == "groups":
        g.overlay_reserve(2048, 4096, 2048)
    g.batch_set_priors(np.array(xs), np.array(Ps))
    poses = g.batch_replay_overlay_dev(d_pts, S, n_pts, 0.0, off, dt)
    Xall, Pall = g.batch_get_states(0, S)
    # the shared map is untouched, and a second replay gives the same bits (the overlays start empty every time)
    assert np.array_equal(g.map_export(), np.frombuffer(blob, dtype=np.uint8)) or scenes.maps_identical(g.map_export(), blob)
    g.batch_set_priors(np.array(xs), np.array(Ps))
    poses2 = g.batch_replay_overlay_dev(d_pts, S, n_pts, 0.0, off, dt)
    X2, P2 = g.batch_get_states(0, S)
    assert np.array_equal(Xall, X2) and np.array_equal(Pall, P2)
    assert [int(p.n_effect) for p in poses] == [int(p.n_effect) for p in poses2]
    g.device_free(d_pts)
    differs = 0
    for s in check_slots:
        o.map_import(blob)
        o.set_map_insert(True)
        o.set_state(xs[s], Ps[s])
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(scans[s], 0.0)
        xo, Po = o.get_state()
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (poses[s].n_buckets, poses[s].n_updates, int(poses[s].n_effect)), \
            (case, s, po.n_buckets, po.n_updates, po.n_effect, poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect, n_eff_frozen[s])
        assert np.abs(xo - Xall[s]).max() < 1e-6, (case, s, np.abs(xo - Xall[s]).max())
        assert np.abs(Pall[s] - Po).max() <= 1e-6 * np.abs(Po).max(), (case, s)
        st = scenes.compare_overlay(g.overlay_export(s), base, scenes.canon_map(o.map_export()), (case, s), rtol=1e-5, ptol=1e-7)
        assert st["private_roots"] > 0 and st["changed_roots"] > 0, (case, s, st)
        differs += int(int(po.n_effect) != n_eff_frozen[s])
        print(f"overlay {case} slot {s}: n_effect {int(po.n_effect)} (frozen map: {n_eff_frozen[s]}), private roots {st['private_roots']}, "
              f"changed by the oracle {st['changed_roots']}, nodes compared {st.get('nodes', 0)}, max |dx| {np.abs(xo - Xall[s]).max():.2e}")
    if case == "scattered":   # the insert really changed what later buckets matched
        assert differs == S, (differs, S)
    print("overlay high-water marks (roots, nodes, blocks):", g.overlay_stats())
    g.close()
    o.close()


@pytest.mark.parametrize("mode", ["plain", "imu", "kin"])
def test_batch_replay_overlay_ragged_scan_resident(scene, oracle_lib, hip_lib, mode, monkeypatch):
    """The scan-resident form of lk_batch_replay_overlay_ragged_dev (every bucket <= 512 points: ONE launch carries a scan through its whole bucket
    chain with the insert; a scan whose bucket leaves items for the insert's fallback code stops there, the fallback launch runs, the scans go on):
    on a YOUNG map - voxels are created, cut and refitted all the time, so scans do stop - the same bits as the launch-by-launch form
    (LEGKILO_RAG_RESIDENT=0: states, covariances, every private voxel), and per slot the oracle's KILO::process on a private copy of the map:
    counts exact, state 1e-6, private voxels equal.  Shapes: config-1 scans, a scan of 40 buckets of ~150 points, a one-point scan."""
    if mode == "kin":
        sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3), **CAPS)
        o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
    else:
        sc = scene
        o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=True)
    t0 = 2.0
    x0 = scenes.init_filter(o, sc, t0)
    scenes.first_frame(o, sc, t0, x0)   # a first frame only: most of what the scans see is new
    o.map_import(o.map_export())
    blob = o.map_export()
    base = scenes.canon_map(blob)
    rng = np.random.default_rng(717171)
    shapes = [None, (6000, 40), "clutter", (1, 1), None, "clutter"]
    scans, tbs, xs, Ps, msgs = [], [], [], [], []
    for s, shp in enumerate(shapes):
        tb = t0 + 0.4 + 0.23 * s
        x_prior = synth.initial_state(sc.traj, tb, sc.P, rng, 0.02, 0.5)
        if shp is None:
            pts = scenes.vlp_scan_input(sc, tb, 180 + s)
        elif shp == "clutter":
            # a config-1 scan with volumetric clutter in front of the robot in 30 of its buckets: voxels that are NOT planes - cut down to layer 2,
            # leftover points, the insert's fallback items (what stops a scan in the resident launch); the scan's own points keep the state well determined
            pts = scenes.vlp_scan_input(sc, tb, 180 + s)
            R0, p0 = x_prior[:9].reshape(3, 3), x_prior[9:12]
            eR, eT = np.asarray(sc.P["extrinsic_R"], dtype=np.float64).reshape(3, 3), np.asarray(sc.P["extrinsic_T"], dtype=np.float64)
            pw = scenes.corner_clutter(rng, n_cells=30, per_cell=60, origin=tuple(p0 + np.array([1.5, -1.0, -0.2])))
            pb = ((pw - p0) @ R0 - eT) @ eR
            cl = np.zeros(len(pb), dtype=synth.POINT_DTYPE)
            cl["x"], cl["y"], cl["z"] = pb[:, 0], pb[:, 1], pb[:, 2]
            stamps = np.unique(pts["curvature"])
            cl["curvature"] = stamps[((np.arange(len(pb)) // 60) * (len(stamps) // 31)) % len(stamps)]   # 60 at a time, in 30 of the scan's buckets
            pts = np.concatenate([pts, cl])
            pts = pts[np.argsort(pts["curvature"], kind="stable")]
        else:
            pts = synth.dense_scan(sc.world, sc.traj, tb, sc.P, n=shp[0], n_buckets=shp[1], seed_scan=7600 + s, seed_noise=7700 + s)
        assert np.diff(synth.buckets_of(pts)[0].astype(np.int64)).max() <= 512
        scans.append(pts)
        tbs.append(tb)
        xs.append(x_prior)
        Ps.append(1e-4 * np.eye(30))
        if mode == "imu":
            msgs.append(synth.imu_stream(sc.traj, tb, tb + 0.1, seed=9500 + s))
        elif mode == "kin":
            msgs.append(synth.kin_stream(sc.traj, tb, tb + 0.1, sc.P, seed=9600 + s))
    S = len(scans)
    g = hip_lib.LegKiloHip(sc.cfg(n_slots=S))
    g.map_import(blob)
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    o.set_acc_norm(9.81)
    kw = {"imus": msgs} if mode == "imu" else {"kins": msgs} if mode == "kin" else {}
    monkeypatch.delenv("LEGKILO_RAG_RESIDENT", raising=False)
    poses = g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)
    rounds = g.overlay_resident_rounds()
    Xall, Pall = g.batch_get_states(0, S)
    exports = [g.overlay_export(s) for s in range(S)]
    assert rounds >= 2, f"no scan stopped for the fallback launch ({rounds} launch): the protocol is not exercised"
    monkeypatch.delenv("LEGKILO_RAG_RESIDENT", raising=False)
    for s in range(S):
        o.map_import(blob)
        o.set_map_insert(True)
        o.set_state(xs[s], Ps[s])
        o.set_times(tbs[s], tbs[s])
        okw = {"imus": msgs[s]} if mode == "imu" else {"kins": msgs[s]} if mode == "kin" else {}
        po, _ = o.process_scan(scans[s], tbs[s], **okw)
        xo, Po = o.get_state()
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (poses[s].n_buckets, poses[s].n_updates, int(poses[s].n_effect)), \
            (mode, s, po.n_buckets, po.n_updates, po.n_effect, poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect)
        xtol = 1e-6
        assert np.abs(xo - Xall[s]).max() < xtol, (mode, s, np.abs(xo - Xall[s]).max())
        assert np.abs(Pall[s] - Po).max() <= xtol * np.abs(Po).max(), (mode, s)
        # (stored points: the state's 1e-6 times the lever arm of a voxel 25 m away)
        st = scenes.compare_overlay(exports[s], base, scenes.canon_map(o.map_export()), (mode, s), rtol=1e-4, ptol=2e-6)
        print(f"scan-resident overlay {mode} slot {s}: {len(scans[s])} points, {po.n_buckets} buckets, n_effect {int(po.n_effect)}, private roots {st['private_roots']}, "
              f"max |dx| {np.abs(xo - Xall[s]).max():.2e}; {rounds} launches")
    if mode == "plain":
        # pools far too small for what a scan touches, poisoned: the resident launch leaves the scan at its overflow, the call fails loudly with
        # LK_ERR_CAPACITY (no fault, no hang), and the handle replays the same batch to the same bits once the pools may grow again
        monkeypatch.setenv("LEGKILO_POISON_POOLS", "1")
        g.overlay_reserve(64, 128, 64)
        with pytest.raises(hip_lib.LegKiloError, match="overlay pool overflow"):
            g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)
        monkeypatch.delenv("LEGKILO_POISON_POOLS", raising=False)
        g.overlay_reserve(0, 0, 0)
        g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)
        assert g.overlay_resident_rounds() == rounds
        Xr, Pr = g.batch_get_states(0, S)
        assert np.array_equal(Xall, Xr) and np.array_equal(Pall, Pr)
    monkeypatch.setenv("LEGKILO_RAG_RESIDENT", "0")
    poses0 = g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)
    assert g.overlay_resident_rounds() == 0
    X0, P0 = g.batch_get_states(0, S)
    assert np.array_equal(Xall, X0) and np.array_equal(Pall, P0), "scan-resident and launch-by-launch replay differ"
    for s in range(S):
        assert (poses0[s].n_buckets, poses0[s].n_updates, int(poses0[s].n_effect)) == (poses[s].n_buckets, poses[s].n_updates, int(poses[s].n_effect))
        assert scenes.maps_identical(g.overlay_export(s), exports[s]), (mode, s)
    g.close()
    o.close()


@pytest.mark.parametrize("mode", ["plain", "imu", "kin"])
def test_batch_replay_overlay_ragged(scene, oracle_lib, hip_lib, mode):
    """lk_batch_replay_overlay_ragged_dev: a recorded run's scans - every scan its own size, time buckets and start time, optionally its IMU
    (only_imu_use) or kinematic + IMU (leg fusion, KILO.cc:379-390) messages between the buckets - replayed WITH the map insert, each scan on
    its own copy-on-write overlay.  Per slot the oracle runs KILO::process on that scan alone on a private copy of the map (blob re-imported,
    insert ON): identical bucket / update / match counts, state to 1e-6, every private voxel equal to the oracle's and every voxel the
    oracle changed private on the device.  The shapes cover config-1 scans (hundreds of two-ms buckets of a dozen points), dense scans with
    few large buckets, one-point and one-bucket scans; the insert really matters (the frozen-map replay of the same batch matches less)."""
    if mode == "kin":
        sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3), **CAPS)
        o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
    else:
        sc = scene
        o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=True)
    t0 = 2.0
    x0 = scenes.init_filter(o, sc, t0)
    scenes.first_frame(o, sc, t0, x0)
    scenes.replay_vlp(o, sc, t0, 3, use_kin=(mode == "kin"))
    o.map_import(o.map_export())   # the form a blob round trip leaves the map in (see test_batch_replay_overlay)
    blob = o.map_export()
    base = scenes.canon_map(blob)
    rng = np.random.default_rng(606060)
    shapes = [None, (6000, 4), None, (1, 1), (2500, 1), None] if mode == "plain" else [None, None, (3000, 3), None]
    scans, tbs, xs, Ps, msgs = [], [], [], [], []
    for s, shp in enumerate(shapes):
        tb = t0 + 0.6 + 0.17 * s
        if shp is None:
            pts = scenes.vlp_scan_input(sc, tb, 80 + s)
        else:
            pts = synth.dense_scan(sc.world, sc.traj, tb, sc.P, n=shp[0], n_buckets=shp[1], seed_scan=8600 + s, seed_noise=8700 + s)
        scans.append(pts)
        tbs.append(tb)
        xs.append(synth.initial_state(sc.traj, tb, sc.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
        if mode == "imu":
            msgs.append(synth.imu_stream(sc.traj, tb, tb + 0.1, seed=9300 + s))
        elif mode == "kin":
            msgs.append(synth.kin_stream(sc.traj, tb, tb + 0.1, sc.P, seed=9400 + s))
    S = len(scans)
    g = hip_lib.LegKiloHip(sc.cfg(n_slots=S))
    g.map_import(blob)
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    o.set_acc_norm(9.81)
    kw = {"imus": msgs} if mode == "imu" else {"kins": msgs} if mode == "kin" else {}
    frozen = g.batch_replay_ragged(scans, tbs, xs, Ps, host_tables=True, **kw) if all(np.diff(synth.buckets_of(p_)[0].astype(np.int64)).max() <= 512 for p_ in scans) or mode == "plain" else None
    poses = g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)
    Xall, Pall = g.batch_get_states(0, S)
    assert np.array_equal(g.map_export(), np.frombuffer(blob, dtype=np.uint8)) or scenes.maps_identical(g.map_export(), blob)   # the shared map is untouched
    poses2 = g.batch_replay_overlay_ragged(scans, tbs, xs, Ps, **kw)   # the overlays start empty every time: the same bits again
    X2, P2 = g.batch_get_states(0, S)
    assert np.array_equal(Xall, X2) and np.array_equal(Pall, P2)
    differs = 0
    for s in range(S):
        o.map_import(blob)
        o.set_map_insert(True)
        o.set_state(xs[s], Ps[s])
        o.set_times(tbs[s], tbs[s])
        okw = {"imus": msgs[s]} if mode == "imu" else {"kins": msgs[s]} if mode == "kin" else {}
        po, _ = o.process_scan(scans[s], tbs[s], **okw)
        xo, Po = o.get_state()
        assert (po.n_buckets, po.n_updates, int(po.n_effect)) == (poses[s].n_buckets, poses[s].n_updates, int(poses[s].n_effect)), \
            (mode, s, po.n_buckets, po.n_updates, po.n_effect, poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect)
        assert (poses2[s].n_buckets, poses2[s].n_updates, int(poses2[s].n_effect)) == (poses[s].n_buckets, poses[s].n_updates, int(poses[s].n_effect))
        assert np.abs(xo - Xall[s]).max() < 1e-6, (mode, s, np.abs(xo - Xall[s]).max())
        assert np.abs(Pall[s] - Po).max() <= 1e-6 * np.abs(Po).max(), (mode, s)
        st = scenes.compare_overlay(g.overlay_export(s), base, scenes.canon_map(o.map_export()), (mode, s), rtol=1e-5, ptol=1e-7)
        if len(scans[s]) > 100:
            assert st["private_roots"] > 0 and st["changed_roots"] > 0, (mode, s, st)
        if frozen is not None:
            differs += int(int(po.n_effect) != int(frozen[s].n_effect))
        print(f"overlay ragged {mode} slot {s}: {len(scans[s])} points, {po.n_buckets} buckets, n_effect {int(po.n_effect)}"
              + (f" (frozen map: {int(frozen[s].n_effect)})" if frozen is not None else "") + f", private roots {st['private_roots']}, max |dx| {np.abs(xo - Xall[s]).max():.2e}")
    if frozen is not None:
        assert differs >= 1, "the insert changed nothing any later bucket matched"
    if mode != "plain":   # without the messages the outcome differs (they are really applied)
        g.batch_replay_overlay_ragged(scans[:1], tbs[:1], xs[:1], Ps[:1])
        assert not np.array_equal(g.get_state(slot=0)[0], Xall[0])
    g.close()
    o.close()


def test_frozen_grid_equals_hash_and_follows_the_map(scene, oracle_lib, hip_lib, monkeypatch):
    """Batch replay looks root voxels up through the frozen-map grid (dense array of root records + flattened subtree lists,
    rebuilt when the map changes) and runs the kernel specialised for ext_R == I; LEGKILO_GRID=0 / LEGKILO_XID=0 keep the hash
    table and the generic kernel.  On a map WITH cut voxels (clutter: non-plane roots whose planes sit at layers 1 and 2) all
    variants must give the same bits and equal the oracle; after the map has changed (new voxels inserted through the stream
    path) a second replay must see the new map - on every variant."""
    n_pts, nb, S = 6000, 3, 5
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0, n_scans=4)
    xs0, Ps0 = o.get_state()
    # clutter around the robot's path: corners of 0.25 m cells -> layer-0 and layer-1 nodes fail the plane test, planes at layer 2
    rngc = np.random.default_rng(77)
    centre = scene.traj.pos(t0 + 1.3)
    clutter = scenes.corner_clutter(rngc, n_cells=40, per_cell=80, origin=(centre[0] + 2.0, centre[1] - 1.0, 3.0))
    var = np.tile((1e-4 * np.eye(3)).reshape(1, 9), (len(clutter), 1))
    o.map_update(clutter, var)
    blob = o.map_export()
    cm = scenes.canon_map(blob)
    assert sum(1 for n in cm.values() if not n["is_plane"] and n["children"]) >= 3, "the map must contain cut voxels"
    o.set_map_insert(False)
    rng = np.random.default_rng(991)
    xs, Ps, scans = [], [], []
    for s in range(S):
        tb = t0 + 1.2 + 0.05 * s
        scans.append(synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=7100 + s, seed_noise=7200 + s))
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
    # points ON the clutter (body frame of scan 0's prior), so that the flattened lists are really walked
    R0, p0 = xs[0][:9].reshape(3, 3), xs[0][9:12]
    E = np.array(scene.P["extrinsic_R"], float).reshape(3, 3)
    T = np.array(scene.P["extrinsic_T"], float)
    cb = ((clutter[:n_pts // nb] - p0) @ R0 - T) @ E
    scans[0]["x"][:len(cb)], scans[0]["y"][:len(cb)], scans[0]["z"][:len(cb)] = cb[:, 0], cb[:, 1], cb[:, 2]
    off, dt = synth.buckets_of(scans[0])
    allpts = np.concatenate(scans)
    want = []
    for s in range(S):
        o.set_state(xs[s], Ps[s])
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(scans[s], 0.0)
        want.append((po.n_buckets, po.n_updates, po.n_effect, o.get_state()[0].copy()))
    results = {}
    for name, env in (("grid+xid", {}), ("grid", {"LEGKILO_XID": "0"}), ("hash", {"LEGKILO_GRID": "0"})):
        for k in ("LEGKILO_XID", "LEGKILO_GRID"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
        g.map_import(blob)
        g.init_process_cov_q()
        d_pts = g.device_malloc(allpts.nbytes)
        g.h2d(d_pts, allpts)
        g.batch_set_priors(np.array(xs), np.array(Ps))
        poses = g.batch_replay_dev(d_pts, S, n_pts, 0.0, off, dt)
        first = [(poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect, g.get_state(slot=s)[0].copy()) for s in range(S)]
        # the map changes: a stream-mode scan with insert from a pose the map has not seen (new roots, new planes) ...
        tb = t0 + 2.5
        g.set_state(synth.initial_state(scene.traj, tb, scene.P), 1e-6 * np.eye(30), slot=0)
        g.set_times(tb, tb)
        roots_before = g.map_stats()[0]
        g.process_scan(synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=20000, n_buckets=2, seed_scan=7300, seed_noise=7301), tb)
        assert g.map_stats()[0] > roots_before
        # ... and the next replay of the same batch runs against the NEW map
        g.batch_set_priors(np.array(xs), np.array(Ps))
        poses2 = g.batch_replay_dev(d_pts, S, n_pts, 0.0, off, dt)
        second = [(poses2[s].n_buckets, poses2[s].n_updates, poses2[s].n_effect, g.get_state(slot=s)[0].copy()) for s in range(S)]
        results[name] = (first, second)
        g.device_free(d_pts)
        g.close()
    for k in ("LEGKILO_XID", "LEGKILO_GRID"):
        monkeypatch.delenv(k, raising=False)
    ref_first, ref_second = results["hash"]
    for s in range(S):
        assert ref_first[s][:3] == want[s][:3], (s, ref_first[s][:3], want[s][:3])
        assert np.allclose(ref_first[s][3], want[s][3], rtol=1e-8, atol=1e-9), (s, np.abs(ref_first[s][3] - want[s][3]).max())
    assert want[0][2] > 200, "scan 0 must match points in the cut voxels"
    for name in ("grid", "grid+xid"):
        for which, (a, b) in (("first", (results[name][0], ref_first)), ("second", (results[name][1], ref_second))):
            for s in range(S):
                assert a[s][:3] == b[s][:3], (name, which, s, a[s][:3], b[s][:3])
                assert np.array_equal(a[s][3], b[s][3]), (name, which, s, np.abs(a[s][3] - b[s][3]).max())
    assert any(ref_first[s][2] != ref_second[s][2] for s in range(S)), "the inserted scan must have changed what the batch matches"
    o.close()


def test_batch_replay_ragged(scene, oracle_lib, hip_lib, monkeypatch):
    """lk_batch_replay_ragged_dev: scans of different sizes, bucket tables and start times in one batch (what a recorded
    run looks like) - each must come out as the oracle's own bucket loop over that scan alone; on equally shaped scans the
    ragged entry reproduces the uniform one bit for bit."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    o.set_map_insert(False)
    rng = np.random.default_rng(8118)
    shapes = [(8000, 5), (3000, 3), None, (65, 1), (5000, 7), (1, 1)]   # None: a config-1 scan (VLP-16, ~370 two-ms buckets)
    scans, tbs, xs, Ps = [], [], [], []
    for s, shp in enumerate(shapes):
        tb = t0 + 1.1 + 0.23 * s
        if shp is None:
            sc = scenes.vlp_scan_input(scene, tb, 40 + s)
        else:
            sc = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=shp[0], n_buckets=shp[1], seed_scan=8200 + s, seed_noise=8300 + s)
        scans.append(sc)
        tbs.append(tb)
        xs.append(synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
    S = len(scans)
    nbs = [len(synth.buckets_of(sc)[1]) for sc in scans]
    assert max(nbs) > 100 and min(nbs) == 1, nbs
    g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
    g.map_import(blob)
    g.init_process_cov_q()
    poses = g.batch_replay_ragged(scans, tbs, xs, Ps, host_tables=True)
    for s in range(S):
        o.set_state(xs[s], Ps[s])
        o.set_times(tbs[s], tbs[s])
        po, _ = o.process_scan(scans[s], tbs[s])
        xo, Po = o.get_state()
        xg, Pg = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect), (s, nbs[s])
        assert po.n_buckets == nbs[s]
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
        assert np.allclose(Po, Pg, rtol=1e-6, atol=1e-11), (s, np.abs(Po - Pg).max())
    # the same batch with the bucket tables built ON THE DEVICE (lk_batch_replay_scans_dev: runs of equal curvature found by a flag /
    # scan / scatter pass, CSR tables): bit-identical to the host-table entry, dense scans (per-bucket-index launches) included
    first = [(poses[s].n_buckets, poses[s].n_updates, poses[s].n_effect, g.get_state(slot=s)) for s in range(S)]
    allp = np.ascontiguousarray(np.concatenate(scans))
    so = np.r_[0, np.cumsum([len(sc) for sc in scans])]
    d_all = g.device_malloc(allp.nbytes)
    g.h2d(d_all, allp)
    g.batch_set_priors(np.asarray(xs), np.asarray(Ps))
    pd = g.batch_replay_scans_dev(d_all, so, tbs)
    for s in range(S):
        assert (pd[s].n_buckets, pd[s].n_updates, pd[s].n_effect) == first[s][:3], (s, first[s][:3])
        xd, Pd = g.get_state(slot=s)
        assert np.array_equal(xd, first[s][3][0]) and np.array_equal(Pd, first[s][3][1]), s
    g.device_free(d_all)
    # small buckets only (<= 512 points each): the entry runs each scan's whole bucket chain as ONE wave in one launch
    # (lk_scan_wave_kernel); it must equal the per-bucket launches (LEGKILO_RAGGED_LEVELS=1) bit for bit, and the oracle
    small = [i for i, sc in enumerate(scans) if np.diff(synth.buckets_of(sc)[0].astype(np.int64)).max() <= 512]
    small_scans = [scans[i] for i in small] + [scenes.vlp_scan_input(scene, t0 + 3.0 + 0.1 * k, 60 + k) for k in range(2)]
    small_tb = [tbs[i] for i in small] + [t0 + 3.0 + 0.1 * k for k in range(2)]
    small_x = [xs[i] for i in small] + [synth.initial_state(scene.traj, t0 + 3.0 + 0.1 * k, scene.P, rng, 0.02, 0.5) for k in range(2)]
    small_P = [1e-4 * np.eye(30)] * len(small_scans)
    assert len(small_scans) >= 4 and sum(len(synth.buckets_of(sc)[1]) > 100 for sc in small_scans) >= 3
    res = []
    for levels in (False, True):
        if levels:
            monkeypatch.setenv("LEGKILO_RAGGED_LEVELS", "1")
        ps = g.batch_replay_ragged(small_scans, small_tb, small_x, small_P, host_tables=True)
        res.append(([g.get_state(slot=s) for s in range(len(small_scans))], [(p.n_buckets, p.n_updates, p.n_effect) for p in ps]))
        if levels:
            monkeypatch.delenv("LEGKILO_RAGGED_LEVELS")
    assert res[0][1] == res[1][1], (res[0][1], res[1][1])
    for s in range(len(small_scans)):
        assert np.array_equal(res[0][0][s][0], res[1][0][s][0]) and np.array_equal(res[0][0][s][1], res[1][0][s][1]), s
        o.set_state(small_x[s], small_P[s])
        o.set_times(small_tb[s], small_tb[s])
        po, _ = o.process_scan(small_scans[s], small_tb[s])
        xo, Po = o.get_state()
        assert (po.n_buckets, po.n_updates, po.n_effect) == res[0][1][s], (s, res[0][1][s])
        assert np.allclose(xo, res[0][0][s][0], rtol=1e-8, atol=1e-9), (s, np.abs(xo - res[0][0][s][0]).max())
        assert np.allclose(Po, res[0][0][s][1], rtol=1e-6, atol=1e-11), s
    # the same scans with their IMU messages applied between the buckets (only_imu_use mode, KILO.cc:379-383)
    g.set_acc_norm(9.81)
    o.set_acc_norm(9.81)
    small_imus = [synth.imu_stream(scene.traj, tb_, tb_ + 0.1, seed=8600 + s) for s, tb_ in enumerate(small_tb)]
    ps = g.batch_replay_ragged(small_scans, small_tb, small_x, small_P, imus=small_imus, host_tables=True)
    for s in range(len(small_scans)):
        o.set_state(small_x[s], small_P[s])
        o.set_times(small_tb[s], small_tb[s])
        po, _ = o.process_scan(small_scans[s], small_tb[s], imus=small_imus[s])
        xo, Po = o.get_state()
        xg, Pg = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (ps[s].n_buckets, ps[s].n_updates, ps[s].n_effect), s
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
        assert np.allclose(Po, Pg, rtol=1e-6, atol=1e-11), (s, np.abs(Po - Pg).max())
        if po.n_buckets > 100:
            assert not np.array_equal(xg, res[0][0][s][0])   # the IMU updates did change the outcome
    with_imu = [g.get_state(slot=s) for s in range(len(small_scans))]
    allp = np.ascontiguousarray(np.concatenate(small_scans))
    so = np.r_[0, np.cumsum([len(sc) for sc in small_scans])]
    d_all = g.device_malloc(allp.nbytes)
    g.h2d(d_all, allp)
    g.batch_set_priors(np.asarray(small_x), np.asarray(small_P))
    pd = g.batch_replay_scans_dev(d_all, so, small_tb, imus=small_imus)      # device-built tables + IMU messages
    for s in range(len(small_scans)):
        assert (pd[s].n_buckets, pd[s].n_updates, pd[s].n_effect) == (ps[s].n_buckets, ps[s].n_updates, ps[s].n_effect), s
        xd, Pd = g.get_state(slot=s)
        assert np.array_equal(xd, with_imu[s][0]) and np.array_equal(Pd, with_imu[s][1]), s
    with pytest.raises(hip_lib.LegKiloError):
        g.batch_replay_scans_dev(d_all, np.r_[so[:-1], so[-2]], small_tb)     # an empty scan is refused
    # a scan that skipped the time sort of KILO.cc:367 is refused before any slot is touched (a NaN stamp counts as unsorted)
    for spoil in ("swap", "nan"):
        bad = allp.copy()
        sb = next(i for i, sc in enumerate(small_scans) if len(synth.buckets_of(sc)[1]) > 100)
        j = int(so[sb]) + 5
        k = j + int(np.nonzero(bad["curvature"][j:so[sb + 1]] != bad["curvature"][j])[0][0])
        if spoil == "swap":
            bad[[j, k]] = bad[[k, j]]
        else:
            bad["curvature"][k] = np.nan
        g.h2d(d_all, bad)
        before = g.get_state(slot=1)
        with pytest.raises(hip_lib.LegKiloError, match="not sorted by time"):
            g.batch_replay_scans_dev(d_all, so, small_tb)
        after = g.get_state(slot=1)
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    g.device_free(d_all)
    # a process noise with off-diagonal terms (the scan-resident wave has a fast path for the diagonal Q of initProcessCovQ)
    Qn = o.get_Q().copy()
    Qn[6, 7] = Qn[7, 6] = 3.0
    Qn[18, 21] = Qn[21, 18] = 40.0
    o.set_Q(Qn)
    g.set_Q(Qn)
    ps = g.batch_replay_ragged(small_scans[:3], small_tb[:3], small_x[:3], small_P[:3], host_tables=True)
    for s in range(3):
        o.set_state(small_x[s], small_P[s])
        o.set_times(small_tb[s], small_tb[s])
        po, _ = o.process_scan(small_scans[s], small_tb[s])
        xo, Po = o.get_state()
        xg, Pg = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (ps[s].n_buckets, ps[s].n_updates, ps[s].n_effect), s
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
        assert np.allclose(Po, Pg, rtol=1e-6, atol=1e-11), (s, np.abs(Po - Pg).max())
        if po.n_buckets > 1:   # a one-bucket scan starts at its bucket's time: dt = 0, Q does not enter
            assert not np.array_equal(Pg, res[0][0][s][1])
    o.init_process_cov_q()
    g.init_process_cov_q()
    # equally shaped scans: ragged == uniform, bit for bit
    uni = [synth.dense_scan(scene.world, scene.traj, t0 + 2.0 + 0.1 * s, scene.P, n=4000, n_buckets=5, seed_scan=8400 + s, seed_noise=8500 + s)
           for s in range(3)]
    off, dt = synth.buckets_of(uni[0])
    allpts = np.concatenate(uni)
    d_pts = g.device_malloc(allpts.nbytes)
    g.h2d(d_pts, allpts)
    g.batch_set_priors(np.array(xs[:3]), np.array(Ps[:3]))
    g.batch_order(0)   # both entries on the buffer AS GIVEN (the uniform entry would otherwise replay its voxel-ordered copy: same scans, another order of sums)
    g.batch_replay_dev(d_pts, 3, 4000, 0.0, off, dt)
    g.batch_order(1)
    ref = [g.get_state(slot=s) for s in range(3)]
    g.batch_set_priors(np.array(xs[:3]), np.array(Ps[:3]))
    g.batch_replay_ragged_dev(d_pts, g.ragged_tables([0, 4000, 8000, 12000], [off] * 3, [dt] * 3, [0.0] * 3))
    for s in range(3):
        xr, Pr = g.get_state(slot=s)
        assert np.array_equal(xr, ref[s][0]) and np.array_equal(Pr, ref[s][1]), s
    # malformed tables are refused
    with pytest.raises(hip_lib.LegKiloError):
        g.batch_replay_ragged_dev(d_pts, g.ragged_tables([0, 4000, 8000, 12001], [off] * 3, [dt] * 3, [0.0] * 3))
    g.device_free(d_pts)
    g.close()
    o.close()


def test_batch_replay_ragged_leg_fusion(oracle_lib, hip_lib):
    """Recorded-run replay in the reference's DEFAULT mode (only_imu_use: false, KILO.cc:384-390): every scan's 500 Hz kinematic +
    IMU messages are applied between its time buckets (predictUpdateKinImu, KILO.cc:260-314; updateByKinImu, eskf.cc:137-145:
    6 IMU rows + 3 rows per foot in contact, 6..18 rows) inside the one-wave-per-scan kernel (wave_kin_update_core).  diter.yaml
    parameters.  Each scan equals the oracle's process_scan(..., kins=) on that scan alone: counts exact, x to 1e-8, P to 1e-6;
    the message mix covers 0..4 contacts (M = 6..18)."""
    sc = scenes.Scene(params=dict(config.DITER, voxel_grid_resolution=0.3), **CAPS)
    o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=False)
    t0 = 2.0
    x0 = scenes.init_filter(o, sc, t0)
    scenes.first_frame(o, sc, t0, x0)
    scenes.replay_vlp(o, sc, t0, 4, use_kin=True)
    blob = o.map_export()
    o.set_map_insert(False)
    rng = np.random.default_rng(4242)
    S = 6
    scans, tbs, xs, Ps, kins = [], [], [], [], []
    seen_m = set()
    for s in range(S):
        tb = t0 + 0.5 + 0.13 * s
        scans.append(scenes.vlp_scan_input(sc, tb, 70 + s))
        tbs.append(tb)
        xs.append(synth.initial_state(sc.traj, tb, sc.P, rng, 0.02, 0.5))
        Ps.append(1e-4 * np.eye(30))
        k = synth.kin_stream(sc.traj, tb, tb + 0.1, sc.P, seed=9100 + s)
        if s == 1:
            k["contact"][::3] = 1          # all four feet down on every third message (M = 18)
        if s == 2:
            k["contact"][::4] = 0          # flight phase: IMU rows only (M = 6)
            k["contact"][1::4] = [1, 0, 0, 0]
        seen_m |= set(int(6 + 3 * c.sum()) for c in (k["contact"] != 0))
        kins.append(k)
    assert seen_m >= {6, 9, 12, 18}, seen_m
    g = hip_lib.LegKiloHip(sc.cfg(n_slots=S))
    g.map_import(blob)
    g.init_process_cov_q()
    g.set_acc_norm(9.81)
    o.set_acc_norm(9.81)
    ps = g.batch_replay_ragged(scans, tbs, xs, Ps, kins=kins, host_tables=True)
    plain = None
    for s in range(S):
        o.set_state(xs[s], Ps[s])
        o.set_times(tbs[s], tbs[s])
        po, _ = o.process_scan(scans[s], tbs[s], kins=kins[s])
        xo, Po = o.get_state()
        xg, Pg = g.get_state(slot=s)
        assert (po.n_buckets, po.n_updates, po.n_effect) == (ps[s].n_buckets, ps[s].n_updates, ps[s].n_effect), s
        assert po.n_buckets > 100 and po.n_effect > 500
        assert np.allclose(xo, xg, rtol=1e-8, atol=1e-9), (s, np.abs(xo - xg).max())
        assert np.allclose(Po, Pg, rtol=1e-6, atol=1e-11), (s, np.abs(Po - Pg).max())
        if s == 0:
            plain = xg.copy()
    # device-built bucket tables + kinematic messages: the same bits
    host_tab = [g.get_state(slot=s) for s in range(S)]
    allp = np.ascontiguousarray(np.concatenate(scans))
    so = np.r_[0, np.cumsum([len(sc_) for sc_ in scans])]
    d_all = g.device_malloc(allp.nbytes)
    g.h2d(d_all, allp)
    g.batch_set_priors(np.asarray(xs), np.asarray(Ps))
    pd = g.batch_replay_scans_dev(d_all, so, tbs, kins=kins)
    for s in range(S):
        assert (pd[s].n_buckets, pd[s].n_updates, pd[s].n_effect) == (ps[s].n_buckets, ps[s].n_updates, ps[s].n_effect), s
        xd, Pd = g.get_state(slot=s)
        assert np.array_equal(xd, host_tab[s][0]) and np.array_equal(Pd, host_tab[s][1]), s
    g.device_free(d_all)
    # without the messages the outcome differs (they are really applied)
    g.batch_replay_ragged(scans[:1], tbs[:1], xs[:1], Ps[:1])
    assert not np.array_equal(g.get_state(slot=0)[0], plain)
    # IMU and kinematic messages together are refused, and so are dense buckets with messages
    with pytest.raises(AssertionError):
        g.ragged_tables([0, len(scans[0])], [synth.buckets_of(scans[0])[0]], [synth.buckets_of(scans[0])[1]], [tbs[0]],
                        imus=[synth.imu_stream(sc.traj, tbs[0], tbs[0] + 0.1)], kins=kins[:1])
    dense = synth.dense_scan(sc.world, sc.traj, tbs[0], sc.P, n=4000, n_buckets=2, seed_scan=1)
    with pytest.raises(hip_lib.LegKiloError, match="512"):
        g.batch_replay_ragged([dense], tbs[:1], xs[:1], Ps[:1], kins=kins[:1], host_tables=True)
    with pytest.raises(hip_lib.LegKiloError, match="512"):
        g.batch_replay_ragged([dense], tbs[:1], xs[:1], Ps[:1], kins=kins[:1])
    g.close()
    o.close()


def test_batch_update_kernels_agree_on_edge_buckets(scene, oracle_lib, hip_lib, monkeypatch):
    """Batch replay has two implementations of update(k) + predict(k+1): the single-wave kernel (default) and the
    256-thread one (LEGKILO_UPDATE_CLASSIC=1).  They must agree BIT FOR BIT, also on buckets that match nothing (no
    update, predict only) and on buckets with exactly one match (the +1e-4 branch of eskf.cc:98-104), and both must
    reproduce the oracle."""
    S, n_pts, nb = 4, 4000, 5
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    t0 = 1.0
    blob = mature_oracle_map(o, scene, t0)
    o.set_map_insert(False)
    tb = t0 + 1.3
    base = synth.dense_scan(scene.world, scene.traj, tb, scene.P, n=n_pts, n_buckets=nb, seed_scan=7117, seed_noise=7227)
    off, dt = synth.buckets_of(base)
    rng = np.random.default_rng(7337)
    x0 = synth.initial_state(scene.traj, tb, scene.P, rng, 0.02, 0.5)
    P0 = 1e-4 * np.eye(30)
    far = base.copy()
    fxyz = (rng.uniform(-1, 1, (n_pts, 3)) + np.array([300.0, -200.0, 50.0])).astype(np.float32)
    far["x"], far["y"], far["z"] = fxyz[:, 0], fxyz[:, 1], fxyz[:, 2]
    # one matching point per bucket, the rest far away: candidates that match under the prior, deep inside their gate
    o.set_state(x0, P0)
    valid = o.residuals(scenes.xyz_of(base))[3].astype(bool)
    single = far.copy()
    for b in range(nb):
        cand = np.flatnonzero(valid[off[b]:off[b + 1]])
        assert len(cand) > 10
        k = off[b] + cand[len(cand) // 2]
        for f in ("x", "y", "z"):
            single[f][off[b]] = base[f][k]
    mixed = base.copy()
    mixed[: off[2]] = far[: off[2]]          # two empty buckets, then three normal ones
    scans = [base, far, single, mixed]
    allpts = np.concatenate(scans)
    states = []
    for classic in ("1", "0"):
        monkeypatch.setenv("LEGKILO_UPDATE_CLASSIC", classic)
        g = hip_lib.LegKiloHip(scene.cfg(n_slots=S))
        monkeypatch.delenv("LEGKILO_UPDATE_CLASSIC")
        g.map_import(blob)
        g.init_process_cov_q()
        d_pts = g.device_malloc(allpts.nbytes)
        g.h2d(d_pts, allpts)
        g.batch_set_priors(np.array([x0] * S), np.array([P0] * S))
        poses = g.batch_replay_dev(d_pts, S, n_pts, 0.0, off, dt)
        states.append(([g.get_state(slot=s) for s in range(S)], [(p.n_buckets, p.n_updates, p.n_effect) for p in poses]))
        g.device_free(d_pts)
        g.close()
    (st_c, cnt_c), (st_w, cnt_w) = states
    assert cnt_c == cnt_w, (cnt_c, cnt_w)
    for s in range(S):
        assert np.array_equal(st_c[s][0], st_w[s][0]) and np.array_equal(st_c[s][1], st_w[s][1]), s
    assert cnt_w[1] == (nb, 0, 0), cnt_w[1]                  # nothing matched: five predicts, no update
    assert cnt_w[2][1] >= 1 and cnt_w[2][2] == cnt_w[2][1], cnt_w[2]   # every updating bucket of the single scan had N == 1
    assert cnt_w[3][1] == nb - 2, cnt_w[3]
    for s in range(S):
        o.set_state(x0, P0)
        o.set_times(0.0, 0.0)
        po, _ = o.process_scan(scans[s], 0.0)
        xo, Po = o.get_state()
        assert (po.n_buckets, po.n_updates, po.n_effect) == cnt_w[s], (s, cnt_w[s])
        assert np.allclose(xo, st_w[s][0], rtol=1e-8, atol=1e-9), (s, np.abs(xo - st_w[s][0]).max())
        assert np.allclose(Po, st_w[s][1], rtol=1e-6, atol=1e-11), (s, np.abs(Po - st_w[s][1]).max())
    o.close()


def test_block_recycling_and_capacity_errors(scene, oracle_lib, hip_lib):
    """Point blocks retired by freezes / cuts are re-used from the next bucket on: the pool's high-water mark stays
    near the number of LIVE blocks instead of growing with every leaf that ever existed; exhausting a pool is a loud
    LK_ERR_CAPACITY, never silent corruption."""
    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    t0 = 1.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    scenes.replay_vlp(o, scene, t0, 10)
    scenes.replay_vlp(g, scene, t0, 10)
    bo = abi.parse_blob(o.map_export())
    live = int((bo["nodes"]["block"] >= 0).sum())
    ever = int(((bo["nodes"]["state"] & abi.LK_NODE_INIT_OCTO) > 0).sum())  # leaves that held a block at some time
    high_water = g.map_stats()[2]
    print(f"live blocks {live}, initialised nodes {ever}, GPU pool high-water {high_water}")
    assert high_water <= live + 0.35 * ever + 64, (high_water, live, ever)
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-5, ptol=1e-6)
    g.close()
    # a pool that is too small for the first frame fails loudly
    small = hip_lib.LegKiloHip(scene.cfg(max_point_blocks=64))
    x0 = scenes.init_filter(small, scene, t0)
    with pytest.raises(hip_lib.LegKiloError, match="overflow"):
        scenes.first_frame(small, scene, t0, x0)
    small.close()
    tiny = hip_lib.LegKiloHip(scene.cfg(max_scan_points=256))
    with pytest.raises(hip_lib.LegKiloError):
        tiny.residuals(np.zeros((1000, 3), dtype=np.float32))
    with pytest.raises(hip_lib.LegKiloError):
        tiny.set_state(np.zeros(36), None, slot=5)
    tiny.close()
    o.close()


def test_checkpoint_resume_is_bit_identical(scene, hip_lib, tmp_path):
    """Save (state + map blob) mid-run, restore into a FRESH handle, continue: every later state is bit-identical to
    the uninterrupted run (also a run-to-run determinism check: node / block ids differ, results must not)."""
    from legkilo_amd import checkpoint

    a = hip_lib.LegKiloHip(scene.cfg())
    t0 = 1.0
    x0 = scenes.init_filter(a, scene, t0)
    # acc_norm_ is |mean acc| of the initialisation window in a real run (KILO.cc:349), not 9.81: the checkpoint has to
    # carry what the HANDLE holds (the IMU rows are scaled by gravity / acc_norm, KILO.cc:246)
    a.set_acc_norm(9.6317)
    scenes.first_frame(a, scene, t0, x0)
    scenes.replay_vlp(a, scene, t0, 4)
    checkpoint.save(tmp_path / "ck.npz", a)
    b = hip_lib.LegKiloHip(scene.cfg())
    checkpoint.restore(tmp_path / "ck.npz", b)
    assert b.get_acc_norm() == 9.6317
    ra = scenes.replay_vlp(a, scene, t0, 4, start=4)
    rb = scenes.replay_vlp(b, scene, t0, 4, start=4)
    for k, ((pa, xa), (pb, xb)) in enumerate(zip(ra, rb)):
        assert (pa.n_buckets, pa.n_updates, pa.n_effect) == (pb.n_buckets, pb.n_updates, pb.n_effect), k
        assert np.array_equal(xa, xb), (k, np.abs(xa - xb).max())
    _, Pa = a.get_state()
    _, Pb = b.get_state()
    assert np.array_equal(Pa, Pb)
    scenes.compare_maps(a.map_export(), b.map_export(), rtol=0.0, ptol=0.0)
    a.close()
    b.close()


def test_map_import_rejects_malformed_blobs(scene, hip_lib):
    """lk_map_import / lk_map_import_dev: a truncated blob, counts that do not add up to the size, ids that point outside the
    pools and a blob written with another voxel size are refused with LK_ERR_INVALID before anything is used."""
    import torch

    a = hip_lib.LegKiloHip(scene.cfg())
    t0 = 1.0
    x0 = scenes.init_filter(a, scene, t0)
    scenes.first_frame(a, scene, t0, x0)
    blob = np.asarray(a.map_export(), dtype=np.uint8).copy()
    hd = blob[: abi.blob_header_dtype().itemsize].view(abi.blob_header_dtype())[0]
    b = hip_lib.LegKiloHip(scene.cfg())
    with pytest.raises(hip_lib.LegKiloError):
        b.map_import(blob[:-64])                      # truncated
    bad = blob.copy()
    bad[: abi.blob_header_dtype().itemsize].view(abi.blob_header_dtype())["n_nodes"] += 3   # counts exceed the bytes
    with pytest.raises(hip_lib.LegKiloError, match="size does not match"):
        b.map_import(bad)
    bad = blob.copy()
    off_nodes = abi.blob_header_dtype().itemsize + int(hd["n_roots"]) * 16
    bad[off_nodes: off_nodes + 4].view(np.int32)[0] = int(hd["n_nodes"]) + 7          # child[0] of node 0 out of range
    with pytest.raises(hip_lib.LegKiloError, match="child id"):
        b.map_import(bad)
    P2 = dict(scene.P)
    P2["voxel_size"] = 0.25
    c = hip_lib.LegKiloHip(config.make_config(P2, **CAPS))
    with pytest.raises(hip_lib.LegKiloError, match="voxel_size"):
        c.map_import(blob)
    # device-resident blob: same configuration check, ids validated on the device
    nbytes = a.map_export_dev_size()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    a.map_export_dev(d.data_ptr(), nbytes)
    with pytest.raises(hip_lib.LegKiloError, match="voxel_size"):
        c.map_import_dev(d.data_ptr(), nbytes)
    b.map_import_dev(d.data_ptr(), nbytes)            # the intact blob loads
    assert b.map_stats() == a.map_stats()
    hsz = abi.blob_header_dtype().itemsize
    # corrupt a child id inside the node pool of the device blob: header | hash table (8 x max_roots slots of 16 B) | nodes ...
    node0 = hsz + 16 * (1 << int(np.ceil(np.log2(8 * int(scene.cfg().max_roots)))))
    d[node0: node0 + 4] = torch.from_numpy(np.array([2 ** 30], dtype=np.int32).view(np.uint8)).cuda()
    with pytest.raises(hip_lib.LegKiloError, match="out of range"):
        b.map_import_dev(d.data_ptr(), nbytes)
    assert b.map_stats() == (0, 0, 0)
    for hnd in (a, b, c):
        hnd.close()


def count_tree(blob_bytes):
    """(nodes, point blocks) reachable from the roots of a blob."""
    cm = scenes.canon_map(blob_bytes)

    def walk(n):
        nn, nb = 1, 1 if n["pts"] is not None else 0
        for c in n["children"].values():
            a, b = walk(c)
            nn, nb = nn + a, nb + b
        return nn, nb

    tot = [walk(n) for n in cm.values()]
    return sum(t[0] for t in tot), sum(t[1] for t in tot)


def test_map_sliding_parity_and_compaction(scene, oracle_lib, hip_lib, tmp_path):
    """mapSliding / clearMemOutOfMap (voxel_map.cc:552-594).  Same decision, same surviving map as
    the oracle; the device pools are COMPACTED to the live map; the path keeps running on the compacted pools with
    results identical to the oracle's; last_slide_position travels with a checkpoint."""
    from legkilo_amd import checkpoint

    o = oracle_lib.Oracle(scene.cfg(), imu_mode_only=True)
    g = hip_lib.LegKiloHip(scene.cfg())
    t0 = 1.0
    for obj in (o, g):
        x0 = scenes.init_filter(obj, scene, t0)
        scenes.first_frame(obj, scene, t0, x0)
    ro = scenes.replay_vlp(o, scene, t0, 3)
    rg = scenes.replay_vlp(g, scene, t0, 3)
    roots0, nodes0, blocks0 = g.map_stats()
    pos = rg[-1][1][9:12]
    # below the threshold (distance from the origin): no slide on either side
    far = float(np.linalg.norm(pos)) + 1.0
    assert o.map_slide(pos, sliding_thresh=far, half_map_size=4) == (False, 0)
    assert g.map_slide(pos, sliding_thresh=far, half_map_size=4) == (False, 0)
    assert g.map_stats() == (roots0, nodes0, blocks0)
    # slide: keep +-8 voxels (4 m) around the robot
    so = o.map_slide(ro[-1][1][9:12], sliding_thresh=0.0, half_map_size=8)
    sg = g.map_slide(pos, sliding_thresh=0.0, half_map_size=8)
    assert so == sg and sg[0] and 0 < sg[1] < roots0, (so, sg, roots0)
    assert np.allclose(g.get_last_slide_position(), pos, rtol=0, atol=0)
    blob_g = g.map_export()
    scenes.compare_maps(o.map_export(), blob_g, rtol=1e-6, ptol=1e-7)
    roots1, nodes1, blocks1 = g.map_stats()
    live_nodes, live_blocks = count_tree(blob_g)
    assert roots1 == roots0 - sg[1]
    assert (nodes1, blocks1) == (live_nodes, live_blocks), "pools must be compacted to the live map"
    assert nodes1 < nodes0
    # a second call inside the threshold does nothing
    assert o.map_slide(ro[-1][1][9:12], sliding_thresh=1.0, half_map_size=8) == (False, 0)
    assert g.map_slide(pos, sliding_thresh=1.0, half_map_size=8) == (False, 0)
    # the path keeps running on the compacted pools (new roots / children / blocks are bump-allocated again)
    ro2 = scenes.replay_vlp(o, scene, t0, 3, start=3)
    rg2 = scenes.replay_vlp(g, scene, t0, 3, start=3)
    for k, ((po, xo), (pg, xg)) in enumerate(zip(ro2, rg2)):
        assert (po.n_buckets, po.n_updates, po.n_effect) == (pg.n_buckets, pg.n_updates, pg.n_effect), k
        assert np.allclose(xo, xg, rtol=1e-7, atol=1e-8), (k, np.abs(xo - xg).max())
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-6, ptol=1e-6)
    # clearMemOutOfMap with an explicit, asymmetric box
    k = np.floor(pos / 0.5).astype(int)
    box = (k[0] + 3, k[0] - 6, k[1] + 5, k[1] - 2, k[2] + 8, k[2] - 8)
    assert o.map_clear_outside(*box) == g.map_clear_outside(*box)
    scenes.compare_maps(o.map_export(), g.map_export(), rtol=1e-6, ptol=1e-6)
    # checkpoint carries last_slide_position
    checkpoint.save(tmp_path / "ck.npz", g)
    b = hip_lib.LegKiloHip(scene.cfg())
    checkpoint.restore(tmp_path / "ck.npz", b)
    assert np.array_equal(b.get_last_slide_position(), g.get_last_slide_position())
    b.close()
    # everything removed: an empty map is a valid map
    roots_left = g.map_stats()[0]
    assert g.map_clear_outside(-1000, -1001, 0, 0, 0, 0) == roots_left
    assert g.map_stats() == (0, 0, 0)
    g.close()
    o.close()


def test_no_device_fallback_is_loud(hip_lib, scene):
    bad = scene.cfg(device_id=99)
    with pytest.raises(hip_lib.LegKiloError):
        hip_lib.LegKiloHip(bad)
