"""Golden fixture tests/golden/path_small.npz (made by tests/golden/make_golden.py with the oracle).

CPU : the oracle must keep reproducing it (guards the restatement against accidental edits).
GPU : the HIP path, through the C-ABI, must reproduce it from the stored inputs alone."""
import os

import numpy as np
import pytest

import scenes
from legkilo_amd import abi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "path_small.npz")
CAPS = dict(max_roots=1 << 14, max_nodes=1 << 15, max_point_blocks=1 << 14, max_scan_points=1 << 15)


def replay(obj, g, P0=1e-6):
    """The stored call sequence, identical for the oracle and the HIP library."""
    obj.set_state(g["x0"], P0 * np.eye(30))
    obj.init_process_cov_q()
    obj.set_acc_norm(9.81)
    t0 = float(g["t0"])
    obj.set_times(t0, t0)
    obj.map_build(g["build_world"], g["build_body"])
    po = np.r_[0, np.cumsum(g["seq_len"])]
    io = np.r_[0, np.cumsum(g["seq_imu_len"])]
    out = []
    for k in range(len(g["seq_len"])):
        pose, _ = obj.process_scan(g["seq_pts"][po[k]:po[k + 1]], float(g["seq_tb"][k]), imus=g["seq_imus"][io[k]:io[k + 1]])
        x, _ = obj.get_state()
        out.append((pose, x))
    return out


def check(obj, g, tol_state, tol_rows, reimport=False):
    out = replay(obj, g)
    for k, (pose, x) in enumerate(out):
        assert [pose.n_buckets, pose.n_updates, pose.n_effect] == list(g["seq_counts"][k]), k
        assert np.allclose(x, g["seq_x"][k], rtol=0, atol=tol_state), (k, np.abs(x - g["seq_x"][k]).max())
    scenes.compare_maps(g["map_blob"], obj.map_export(), rtol=1e-5, ptol=max(tol_state, 1e-9))
    if reimport:
        # isolate the per-bucket kernels from the (tiny, but plane-fit-amplified) drift of the 4-scan replay:
        # continue from the golden map and the golden state themselves
        obj.map_import(g["map_blob"])
        obj.set_state(g["xs"], g["Ps"])
        obj.set_times(float(g["times"][0]), float(g["times"][1]))
    h6, z, R, valid = obj.residuals(g["q_body"])
    assert np.array_equal(valid, g["q_valid"]), int((valid != g["q_valid"]).sum())
    scenes.rows_close(h6, z, R, g["q_h6"], g["q_z"], g["q_R"], valid, rtol=tol_rows)
    w, inten, ne = obj.update_points(float(g["bk_t"]), g["bk_body"])
    assert ne == int(g["bk_n_effect"]) and np.array_equal(inten, g["bk_intensity"])
    assert np.abs(w - g["bk_world"]).max() < 1e-5
    x1, P1 = obj.get_state()
    assert np.allclose(x1, g["x1"], rtol=0, atol=min(tol_state, 1e-9) if reimport else tol_state), np.abs(x1 - g["x1"]).max()
    assert np.abs(P1 - g["P1"]).max() <= 1e-6 * np.abs(g["P1"]).max()


def test_oracle_reproduces_golden(oracle_lib):
    g = np.load(G)
    sc = scenes.Scene(**CAPS)
    o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=True)
    check(o, g, tol_state=1e-11, tol_rows=1e-11)
    assert o.map_stats() == int(g["n_roots_after"])
    o.close()


@pytest.mark.gpu
def test_hip_reproduces_golden(hip_lib):
    g = np.load(G)
    sc = scenes.Scene(**CAPS)
    h = hip_lib.LegKiloHip(sc.cfg())
    check(h, g, tol_state=1e-6, tol_rows=1e-9, reimport=True)
    assert h.map_stats()[0] == int(g["n_roots_after"])
    h.close()


@pytest.mark.gpu
def test_host_mirror_example_runs(hip_lib, oracle_lib, tmp_path):
    """The C++ mirror of the reference classes (leg-kilo_amd/host) drives the same library end to end, and what it computes -
    BuildVoxelMap, one predictUpdatePoint bucket, a two-scan recorded-run replay - equals the oracle on the SAME inputs (the
    example dumps its clouds and results): match count exact, state to 1e-7, replay poses to 1e-7."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "lk_host_example_gpu")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"), "-I", os.path.join(root, "leg-kilo_amd", "host"),
                        os.path.join(root, "leg-kilo_amd", "host", "example_kilo_path.cc"), "-o", exe, "-L", os.path.join(root, "leg-kilo_amd"),
                        "-llegkilo_hip", "-Wl,-rpath," + os.path.join(root, "leg-kilo_amd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    dump = str(tmp_path / "lk_host_example_dump.bin")   # per-test directory: parallel pytest workers do not collide
    r = subprocess.run([exe, dump], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-1500:])
    from legkilo_amd import abi, config

    raw = open(dump, "rb").read()
    n, nb, n_succ, n_poses = np.frombuffer(raw, dtype=np.uint32, count=4)
    o_ = 16
    world = np.frombuffer(raw, dtype=np.float32, count=3 * n, offset=o_).reshape(n, 3)
    o_ += 12 * n
    body = np.frombuffer(raw, dtype=np.float32, count=3 * n, offset=o_).reshape(n, 3)
    o_ += 12 * n
    x_hip = np.frombuffer(raw, dtype=np.float64, count=36, offset=o_)
    o_ += 288
    poses = np.frombuffer(raw, dtype=abi.pose_dtype(), count=n_poses, offset=o_)
    P = dict(config.LEG_FUSION)     # the example's ESKF::Config / VoxelMapConfig literals are leg_fusion.yaml's; extrinsic T = (0,0,0.2)
    o = oracle_lib.Oracle(config.make_config(P), imu_mode_only=True)
    x0 = np.zeros(36)
    x0[:9] = np.eye(3).reshape(9)
    x0[9:12] = [0, 0, 0.5]
    x0[21:24] = [0, 0, -9.81]
    o.set_state(x0, 1e-6 * np.eye(30))
    o.init_process_cov_q()
    o.set_times(0.0, 0.0)
    o.map_build(world, body)
    _, _, ne = o.update_points(0.01, body[:nb])
    xo, Po = o.get_state()
    assert int(ne) == int(n_succ) > 500, (ne, n_succ)
    assert np.abs(xo - x_hip).max() < 1e-7, np.abs(xo - x_hip).max()
    # the two-scan recorded-run replay: same bucket as two 2-ms buckets, from (posterior, P_post) and from (x0, P0), frozen map
    pts = np.zeros(nb, dtype=synth_point_dtype())
    pts["x"], pts["y"], pts["z"] = body[:nb, 0], body[:nb, 1], body[:nb, 2]
    pts["curvature"] = np.where(np.arange(nb) < nb // 2, 0.0, 0.002).astype(np.float32)
    o.set_map_insert(False)
    for k, (xs, Ps, tb) in enumerate([(xo, Po, 0.02), (x0, 1e-6 * np.eye(30), 0.05)]):
        o.set_state(xs, Ps)
        o.set_times(tb, tb)
        po, _ = o.process_scan(pts, tb)
        assert (int(po.n_buckets), int(po.n_effect)) == (int(poses[k]["n_buckets"]), int(poses[k]["n_effect"])), k
        assert np.abs(np.array(po.pos) - poses[k]["pos"]).max() < 1e-7 and np.abs(np.array(po.rot) - poses[k]["rot"]).max() < 1e-7, k
    # replayWithInsert: the same two scans, each with KILO::process's insert after every bucket on its OWN copy of the map
    assert n_poses == 4
    o.set_map_insert(True)
    blob = o.map_export()
    for k, (xs, Ps) in enumerate([(xo, Po), (x0, 1e-6 * np.eye(30))]):
        o.map_import(blob)
        o.set_state(xs, Ps)
        o.set_times(0.02, 0.02)
        po, _ = o.process_scan(pts, 0.02)
        assert (int(po.n_buckets), int(po.n_effect)) == (int(poses[2 + k]["n_buckets"]), int(poses[2 + k]["n_effect"])), (k, po.n_effect, poses[2 + k]["n_effect"])
        dpos, drot = np.abs(np.array(po.pos) - poses[2 + k]["pos"]).max(), np.abs(np.array(po.rot) - poses[2 + k]["rot"]).max()
        print(f"replayWithInsert scan {k}: n_effect {int(po.n_effect)}, |dpos| {dpos:.2e}, |drot| {drot:.2e}")
        assert dpos < 1e-6 and drot < 1e-6, (k, dpos, drot)   # the bar of test_batch_replay_overlay: the second bucket matches planes the first one refitted
    o.close()


def synth_point_dtype():
    from legkilo_amd import synth

    return synth.POINT_DTYPE


# ----------------------------------------------------------------------------- golden vectors made BY THE REFERENCE
GR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kilo_small.npz")
GR4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kilo_config4.npz")


def replay_ref_golden(obj, g, mode):
    """tests/golden/make_golden_ref.py's call sequence (KILO::process of the reference build, oracle/_ref)."""
    obj.set_state(g[f"{mode}_x0"], 1e-6 * np.eye(30))
    obj.init_process_cov_q()
    obj.set_acc_norm(9.81)
    t0 = float(g[f"{mode}_t0"])
    obj.set_times(t0, t0)
    obj.map_build(g[f"{mode}_build_world"], g[f"{mode}_build_body"])
    po = np.r_[0, np.cumsum(g[f"{mode}_len"])]
    ao = np.r_[0, np.cumsum(g[f"{mode}_aux_len"])]
    for k in range(len(g[f"{mode}_len"])):
        pts, aux = g[f"{mode}_pts"][po[k]:po[k + 1]], g[f"{mode}_aux"][ao[k]:ao[k + 1]]
        kw = dict(imus=aux) if mode == "imu" else dict(kins=aux)
        pose, _ = obj.process_scan(pts, float(g[f"{mode}_tb"][k]), **kw)
        x, _ = obj.get_state()
        yield k, pose, x


def check_ref_golden(make, tol_state, tol_cov, ptol):
    from legkilo_amd import config

    for mode in ("imu", "kin", "c4"):
        # "c4": config 4 at its stated shape - diter.yaml as is, Ouster 64 x 1024 scans, leg fusion (make_golden_ref.py: run_config4)
        g = np.load(GR4 if mode == "c4" else GR)
        params = {"imu": None, "kin": dict(config.DITER, voxel_grid_resolution=0.3), "c4": config.DITER}[mode]
        sc = scenes.Scene(params=params, **CAPS)
        obj = make(sc, mode)
        for k, pose, x in replay_ref_golden(obj, g, mode):
            assert int(pose.n_effect) == int(g[f"{mode}_n_effect"][k]), (mode, k, pose.n_effect, g[f"{mode}_n_effect"][k])
            assert np.allclose(x, g[f"{mode}_x"][k], rtol=0, atol=tol_state), (mode, k, np.abs(x - g[f"{mode}_x"][k]).max())
        _, P = obj.get_state()
        assert np.abs(P - g[f"{mode}_P"]).max() <= tol_cov * np.abs(g[f"{mode}_P"]).max(), mode
        tp, tu = obj.get_times()
        assert (tp, tu) == tuple(g[f"{mode}_times"]), mode
        if mode == "imu":
            scenes.compare_maps(g["imu_map_blob"], obj.map_export(), rtol=1e-5, ptol=ptol)
        obj.close()


def test_oracle_reproduces_what_the_reference_computed(oracle_lib):
    """The oracle against outputs of the reference's own KILO::process (no reference tree needed to run this)."""
    check_ref_golden(lambda sc, mode: oracle_lib.Oracle(sc.cfg(), imu_mode_only=(mode == "imu")), 1e-8, 1e-6, 1e-7)


@pytest.mark.gpu
def test_hip_reproduces_what_the_reference_computed(hip_lib):
    """The HIP path, through the C-ABI, against outputs of the reference's own KILO::process: identical match counts
    on every scan, states to 1e-6, the same map."""
    check_ref_golden(lambda sc, mode: hip_lib.LegKiloHip(sc.cfg()), 1e-6, 1e-5, 1e-6)
