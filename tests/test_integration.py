"""The drop-in boundary exercised FROM THE REFERENCE'S SIDE.

tests/integration/kilo_hip.ed is the reference-side patch as line-range edits (no reference text in it): KILO.h / KILO.cc keep their
structure - YAML parsing, first-frame initialisation through state_initial.hpp, pcl::VoxelGrid, the time sort, the bucket loop with its
message queues - and lose the bodies of predictUpdatePoint / predictUpdateImu / predictUpdateKinImu (KILO.cc:108-314) to one call each;
core/slam/eskf.h and voxel_map.h forward to leg-kilo_amd/host/legkilo_host_eigen.hpp; eskf.cc and voxel_map.cc are not compiled.  The
result links against liblegkilo_hip.so only.

CPU : the patch still applies to the reference tree (sha256 of the four files), the patched sources compile and link, the library needs
      nothing of eskf.cc / voxel_map.cc, and without a GPU KILO's constructor fails loudly (no fallback behind the patched class).
GPU : that library's KILO::process replays the inputs of tests/golden/ref_kilo_small.npz / ref_kilo_config4.npz - outputs of the UNPATCHED
      build of the same class - and gives the golden match counts exactly and the golden states to 1e-6, in IMU, kinematic + IMU and
      config-4 mode, with the reference's own bucket loop (one C-ABI call per bucket / message) and with the one-call scan."""
import os
import subprocess
import sys

import numpy as np
import pytest

import scenes

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "integration"))
import kilo_patch  # noqa: E402
import refhip  # noqa: E402

REF_SRC = refhip.REF_SRC
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "core", "slam", "KILO.cc")), reason="the reference tree is not on this machine")


@needs_reference
def test_patch_applies_and_holds_no_reference_text(tmp_path):
    out = kilo_patch.apply(REF_SRC, str(tmp_path))
    assert len(out) == 4 and all(os.path.getsize(p) > 0 for p in out)
    patched = open(os.path.join(tmp_path, "core", "slam", "KILO.cc"), encoding="utf-8").read()
    for gone in ("build_single_residual(", "->updateByPoints(", "->UpdateVoxelMap(", "ki_h.block", "voxel_map_.find", "eskf_->", "map_manager_->"):
        assert gone not in patched, gone   # KILO.cc:108-314's bodies are calls into the library now
    assert "path_->predictUpdatePoint(" in patched and "state_initial_->processing(measure, path_->eskf())" in patched
    # an edit script carries only NEW text: none of its longer lines may be a line of the files it edits
    spec = kilo_patch.parse(open(kilo_patch.ED, encoding="utf-8").read())
    for rel, (_, script) in spec.items():
        ref_lines = {l.strip() for l in open(os.path.join(REF_SRC, rel), encoding="utf-8").read().split("\n") if len(l.strip()) > 30}
        copied = [l for l in script if l.strip() in ref_lines]
        assert not copied, (rel, copied)


@needs_reference
def test_patched_reference_compiles_and_links_against_the_c_abi(tmp_path):
    path = refhip.build(force=True)
    assert path and os.path.exists(path)
    und = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
    und = [l.split()[-1] for l in und.splitlines()]
    from_lib = sorted(s for s in und if s.startswith("lk_"))
    assert {"lk_create", "lk_update_points", "lk_update_imu", "lk_update_kin_imu", "lk_process_scan", "lk_map_build", "lk_set_state", "lk_get_state",
            "lk_init_process_cov_q"} <= set(from_lib), from_lib
    # nothing of the dropped translation units is wanted: no ESKF:: / VoxelMapManager:: / VoxelOctoTree symbol is undefined
    assert not [s for s in und if "ESKF" in s or "VoxelMapManager" in s or "VoxelOctoTree" in s or "calcBodyCov" in s], und
    from legkilo_amd import config

    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="lk_create"):   # LK_ERR_NO_DEVICE surfaces through KILO's constructor: no CPU path behind it
            refhip.PatchedReferenceKilo(config.LEG_FUSION, True, tmp_path / "k.yaml")


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_reference_kilo_on_the_c_abi_reproduces_the_reference_goldens(hip_lib, tmp_path, fused):
    import test_golden as tg
    from legkilo_amd import config

    for mode in ("imu", "kin", "c4"):
        g = np.load(tg.GR4 if mode == "c4" else tg.GR)
        params = {"imu": config.LEG_FUSION, "kin": dict(config.DITER, voxel_grid_resolution=0.3), "c4": config.DITER}[mode]
        k = refhip.PatchedReferenceKilo(params, mode == "imu", tmp_path / f"{mode}.yaml", fused=fused)
        worst = 0.0
        for s, pose, x in tg.replay_ref_golden(k, g, mode):
            assert int(pose.n_effect) == int(g[f"{mode}_n_effect"][s]), (mode, s, pose.n_effect, g[f"{mode}_n_effect"][s])
            worst = max(worst, np.abs(x - g[f"{mode}_x"][s]).max())
            assert np.allclose(x, g[f"{mode}_x"][s], rtol=0, atol=1e-6), (mode, s, worst)
        _, P = k.get_state()
        assert np.abs(P - g[f"{mode}_P"]).max() <= 1e-5 * np.abs(g[f"{mode}_P"]).max(), mode
        assert k.get_times() == tuple(g[f"{mode}_times"]), mode
        if mode == "imu":
            scenes.compare_maps(g["imu_map_blob"], k.map_export(), rtol=1e-5, ptol=1e-6)
        print(f"patched KILO on liblegkilo_hip.so, mode {mode}, fused={fused}: {len(g[f'{mode}_len'])} scans, n_effect {list(g[f'{mode}_n_effect'])}, max |dx| {worst:.2e}")
        k.close()


@pytest.mark.gpu
@pytest.mark.parametrize("imu_only", [True, False])
def test_reference_kilo_first_frame_on_the_c_abi(hip_lib, oracle_lib, tmp_path, imu_only):
    """KILO.cc:332-353 of the patched build: state_initial.hpp's own code writes gravity, gyro bias, rotation, P0 and Q through the ESKF
    proxies (`eskf.state().grav_ = ...`, `eskf.cov() = ...`), BuildVoxelMap runs on the device - against the oracle's first frame."""
    from legkilo_amd import synth

    sc = scenes.Scene()
    o = oracle_lib.Oracle(sc.cfg(), imu_mode_only=imu_only)
    k = refhip.PatchedReferenceKilo(sc.P, imu_only, tmp_path / "ff.yaml")
    t0 = 2.0
    raw = synth.vlp16_scan(sc.world, scenes.Frozen(sc.traj, t0), t0, sc.P)
    imus = synth.imu_stream(sc.traj, t0 - 0.1, t0, seed=77)
    kins = synth.kin_stream(sc.traj, t0 - 0.1, t0, sc.P, seed=77)
    for obj in (o, k):
        obj.first_frame(raw, t0, imus=imus) if imu_only else obj.first_frame(raw, t0, kins=kins)
    (xo, Po), (xk, Pk) = o.get_state(), k.get_state()
    assert np.allclose(xo, xk, rtol=1e-14, atol=1e-15), np.abs(xo - xk).max()
    assert np.array_equal(Po, Pk) and np.array_equal(o.get_Q(), k.get_Q())
    assert np.isclose(o.get_acc_norm(), k.get_acc_norm(), rtol=1e-15) and o.get_times() == k.get_times() == (t0, t0)
    scenes.compare_maps(o.map_export(), k.map_export(), rtol=1e-5, ptol=1e-7)
    o.close()
    k.close()
