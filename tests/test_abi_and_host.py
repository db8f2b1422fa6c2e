"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/legkilo_hip.h declares, agrees with the ctypes mirrors on struct sizes, and fails LOUDLY
(no CPU fallback) when no gfx950 device is present.  No compute call is made without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from legkilo_amd import abi, binding, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    binding.build()
    return C.CDLL(binding.LIB_PATH)


def test_every_declared_symbol_is_exported(hip):
    syms = ge.declared_symbols()
    assert len(syms) >= 40 and "lk_update_points" in syms and "lk_batch_replay_dev" in syms
    missing = [s for s in syms if not hasattr(hip, s)]
    assert not missing, missing
    assert sorted(binding.EXPORTS) == syms
    assert hip.lk_abi_version() == 1


def test_struct_sizes_match_header(oracle_lib):
    out = (C.c_size_t * 8)()
    oracle_lib.lib().lko_abi_sizes(out)  # sizeof() as the C++ compiler sees include/legkilo_hip.h
    got = [C.sizeof(t) for t in (abi.lk_config, abi.lk_point, abi.lk_imu, abi.lk_kin_imu, abi.lk_pose)]
    assert list(out)[:5] == got
    root, node, plane, block = abi.blob_dtypes()
    assert (plane.itemsize, node.itemsize, block.itemsize) == tuple(out)[5:8] == (256, 128, 72 * abi.LK_BLOCK_PTS)


def test_no_gpu_means_loud_failure_not_fallback(hip):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is covered by test_no_device_fallback_is_loud")
    cfg = config.make_config()
    h = C.c_void_p()
    hip.lk_last_error.restype = C.c_char_p
    rc = hip.lk_create(C.byref(cfg), C.byref(h))
    assert rc in (-4, -2) and not h.value
    assert b"fallback" in hip.lk_last_error(None) or b"HIP" in hip.lk_last_error(None) or b"hip" in hip.lk_last_error(None)
    with pytest.raises(binding.LegKiloError):
        binding.LegKiloHip(cfg)


def test_invalid_configs_are_rejected_before_touching_a_device(hip):
    hip.lk_last_error.restype = C.c_char_p
    for field, val in (("max_layer", 7), ("max_points_num", 64), ("n_slots", 0)):
        cfg = config.make_config()
        setattr(cfg, field, val)
        h = C.c_void_p()
        assert hip.lk_create(C.byref(cfg), C.byref(h)) == -1, field


def test_product_never_references_the_oracle():
    """The product path may not import, link or call anything under oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "leg-kilo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".cc", "Makefile")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"oracle_binding|liblegkilo_oracle|lko_|#include\s+\"[^\"]*oracle", src), os.path.join(dp, f)
    out = subprocess.run(["ldd", binding.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_host_mirror_compiles_against_the_c_abi():
    """leg-kilo_amd/host/*.hpp (the C++ mirror of ESKF / VoxelMapManager / KILO path) is header-only over the C-ABI."""
    src = os.path.join(ROOT, "leg-kilo_amd", "host", "example_kilo_path.cc")
    exe = "/tmp/lk_host_example"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "leg-kilo_amd", "host"),
                        src, "-o", exe, "-L", os.path.join(ROOT, "leg-kilo_amd"), "-llegkilo_hip",
                        "-Wl,-rpath," + os.path.join(ROOT, "leg-kilo_amd")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_host_rccl_example_compiles_against_rccl_and_the_c_abi():
    """The C++ multi-GPU hook (host/legkilo_rccl.hpp: map broadcast over a caller-owned ncclComm_t, shard partition, all-gather of the
    result records) and its one-process-all-GPUs example compile and link against rccl.h, the HIP runtime and the C-ABI."""
    src = os.path.join(ROOT, "leg-kilo_amd", "host", "example_replay_rccl.cc")
    exe = "/tmp/lk_host_rccl_example"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "leg-kilo_amd", "host"), src, "-o", exe, "-L", os.path.join(ROOT, "leg-kilo_amd"), "-llegkilo_hip",
                        "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "leg-kilo_amd"),
                        "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_ragged_tables_layout():
    """binding.ragged_tables flattens per-scan bucket tables into exactly what lk_batch_replay_ragged(_imu)_dev takes."""
    import lk_pkg

    lk_pkg.load()
    import numpy as np
    from legkilo_amd import binding, synth

    offs = [np.array([0, 3, 7], dtype=np.uint32), np.array([0, 5], dtype=np.uint32)]
    dts = [np.array([0.0, 0.002]), np.array([0.004])]
    t = binding.LegKiloHip.ragged_tables([0, 7, 12], offs, dts, [1.0, 1.1])
    assert t["n_scans"] == 2 and t["scan_off"].dtype == np.uint64 and list(t["scan_off"]) == [0, 7, 12]
    assert t["n_buckets"].dtype == np.uint32 and list(t["n_buckets"]) == [2, 1]
    assert t["bucket_off"].dtype == np.uint32 and list(t["bucket_off"]) == [0, 3, 7, 0, 5]
    assert t["bucket_dt"].dtype == np.float64 and list(t["bucket_dt"]) == [0.0, 0.002, 0.004]
    assert list(t["t_begin"]) == [1.0, 1.1] and "n_imu" not in t
    imus = [np.zeros(3, dtype=synth.IMU_DTYPE), np.zeros(0, dtype=synth.IMU_DTYPE)]
    t = binding.LegKiloHip.ragged_tables([0, 7, 12], offs, dts, [1.0, 1.1], imus=imus)
    assert list(t["n_imu"]) == [3, 0] and t["imus"].nbytes == 3 * 56
    with pytest.raises(AssertionError):
        binding.LegKiloHip.ragged_tables([0, 7], offs, dts, [1.0, 1.1])
