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
