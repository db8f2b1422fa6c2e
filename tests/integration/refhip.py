"""TEST INFRASTRUCTURE.  ctypes driver of oracle/_ref/liblegkilo_refhip.so = the reference's own KILO.cc / KILO.h after kilo_hip.ed, built
without eskf.cc / voxel_map.cc on top of liblegkilo_hip.so (tests/integration/Makefile).  Same call surface as
oracle_binding.ReferenceKilo (the UNPATCHED build of the same class), so that one replay function drives either."""
import ctypes as C
import os
import subprocess

import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "liblegkilo_refhip.so")
REF_SRC = "/root/reference/legkilo/src"


def build(force=False):
    """Applies the patch to a scratch copy of the reference's files and builds the library when the reference tree is present
    (this container); where it is not (GPU box) a prebuilt library is used as is.  Returns the path, or None when neither exists."""
    if os.path.exists(os.path.join(REF_SRC, "core", "slam", "KILO.cc")):
        subprocess.check_call(["make", "-C", HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB if os.path.exists(LIB) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build()
        assert path is not None, "oracle/_ref/liblegkilo_refhip.so is not built and /root/reference is absent"
        l = C.CDLL(path)
        l.lkk_create.restype = C.c_void_p
        l.lkk_get_acc_norm.restype = C.c_double
        l.lkk_last_error.restype = C.c_char_p
        _lib = l
    return _lib


class PatchedReferenceKilo(ob.Oracle):
    """KILO::process of the PATCHED reference: YAML -> KiloPath (lk_create), first frame through state_initial.hpp's own code and the
    ESKF proxies, voxel grid + time sort on the host as in the reference, the bucket loop either as the reference's loop (one C-ABI call per
    bucket / message) or, fused=True, as one lk_process_scan call."""

    def __init__(self, params, imu_mode_only, yaml_path, fused=False):
        l = lib()
        self.L = ob._RenamedK(l)
        ob.write_reference_yaml(yaml_path, params, imu_mode_only)
        h = l.lkk_create(str(yaml_path).encode())
        if not h:
            raise RuntimeError("KILO(config_file) on liblegkilo_hip.so failed: " + l.lkk_last_error().decode())
        self.h = C.c_void_p(h)
        l.lkk_set_fused_scan(self.h, int(fused))
