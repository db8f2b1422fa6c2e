"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes wrapper over oracle/liblegkilo_oracle.so.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product (leg-kilo_amd/) never imports this module.  Parity: pinned against oracle/_ref, the reference's own
sources compiled here (see smallmat.hpp); `Reference` / `ReferenceKilo` below drive that build.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import lk_pkg

lk_pkg.load()
from legkilo_amd import abi  # noqa: E402  (ABI struct declarations only)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liblegkilo_oracle.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cc", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "legkilo_hip.h"))
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liblegkilo_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
        _lib.lko_create.restype = C.c_void_p
        _lib.lko_get_acc_norm.restype = C.c_double
        _lib.lko_hash_vec3.restype = C.c_size_t
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """Same call surface as legkilo_amd.binding.LegKiloHip (lko_ instead of lk_)."""

    def __init__(self, cfg, imu_mode_only=True):
        self.cfg = cfg
        self.L = lib()
        self.h = C.c_void_p(self.L.lko_create(C.byref(cfg), int(imu_mode_only)))

    def close(self):
        if self.h:
            self.L.lko_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ESKF ----
    def set_state(self, x36=None, P=None):
        x36 = None if x36 is None else _f64(x36).reshape(36)
        P = None if P is None else _f64(P).reshape(900)
        self.L.lko_set_state(self.h, _p(x36), _p(P))

    def get_state(self):
        x = np.zeros(36)
        P = np.zeros(900)
        self.L.lko_get_state(self.h, _p(x), _p(P))
        return x, P.reshape(30, 30)

    def set_Q(self, Q):
        Q = _f64(Q).reshape(900)
        self.L.lko_set_Q(self.h, _p(Q))

    def get_Q(self):
        Q = np.zeros(900)
        self.L.lko_get_Q(self.h, _p(Q))
        return Q.reshape(30, 30)

    def init_process_cov_q(self):
        self.L.lko_init_process_cov_q(self.h)

    def set_times(self, last_predict_t, last_update_t):
        self.L.lko_set_times(self.h, C.c_double(last_predict_t), C.c_double(last_update_t))

    def get_times(self):
        a, b = C.c_double(), C.c_double()
        self.L.lko_get_times(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def set_acc_norm(self, a):
        self.L.lko_set_acc_norm(self.h, C.c_double(a))

    def get_acc_norm(self):
        return self.L.lko_get_acc_norm(self.h)

    def set_literal_max_n(self, n):
        self.L.lko_set_literal_max_n(self.h, int(n))

    def set_map_insert(self, on):
        self.L.lko_set_map_insert(self.h, int(on))

    def get_fx(self, dt):
        F = np.zeros(900)
        self.L.lko_get_fx(self.h, C.c_double(dt), _p(F))
        return F.reshape(30, 30)

    def get_function_f(self, dt):
        f = np.zeros(30)
        self.L.lko_get_function_f(self.h, C.c_double(dt), _p(f))
        return f

    def predict(self, dt, prop_state, prop_cov):
        self.L.lko_predict(self.h, C.c_double(dt), int(prop_state), int(prop_cov))

    def update_by_points(self, h6, z, R):
        h6, z, R = _f64(h6), _f64(z), _f64(R)
        self.L.lko_update_by_points(self.h, _p(h6), _p(z), _p(R), C.c_size_t(len(z)))

    def update_by_imu(self, z6, R6):
        z6, R6 = _f64(z6), _f64(R6)
        self.L.lko_update_by_imu(self.h, _p(z6), _p(R6))

    def update_by_kin_imu(self, ki_h, ki_z, ki_R):
        ki_h, ki_z, ki_R = _f64(ki_h), _f64(ki_z), _f64(ki_R)
        self.L.lko_update_by_kin_imu(self.h, _p(ki_h), _p(ki_z), _p(ki_R), C.c_size_t(len(ki_z)))

    # ---- map ----
    def map_build(self, xyz_world, xyz_body):
        w = np.ascontiguousarray(xyz_world, dtype=np.float32)
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        self.L.lko_map_build(self.h, _p(w), _p(b), C.c_size_t(len(w)))

    def map_update(self, pw, var9):
        pw, var9 = _f64(pw), _f64(var9)
        self.L.lko_map_update(self.h, _p(pw), _p(var9), C.c_size_t(len(pw)))

    def residuals(self, xyz_body):
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        n = len(b)
        h6, z, R, valid = np.zeros((n, 6)), np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.uint8)
        self.L.lko_residuals(self.h, _p(b), C.c_size_t(n), _p(h6), _p(z), _p(R), _p(valid))
        return h6, z, R, valid

    def map_slide(self, position, sliding_thresh=8.0, half_map_size=100):
        """VoxelMapManager::mapSliding (voxel_map.cc:552-569) -> (slid, n_removed)."""
        pos = _f64(position)
        slid, nrem = C.c_int(0), C.c_uint32(0)
        self.L.lko_map_slide(self.h, _p(pos), C.c_double(sliding_thresh), C.c_int(half_map_size), C.byref(slid), C.byref(nrem))
        return bool(slid.value), int(nrem.value)

    def map_clear_outside(self, x_max, x_min, y_max, y_min, z_max, z_min):
        nrem = C.c_uint32(0)
        self.L.lko_map_clear_outside(self.h, *(C.c_int(int(v)) for v in (x_max, x_min, y_max, y_min, z_max, z_min)), C.byref(nrem))
        return int(nrem.value)

    def get_last_slide_position(self):
        out = np.zeros(3)
        self.L.lko_map_slide_position(self.h, 0, _p(out))
        return out

    def set_last_slide_position(self, p):
        p = _f64(p).copy()
        self.L.lko_map_slide_position(self.h, 1, _p(p))

    def residual_margins(self, xyz_body3):
        """(valid, [range-gate margin, sigma-gate margin, key margin]) of one body point: relative distance of the closest gate
        on its match path from its threshold (test diagnostic, lko_residual_margins)."""
        b = np.ascontiguousarray(xyz_body3, dtype=np.float32).reshape(3)
        out = np.zeros(3)
        v = self.L.lko_residual_margins(self.h, _p(b), _p(out))
        return int(v), out

    def match_voxel(self, key, pw, var):
        """build_single_residual on root voxel `key` (is_success = False, prob = 0 on entry) ->
        dict(found, success, prob, normal, center, d, dis_to_plane, layer)."""
        key = np.ascontiguousarray(key, dtype=np.int32)
        pw, var = _f64(pw).reshape(3), _f64(var).reshape(9)
        found, ok, layer = C.c_int(0), C.c_int(0), C.c_int(-1)
        prob, d, dis = C.c_double(0), C.c_double(0), C.c_float(0)
        n, c = np.zeros(3), np.zeros(3)
        self.L.lko_match_voxel(self.h, _p(key), _p(pw), _p(var), C.byref(found), C.byref(ok), C.byref(prob), _p(n), _p(c),
                               C.byref(d), C.byref(dis), C.byref(layer))
        return dict(found=bool(found.value), success=bool(ok.value), prob=prob.value, normal=n, center=c, d=d.value,
                    dis_to_plane=dis.value, layer=layer.value)

    def map_export(self):
        nbytes = C.c_size_t(0)
        self.L.lko_map_export(self.h, None, C.byref(nbytes))
        buf = np.zeros(nbytes.value, dtype=np.uint8)
        rc = self.L.lko_map_export(self.h, _p(buf), C.byref(nbytes))
        assert rc == 0
        return buf

    def map_import(self, blob):
        """Rebuild the octrees from a blob of include/legkilo_hip.h (e.g. LegKiloHip.map_export()): the checker then
        replays against exactly the map the device handle holds."""
        buf = np.ascontiguousarray(np.frombuffer(blob, dtype=np.uint8))
        rc = self.L.lko_map_import(self.h, _p(buf), C.c_size_t(buf.nbytes))
        assert rc == 0, "lko_map_import: malformed blob"

    def map_stats(self):
        n = C.c_uint32()
        self.L.lko_map_stats(self.h, C.byref(n))
        return n.value

    # ---- KILO path ----
    def update_points(self, t, xyz_body):
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        n = len(b)
        w = np.zeros((n, 3), dtype=np.float32)
        inten = np.zeros(n, dtype=np.float32)
        ne = C.c_size_t(0)
        self.L.lko_update_points(self.h, C.c_double(t), _p(b), C.c_size_t(n), _p(w), _p(inten), C.byref(ne))
        return w, inten, ne.value

    def update_imu(self, imu_rec):
        a = np.ascontiguousarray(imu_rec)
        self.L.lko_update_imu(self.h, _p(a))

    def update_kin_imu(self, kin_rec):
        a = np.ascontiguousarray(kin_rec)
        self.L.lko_update_kin_imu(self.h, _p(a))

    def first_frame(self, raw_pts, end_time, imus=None, kins=None):
        raw = np.ascontiguousarray(raw_pts)
        ni = 0 if imus is None else len(imus)
        nk = 0 if kins is None else len(kins)
        imus = None if imus is None else np.ascontiguousarray(imus)
        kins = None if kins is None else np.ascontiguousarray(kins)
        self.L.lko_first_frame(self.h, _p(raw), C.c_size_t(len(raw)), C.c_double(end_time), _p(imus), C.c_size_t(ni),
                               _p(kins), C.c_size_t(nk))

    def process_scan(self, sorted_pts, t_begin, imus=None, kins=None, want_world=False, with_sort=False):
        pts = np.ascontiguousarray(sorted_pts)
        ni = 0 if imus is None else len(imus)
        nk = 0 if kins is None else len(kins)
        imus = None if imus is None else np.ascontiguousarray(imus)
        kins = None if kins is None else np.ascontiguousarray(kins)
        w = np.zeros((len(pts), 3), dtype=np.float32) if want_world else None
        pose = abi.lk_pose()
        self.L.lko_process_scan(self.h, _p(pts), C.c_size_t(len(pts)), C.c_double(t_begin), _p(imus), C.c_size_t(ni),
                                _p(kins), C.c_size_t(nk), _p(w), C.byref(pose), int(with_sort))
        return pose, w


# ---- the reference's own eskf.cc / voxel_map.cc (oracle/_ref, built by `make ref` against oracle/shim) ----
_REF_LIB = os.path.join(_HERE, "_ref", "liblegkilo_ref.so")
_REF_SRC = "/root/reference/legkilo/src/core/slam/eskf.cc"
_ref = None


def build_ref(force=False):
    """Compile oracle/_ref from the reference tree when it is present (this container); a prebuilt library is used
    as is where the tree does not exist (GPU box).  Returns the path, or None when neither exists."""
    if os.path.exists(_REF_SRC):
        deps = [os.path.join(_HERE, "ref_capi.cc"), os.path.join(_HERE, "ref_kilo_capi.cc"), os.path.join(_HERE, "ref_decode_capi.cc"), os.path.join(_HERE, "ref_tum_capi.cc"),
                os.path.join(_HERE, "export_blob.hpp"), os.path.join(_HERE, "Makefile"), _REF_SRC, _REF_SRC.replace("eskf.cc", "KILO.cc"),
                _REF_SRC.replace("eskf.cc", "voxel_map.cc"), os.path.join(_HERE, "shim", "Eigen", "Dense")]
        if force or not os.path.exists(_REF_LIB) or any(os.path.getmtime(d) > os.path.getmtime(_REF_LIB) for d in deps):
            subprocess.check_call(["make", "-C", _HERE, "-B", "ref"], stdout=subprocess.DEVNULL)
    return _REF_LIB if os.path.exists(_REF_LIB) else None


class _Renamed:
    """lko_* attribute access answered by the lkr_* symbol of the reference library."""

    def __init__(self, cdll):
        self._l = cdll

    def __getattr__(self, name):
        return getattr(self._l, name.replace("lko_", "lkr_", 1))


def ref_lib():
    global _ref
    if _ref is None:
        path = build_ref()
        if path is None:
            return None
        l = C.CDLL(path)
        l.lkr_create.restype = C.c_void_p
        l.lkr_hash_vec3.restype = C.c_size_t
        _ref = _Renamed(l)
    return _ref


class Reference(Oracle):
    """The reference's OWN ESKF + VoxelMapManager (no KILO glue): the subset of the Oracle surface that eskf.cc and
    voxel_map.cc implement — state / covariance access, predict, the three updates, BuildVoxelMap, UpdateVoxelMap,
    build_single_residual on a voxel, map export / sliding."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.L = ref_lib()
        assert self.L is not None, "oracle/_ref is not built and /root/reference is absent"
        self.h = C.c_void_p(self.L.lko_create(C.byref(cfg), 1))


def write_reference_yaml(path, params, imu_mode_only):
    """A flat YAML with the keys KILO::initializeFromYaml reads (KILO.cc:25-83), from a parameter dict keyed like the
    reference's config files (legkilo_amd.config.LEG_FUSION / DITER)."""
    p = dict(params)
    p.update(only_imu_use=bool(imu_mode_only))
    p.setdefault("pub_plane_en", False)
    p.setdefault("map_sliding_en", False)
    p.setdefault("half_map_size", 100)
    p.setdefault("sliding_thresh", 8)
    with open(path, "w") as f:
        for k, v in p.items():
            if isinstance(v, bool):
                v = "true" if v else "false"
            elif isinstance(v, (list, tuple)):
                v = "[" + ", ".join(repr(float(x)) if isinstance(x, float) else str(x) for x in v) + "]"
            elif isinstance(v, float):
                v = repr(v)
            f.write(f"{k}: {v}\n")
    return path


class _RenamedK(_Renamed):
    def __getattr__(self, name):
        return getattr(self._l, name.replace("lko_", "lkk_", 1))


class ReferenceKilo(Oracle):
    """The reference's OWN legkilo::KILO: KILO::process (first frame, bucket loop, predictUpdatePoint / Imu / KinImu)
    behind the whole-scan part of the Oracle surface (set/get_state, set_times, set_acc_norm, map_build, first_frame,
    process_scan, map_export)."""

    def __init__(self, params, imu_mode_only, yaml_path):
        rl = ref_lib()
        assert rl is not None, "oracle/_ref is not built and /root/reference is absent"
        self.L = _RenamedK(rl._l)
        self.L._l.lkk_create.restype = C.c_void_p
        self.L._l.lkk_get_acc_norm.restype = C.c_double
        write_reference_yaml(yaml_path, params, imu_mode_only)
        self.h = C.c_void_p(self.L._l.lkk_create(str(yaml_path).encode()))
        assert self.h, "KILO(config_file) failed"


def ref_decode(raw, layout, time_scale, filter_num, blind, header_stamp=0.0):
    """The reference's own LidarProcessing::processing (lidar_processing.cc:25-108, oracle/_ref) on a PointCloud2 payload
    given as a packed numpy record array + the lk_cloud_layout dict the C-ABI takes -> (points, begin, end)."""
    rl = ref_lib()
    assert rl is not None
    raw = np.ascontiguousarray(raw)
    lay = abi.lk_cloud_layout(**layout)
    out = np.zeros(len(raw), dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("curvature", "<f4")]))
    n, tb, te = C.c_size_t(0), C.c_double(0), C.c_double(0)
    rc = rl._l.lkd_decode(_p(raw), C.c_size_t(len(raw)), C.byref(lay), C.c_double(time_scale), C.c_int(filter_num), C.c_float(blind),
                          C.c_double(header_stamp), _p(out), C.byref(n), C.byref(tb), C.byref(te))
    assert rc == 0, rc
    return out[: n.value], tb.value, te.value


def ref_write_tum(stamps, rots, poss):
    """The reference's own TrajectorySaver::write (trajectory_saver.hpp:43-50, oracle/_ref) -> text of the file it wrote."""
    rl = ref_lib()
    assert rl is not None
    t, R, p = _f64(stamps), _f64(rots).reshape(-1, 9), _f64(poss).reshape(-1, 3)
    buf = C.create_string_buffer(1024)
    rc = rl._l.lkt_write_tum(_p(t), _p(R), _p(p), C.c_size_t(len(t)), buf, C.c_size_t(1024))
    assert rc == 0, rc
    path = buf.value.decode()
    text = open(path).read()
    os.remove(path)
    return text


def _hooks(which):
    return ref_lib() if which == "ref" else lib()


def state_minus(xa, xb, which="oracle"):
    xa, xb, d = _f64(xa), _f64(xb), np.zeros(30)
    _hooks(which).lko_state_minus(_p(xa), _p(xb), _p(d))
    return d


def key_floor(p, voxel_size, which="oracle"):
    p, k = _f64(p), np.zeros(3, dtype=np.int32)
    _hooks(which).lko_key_floor(_p(p), C.c_double(voxel_size), _p(k))
    return tuple(int(v) for v in k)


# ---- unit-level hooks ----
def calc_body_cov(pb, range_inc, degree_inc, which="oracle"):
    pb = _f64(pb)
    cov = np.zeros(9)
    _hooks(which).lko_calc_body_cov(_p(pb), C.c_float(range_inc), C.c_float(degree_inc), _p(cov))
    return cov.reshape(3, 3)


def eig_sym3(A):
    A = _f64(A).reshape(9)
    ev, V = np.zeros(3), np.zeros(9)
    lib().lko_eig_sym3(_p(A), _p(ev), _p(V))
    return ev, V.reshape(3, 3)


def init_plane(pw, var9, planer_threshold=0.01, which="oracle"):
    pw, var9 = _f64(pw), _f64(var9)
    rec = np.zeros(1, dtype=abi.blob_dtypes()[2])
    pv = np.zeros(36)
    _hooks(which).lko_init_plane(_p(pw), _p(var9), C.c_size_t(len(pw)), C.c_float(planer_threshold), _p(rec), _p(pv))
    return rec[0], pv.reshape(6, 6)


def exp_log(v, which="oracle"):
    v = _f64(v)
    a, b, l = np.zeros(9), np.zeros(9), np.zeros(3)
    _hooks(which).lko_exp_log(_p(v), _p(a), _p(b), _p(l))
    return a.reshape(3, 3), b.reshape(3, 3), l


def hash_vec3(x, y, z, which="oracle"):
    return _hooks(which).lko_hash_vec3(C.c_int(x), C.c_int(y), C.c_int(z))
