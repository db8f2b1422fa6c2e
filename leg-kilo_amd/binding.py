"""ctypes wrapper over liblegkilo_hip.so (the C-ABI of include/legkilo_hip.h).

There is no fallback: if the shared library is missing or no gfx950 device is visible the
constructor raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEGKILO_HIP_LIB", os.path.join(_HERE, "liblegkilo_hip.so"))  # override: A/B builds

EXPORTS = [
    "lk_abi_version", "lk_create", "lk_destroy", "lk_last_error", "lk_set_state", "lk_get_state", "lk_set_Q", "lk_get_Q",
    "lk_init_process_cov_q", "lk_set_times", "lk_get_times", "lk_set_acc_norm", "lk_get_acc_norm", "lk_get_fx", "lk_get_function_f",
    "lk_predict", "lk_update_by_points", "lk_update_by_imu", "lk_update_by_kin_imu", "lk_map_build", "lk_map_update",
    "lk_residuals", "lk_match_points", "lk_map_slide", "lk_map_clear_outside", "lk_map_slide_position", "lk_map_stats", "lk_map_export", "lk_map_import", "lk_map_export_dev", "lk_map_import_dev",
    "lk_update_points", "lk_update_imu", "lk_update_kin_imu", "lk_process_scan", "lk_process_scan_dev",
    "lk_decode_scan", "lk_decode_scan_dev", "lk_preprocess_scan", "lk_preprocess_scan_dev", "lk_process_raw_scan", "lk_batch_set_priors", "lk_batch_set_priors_dev", "lk_batch_get_states", "lk_batch_get_states_dev", "lk_batch_residuals_dev", "lk_batch_order", "lk_batch_changed", "lk_batch_prepare_dev", "lk_batch_order_stats", "lk_batch_replay_dev", "lk_batch_sort_by_voxel_dev", "lk_batch_replay_async_dev", "lk_batch_replay_ragged_dev", "lk_batch_replay_ragged_imu_dev", "lk_batch_replay_ragged_kin_dev", "lk_batch_replay_scans_dev", "lk_batch_replay_overlay_dev", "lk_batch_replay_overlay_ragged_dev", "lk_overlay_reserve", "lk_overlay_export", "lk_overlay_stats", "lk_overlay_pool_bytes", "lk_overlay_resident_rounds", "lk_profile_enable", "lk_profile_get", "lk_profile_reset",
    "lk_device_malloc", "lk_device_free", "lk_memcpy_h2d", "lk_memcpy_d2h", "lk_synchronize", "lk_stream", "lk_stream_pipeline", "lk_stream_resident", "lk_stream_grid", "lk_stream_grid_placement", "lk_stream_stats", "lk_stream_resident_stats", "lk_test_stall",
]


class LegKiloError(RuntimeError):
    pass


def build(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(_HERE, "..", "include", "legkilo_hip.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(d) > os.path.getmtime(LIB_PATH) for d in deps):
        subprocess.check_call(["make", "-C", csrc, "-B"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LegKiloError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # PyTorch ships its own copy of the HIP runtime.  When both copies live in one process, torch's has to come up FIRST: brought up
        # after this library has created its streams, torch.cuda fails with "No HIP GPUs are available".  So a process that has torch
        # imported (the multi-GPU replay, the device-pointer entries) gets torch's runtime initialised before the first handle exists.
        import sys
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            torch.cuda.init()
        L = C.CDLL(LIB_PATH)
        L.lk_last_error.restype = C.c_char_p
        L.lk_last_error.argtypes = [C.c_void_p]
        L.lk_stream.restype = C.c_void_p
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class LegKiloHip:
    def __init__(self, cfg):
        self.L = lib()
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.L.lk_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            msg = self.L.lk_last_error(None)
            self.h = None
            raise LegKiloError(f"lk_create failed ({rc}): {msg.decode() if msg else ''}")

    def _chk(self, rc):
        if rc != 0:
            msg = self.L.lk_last_error(self.h)
            raise LegKiloError(f"liblegkilo_hip error {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.L.lk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- ESKF surface ----
    def set_state(self, x36=None, P=None, slot=0):
        x36 = None if x36 is None else _f64(x36).reshape(36)
        P = None if P is None else _f64(P).reshape(900)
        self._chk(self.L.lk_set_state(self.h, C.c_uint32(slot), _p(x36), _p(P)))

    def get_state(self, slot=0):
        x, P = np.zeros(36), np.zeros(900)
        self._chk(self.L.lk_get_state(self.h, C.c_uint32(slot), _p(x), _p(P)))
        return x, P.reshape(30, 30)

    def set_Q(self, Q):
        Q = _f64(Q).reshape(900)
        self._chk(self.L.lk_set_Q(self.h, _p(Q)))

    def get_Q(self):
        Q = np.zeros(900)
        self._chk(self.L.lk_get_Q(self.h, _p(Q)))
        return Q.reshape(30, 30)

    def init_process_cov_q(self):
        self._chk(self.L.lk_init_process_cov_q(self.h))

    def set_times(self, last_predict_t, last_update_t, slot=0):
        self._chk(self.L.lk_set_times(self.h, C.c_uint32(slot), C.c_double(last_predict_t), C.c_double(last_update_t)))

    def get_times(self, slot=0):
        a, b = C.c_double(), C.c_double()
        self._chk(self.L.lk_get_times(self.h, C.c_uint32(slot), C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_acc_norm(self, a):
        self._chk(self.L.lk_set_acc_norm(self.h, C.c_double(a)))

    def get_acc_norm(self):
        a = C.c_double()
        self._chk(self.L.lk_get_acc_norm(self.h, C.byref(a)))
        return a.value

    def get_fx(self, dt, slot=0):
        F = np.zeros(900)
        self._chk(self.L.lk_get_fx(self.h, C.c_uint32(slot), C.c_double(dt), _p(F)))
        return F.reshape(30, 30)

    def get_function_f(self, dt, slot=0):
        f = np.zeros(30)
        self._chk(self.L.lk_get_function_f(self.h, C.c_uint32(slot), C.c_double(dt), _p(f)))
        return f

    def predict(self, dt, prop_state, prop_cov, slot=0):
        self._chk(self.L.lk_predict(self.h, C.c_uint32(slot), C.c_double(dt), int(prop_state), int(prop_cov)))

    def update_by_points(self, h6, z, R, slot=0):
        h6, z, R = _f64(h6), _f64(z), _f64(R)
        self._chk(self.L.lk_update_by_points(self.h, C.c_uint32(slot), _p(h6), _p(z), _p(R), C.c_size_t(len(z))))

    def update_by_imu(self, z6, R6, slot=0):
        z6, R6 = _f64(z6), _f64(R6)
        self._chk(self.L.lk_update_by_imu(self.h, C.c_uint32(slot), _p(z6), _p(R6)))

    def update_by_kin_imu(self, ki_h, ki_z, ki_R, slot=0):
        ki_h, ki_z, ki_R = _f64(ki_h), _f64(ki_z), _f64(ki_R)
        self._chk(self.L.lk_update_by_kin_imu(self.h, C.c_uint32(slot), _p(ki_h), _p(ki_z), _p(ki_R), C.c_size_t(len(ki_z))))

    # ---- VoxelMapManager surface ----
    def map_build(self, xyz_world, xyz_body):
        w = np.ascontiguousarray(xyz_world, dtype=np.float32)
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        self._chk(self.L.lk_map_build(self.h, _p(w), _p(b), C.c_size_t(len(w))))

    def map_update(self, pw, var9):
        pw, var9 = _f64(pw), _f64(var9)
        self._chk(self.L.lk_map_update(self.h, _p(pw), _p(var9), C.c_size_t(len(pw))))

    def residuals(self, xyz_body):
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        n = len(b)
        h6, z, R, valid = np.zeros((n, 6)), np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.uint8)
        self._chk(self.L.lk_residuals(self.h, _p(b), C.c_size_t(n), _p(h6), _p(z), _p(R), _p(valid)))
        return h6, z, R, valid

    def match_points(self, keys, pw, var):
        """VoxelMapManager::build_single_residual (voxel_map.cc:363-427) for n points held as pointWithVar: keys n x 3 int32, pw n x 3,
        var n x 3 x 3 -> dict(found, success, prob, normal, center, d, dis_to_plane, layer), arrays of length n."""
        keys = np.ascontiguousarray(keys, dtype=np.int32).reshape(-1, 3)
        n = len(keys)
        pw, var = _f64(pw).reshape(n, 3), _f64(var).reshape(n, 9)
        found, ok = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        prob, d, dis, layer = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32)
        normal, center = np.zeros((n, 3)), np.zeros((n, 3))
        self._chk(self.L.lk_match_points(self.h, C.c_size_t(n), _p(keys), _p(pw), _p(var), _p(found), _p(ok), _p(prob), _p(normal), _p(center),
                                         _p(d), _p(dis), _p(layer)))
        return dict(found=found.astype(bool), success=ok.astype(bool), prob=prob, normal=normal, center=center, d=d, dis_to_plane=dis, layer=layer)

    def map_stats(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.lk_map_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- local map sliding (voxel_map.cc:552-594) ----
    def map_slide(self, position, sliding_thresh=8.0, half_map_size=100):
        """VoxelMapManager::mapSliding with position_last_ = position -> (slid, n_removed)."""
        pos = _f64(position)
        slid, nrem = C.c_int32(0), C.c_uint32(0)
        self._chk(self.L.lk_map_slide(self.h, _p(pos), C.c_double(sliding_thresh), C.c_int32(int(half_map_size)),
                                      C.byref(slid), C.byref(nrem)))
        return bool(slid.value), int(nrem.value)

    def map_clear_outside(self, x_max, x_min, y_max, y_min, z_max, z_min):
        nrem = C.c_uint32(0)
        self._chk(self.L.lk_map_clear_outside(self.h, *(C.c_int32(int(v)) for v in (x_max, x_min, y_max, y_min, z_max, z_min)),
                                              C.byref(nrem)))
        return int(nrem.value)

    def get_last_slide_position(self):
        out = np.zeros(3)
        self._chk(self.L.lk_map_slide_position(self.h, C.c_int32(0), _p(out)))
        return out

    def set_last_slide_position(self, p):
        p = _f64(p).copy()
        self._chk(self.L.lk_map_slide_position(self.h, C.c_int32(1), _p(p)))

    def map_export(self):
        nbytes = C.c_size_t(0)
        self._chk(self.L.lk_map_export(self.h, None, C.byref(nbytes)))
        buf = np.zeros(nbytes.value, dtype=np.uint8)
        self._chk(self.L.lk_map_export(self.h, _p(buf), C.byref(nbytes)))
        return buf

    def map_import(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._chk(self.L.lk_map_import(self.h, _p(blob), C.c_size_t(blob.size)))

    def map_export_dev_size(self):
        nbytes = C.c_size_t(0)
        self._chk(self.L.lk_map_export_dev(self.h, None, C.byref(nbytes)))
        return nbytes.value

    def map_export_dev(self, d_ptr, nbytes):
        nb = C.c_size_t(nbytes)
        self._chk(self.L.lk_map_export_dev(self.h, C.c_void_p(d_ptr), C.byref(nb)))
        return nb.value

    def map_import_dev(self, d_ptr, nbytes):
        self._chk(self.L.lk_map_import_dev(self.h, C.c_void_p(d_ptr), C.c_size_t(nbytes)))

    # ---- KILO path ----
    def update_points(self, t, xyz_body):
        b = np.ascontiguousarray(xyz_body, dtype=np.float32)
        n = len(b)
        w = np.zeros((n, 3), dtype=np.float32)
        inten = np.zeros(n, dtype=np.float32)
        ne = C.c_size_t(0)
        self._chk(self.L.lk_update_points(self.h, C.c_double(t), _p(b), C.c_size_t(n), _p(w), _p(inten), C.byref(ne)))
        return w, inten, ne.value

    def update_imu(self, imu_rec):
        a = np.ascontiguousarray(imu_rec)
        self._chk(self.L.lk_update_imu(self.h, _p(a)))

    def update_kin_imu(self, kin_rec):
        a = np.ascontiguousarray(kin_rec)
        self._chk(self.L.lk_update_kin_imu(self.h, _p(a)))

    def process_scan(self, sorted_pts, t_begin, imus=None, kins=None, want_world=False):
        pts = np.ascontiguousarray(sorted_pts)
        ni = 0 if imus is None else len(imus)
        nk = 0 if kins is None else len(kins)
        imus = None if imus is None else np.ascontiguousarray(imus)
        kins = None if kins is None else np.ascontiguousarray(kins)
        w = np.zeros((len(pts), 3), dtype=np.float32) if want_world else None
        pose = abi.lk_pose()
        self._chk(self.L.lk_process_scan(self.h, _p(pts), C.c_size_t(len(pts)), C.c_double(t_begin), _p(imus), C.c_size_t(ni),
                                         _p(kins), C.c_size_t(nk), _p(w), C.byref(pose)))
        return pose, w

    def process_scan_dev(self, d_pts, n, t_begin, bucket_off, bucket_dt):
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        dt = _f64(bucket_dt)
        pose = abi.lk_pose()
        self._chk(self.L.lk_process_scan_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n), C.c_double(t_begin), _p(off), _p(dt),
                                             C.c_size_t(len(dt)), C.byref(pose)))
        return pose

    # ---- sensor decode + preprocessing in front of the path ----
    def decode_scan(self, msg_bytes, n_points, layout, time_scale, filter_num, blind, header_stamp=0.0):
        """layout = dict(point_step, off_x, off_y, off_z, off_time, lidar_type)."""
        data = np.ascontiguousarray(np.frombuffer(msg_bytes, dtype=np.uint8))
        lay = abi.lk_cloud_layout(**layout)
        out = np.zeros(n_points, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("curvature", "<f4")]))
        n_out, tb, te = C.c_size_t(0), C.c_double(0), C.c_double(0)
        self._chk(self.L.lk_decode_scan(self.h, _p(data), C.c_size_t(n_points), C.byref(lay), C.c_double(time_scale), int(filter_num),
                                        C.c_float(blind), C.c_double(header_stamp), _p(out), C.byref(n_out), C.byref(tb), C.byref(te)))
        return out[: n_out.value], tb.value, te.value

    def preprocess_scan(self, raw_pts, leaf):
        raw = np.ascontiguousarray(raw_pts)
        out = np.zeros(len(raw), dtype=raw.dtype)
        n_out = C.c_size_t(0)
        self._chk(self.L.lk_preprocess_scan(self.h, _p(raw), C.c_size_t(len(raw)), C.c_float(leaf), _p(out), C.byref(n_out)))
        return out[: n_out.value]

    def preprocess_scan_dev(self, d_raw, n_raw, leaf, d_out):
        n_out = C.c_size_t(0)
        self._chk(self.L.lk_preprocess_scan_dev(self.h, C.c_void_p(d_raw), C.c_size_t(n_raw), C.c_float(leaf), C.c_void_p(d_out),
                                                C.byref(n_out)))
        return n_out.value

    def process_raw_scan(self, raw_pts, leaf, t_begin, imus=None, kins=None):
        raw = np.ascontiguousarray(raw_pts)
        ni = 0 if imus is None else len(imus)
        nk = 0 if kins is None else len(kins)
        imus = None if imus is None else np.ascontiguousarray(imus)
        kins = None if kins is None else np.ascontiguousarray(kins)
        pose = abi.lk_pose()
        nd = C.c_size_t(0)
        self._chk(self.L.lk_process_raw_scan(self.h, _p(raw), C.c_size_t(len(raw)), C.c_float(leaf), C.c_double(t_begin), _p(imus),
                                             C.c_size_t(ni), _p(kins), C.c_size_t(nk), C.byref(nd), C.byref(pose)))
        return pose, nd.value

    # ---- batch replay ----
    def batch_set_priors(self, x36, P900):
        x36 = _f64(x36).reshape(-1, 36)
        P900 = _f64(P900).reshape(-1, 900)
        assert len(x36) == len(P900)
        self._chk(self.L.lk_batch_set_priors(self.h, _p(x36), _p(P900), C.c_size_t(len(x36))))

    def batch_set_priors_dev(self, d_x36, d_P900, n_scans):
        """Priors already in HBM (device pointers to n_scans x 36 and n_scans x 900 doubles); asynchronous."""
        self._chk(self.L.lk_batch_set_priors_dev(self.h, C.c_void_p(d_x36), C.c_void_p(d_P900), C.c_size_t(n_scans)))

    def batch_get_states(self, first_slot, n, want_P=True):
        """State [n, 36] and covariance [n, 30, 30] of the filter slots first_slot .. first_slot + n (one gather kernel)."""
        x = np.zeros((n, 36))
        P = np.zeros((n, 900)) if want_P else None
        self._chk(self.L.lk_batch_get_states(self.h, C.c_uint32(first_slot), C.c_size_t(n), _p(x), _p(P)))
        return x, (P.reshape(n, 30, 30) if want_P else None)

    def batch_get_states_dev(self, first_slot, n, d_x36, d_P900):
        """The same into device buffers (pointers or 0), asynchronous on the handle's stream."""
        self._chk(self.L.lk_batch_get_states_dev(self.h, C.c_uint32(first_slot), C.c_size_t(n), C.c_void_p(d_x36 or None), C.c_void_p(d_P900 or None)))

    def batch_replay_dev(self, d_pts, n_scans, n_pts, t_begin, bucket_off, bucket_dt, want_poses=True):
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        dt = _f64(bucket_dt)
        poses = (abi.lk_pose * n_scans)() if want_poses else None
        self._chk(self.L.lk_batch_replay_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), C.c_size_t(n_pts),
                                             C.c_double(t_begin), _p(off), _p(dt), C.c_size_t(len(dt)), poses))
        return poses

    def batch_residuals_dev(self, d_pts, n_scans, n_pts, d_rows8, d_valid):
        """Config 2 for a device-resident batch (lk_batch_residuals_dev): residual rows of n_scans x n_pts points, scan s under the current state of
        slot s, written to the device buffers d_rows8 [n][8] float64 = (h0..h5, z, R) per point and d_valid (uint8).  Asynchronous on the handle's stream."""
        self._chk(self.L.lk_batch_residuals_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), C.c_size_t(n_pts), C.c_void_p(d_rows8), C.c_void_p(d_valid)))

    def batch_order(self, mode):
        """0: replay every device-resident batch as given; 1 (default): the frozen-map batch entries keep a voxel-ordered copy of a batch they see (lk_batch_order)."""
        self._chk(self.L.lk_batch_order(self.h, int(mode)))

    def batch_prepare_dev(self, d_pts, n_scans, n_pts, bucket_off):
        """Make the voxel-ordered copy of a batch NOW (priors armed first) instead of at its third replay (lk_batch_prepare_dev)."""
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        self._chk(self.L.lk_batch_prepare_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), C.c_size_t(n_pts), _p(off), C.c_size_t(len(off) - 1)))

    def batch_changed(self):
        self._chk(self.L.lk_batch_changed(self.h))

    def batch_order_stats(self):
        """(batches examined at first sight, of those sorted into a copy, replays that found new content in a known buffer)."""
        out = np.zeros(3, dtype=np.uint64)
        self._chk(self.L.lk_batch_order_stats(self.h, _p(out)))
        return int(out[0]), int(out[1]), int(out[2])

    def batch_sort_by_voxel_dev(self, d_in, d_out, n_scans, n_pts, bucket_off):
        """Every bucket of every scan of a device-resident batch into root-voxel order under the slots' prior poses (lk_batch_sort_by_voxel_dev):
        d_out = the same scans, a residual wave's points a handful of voxels apart.  Priors first (batch_set_priors(_dev))."""
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        self._chk(self.L.lk_batch_sort_by_voxel_dev(self.h, C.c_void_p(d_in), C.c_void_p(d_out), C.c_size_t(n_scans), C.c_size_t(n_pts), _p(off),
                                                    C.c_size_t(len(off) - 1)))

    def batch_replay_async_dev(self, d_pts, first_slot, n_scans, n_pts, t_begin, bucket_off, bucket_dt, d_x36=None, d_P900=None,
                               host_out_ptr=None):
        """Enqueue a batch on slots [first_slot, first_slot + n_scans) without synchronising (alternate the slot range
        between consecutive batches to double-buffer); poses land in the PINNED host buffer at host_out_ptr once the
        batch's stream gets there (synchronize() waits for everything)."""
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        dt = _f64(bucket_dt)
        self._chk(self.L.lk_batch_replay_async_dev(self.h, C.c_void_p(d_pts), C.c_uint32(first_slot), C.c_size_t(n_scans),
                                                   C.c_size_t(n_pts), C.c_double(t_begin), _p(off), _p(dt), C.c_size_t(len(dt)),
                                                   C.c_void_p(d_x36) if d_x36 else None, C.c_void_p(d_P900) if d_P900 else None,
                                                   C.c_void_p(host_out_ptr) if host_out_ptr else None))

    def batch_replay_overlay_dev(self, d_pts, n_scans, n_pts, t_begin, bucket_off, bucket_dt, want_poses=True):
        """Batch replay WITH the map insert: every scan on its own copy-on-write overlay of the handle's map (lk_batch_replay_overlay_dev)."""
        off = np.ascontiguousarray(bucket_off, dtype=np.uint32)
        dt = _f64(bucket_dt)
        poses = (abi.lk_pose * n_scans)() if want_poses else None
        self._chk(self.L.lk_batch_replay_overlay_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), C.c_size_t(n_pts),
                                                     C.c_double(t_begin), _p(off), _p(dt), C.c_size_t(len(dt)), poses))
        return poses

    def batch_replay_overlay(self, scans, t_begin, bucket_off, bucket_dt):
        """Convenience: equally shaped host scans (lk_point arrays, time-sorted, one bucket table) -> HBM -> overlay replay on slots
        [0, len(scans)); priors are whatever batch_set_priors put into those slots.  Returns the poses."""
        allpts = np.ascontiguousarray(np.concatenate(scans))
        n_pts = len(scans[0])
        assert all(len(sc) == n_pts for sc in scans), "the overlay batch entry takes equally shaped scans (ragged: lk_batch_replay_overlay_ragged_dev)"
        d = self.device_malloc(allpts.nbytes)
        try:
            self.h2d(d, allpts)
            return self.batch_replay_overlay_dev(d, len(scans), n_pts, t_begin, bucket_off, bucket_dt)
        finally:
            self.device_free(d)

    def batch_replay_overlay_ragged_dev(self, d_pts, tables, want_poses=True):
        """Ragged batch WITH the map insert on slots [0, n_scans): `tables` from ragged_tables() (with imus= or kins= for the messages between
        the buckets); every scan on its own copy-on-write overlay of the handle's map (lk_batch_replay_overlay_ragged_dev)."""
        t = tables
        n_scans = t["n_scans"]
        poses = (abi.lk_pose * n_scans)() if want_poses else None
        kind, n_msg, msgs = (2, t["n_kin"], t["kins"]) if "n_kin" in t else (1, t["n_imu"], t["imus"]) if "n_imu" in t else (0, None, None)
        self._chk(self.L.lk_batch_replay_overlay_ragged_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), _p(t["scan_off"]), _p(t["n_buckets"]),
                                                            _p(t["bucket_off"]), _p(t["bucket_dt"]), _p(t["t_begin"]),
                                                            _p(n_msg) if n_msg is not None else None, _p(msgs) if msgs is not None else None,
                                                            C.c_int(kind), poses))
        return poses

    def batch_replay_overlay_ragged(self, scans, t_begins, xs=None, Ps=None, imus=None, kins=None):
        """Convenience: host scans (lk_point arrays, time-sorted, any sizes) -> HBM, buckets = runs of equal curvature (KILO.cc:375-378),
        optional priors and per-scan messages, ragged replay WITH insert.  Returns the poses."""
        from . import synth

        if xs is not None:
            self.batch_set_priors(np.asarray(xs), np.asarray(Ps))
        allpts = np.ascontiguousarray(np.concatenate(scans))
        scan_off = np.r_[0, np.cumsum([len(sc) for sc in scans])]
        tabs = [synth.buckets_of(sc) for sc in scans]
        tables = self.ragged_tables(scan_off, [t[0] for t in tabs], [t[1] for t in tabs], t_begins, imus, kins)
        d = self.device_malloc(allpts.nbytes)
        try:
            self.h2d(d, allpts)
            return self.batch_replay_overlay_ragged_dev(d, tables)
        finally:
            self.device_free(d)

    def overlay_reserve(self, roots_per_scan=0, nodes_per_scan=0, blocks_per_scan=0):
        self._chk(self.L.lk_overlay_reserve(self.h, C.c_uint32(roots_per_scan), C.c_uint32(nodes_per_scan), C.c_uint32(blocks_per_scan)))

    def overlay_export(self, slot):
        """Map blob of the voxels scan `slot` of the last overlay replay holds privately."""
        nbytes = C.c_size_t(0)
        self._chk(self.L.lk_overlay_export(self.h, C.c_uint32(slot), None, C.byref(nbytes)))
        buf = np.zeros(nbytes.value, dtype=np.uint8)
        self._chk(self.L.lk_overlay_export(self.h, C.c_uint32(slot), _p(buf), C.byref(nbytes)))
        return buf

    def overlay_stats(self):
        """(max roots, max nodes, max point blocks) any scan's overlay reached in the last overlay replay."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.lk_overlay_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def overlay_resident_rounds(self):
        """Launches of the scan-resident kernel in the last recorded-run replay with insert (0: it ran launch by launch)."""
        r = C.c_uint32(0)
        self._chk(self.L.lk_overlay_resident_rounds(self.h, C.byref(r)))
        return int(r.value)

    def overlay_pool_bytes(self):
        """(bytes the overlay pools hold, per-scan root-table entries, child nodes, point blocks)."""
        n, a, b, c = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._chk(self.L.lk_overlay_pool_bytes(self.h, C.byref(n), C.byref(a), C.byref(b), C.byref(c)))
        return n.value, a.value, b.value, c.value

    @staticmethod
    def ragged_tables(scan_off, bucket_offs, bucket_dts, t_begins, imus=None, kins=None):
        """Flatten per-scan bucket tables into the arrays lk_batch_replay_ragged_dev takes (do this once per recorded run,
        not per replay): scan s = points [scan_off[s], scan_off[s+1]) with bucket bounds bucket_offs[s] (n_b + 1 offsets
        relative to the scan), time offsets bucket_dts[s] (n_b) and start time t_begins[s]."""
        n_scans = len(bucket_dts)
        so = np.ascontiguousarray(scan_off, dtype=np.uint64)
        assert len(so) == n_scans + 1 and len(bucket_offs) == n_scans and len(t_begins) == n_scans
        nb = np.fromiter((len(d) for d in bucket_dts), dtype=np.uint32, count=n_scans)
        off = np.ascontiguousarray(np.concatenate(bucket_offs), dtype=np.uint32)
        dt = np.ascontiguousarray(np.concatenate(bucket_dts), dtype=np.float64)
        assert len(off) == int(nb.sum()) + n_scans and len(dt) == int(nb.sum())
        tab = dict(n_scans=n_scans, scan_off=so, n_buckets=nb, bucket_off=off, bucket_dt=dt, t_begin=_f64(t_begins))
        if imus is not None:   # per-scan lk_imu arrays (stamp, acc, gyr), time-sorted: applied between the buckets (KILO.cc:379-383)
            assert len(imus) == n_scans
            tab["n_imu"] = np.fromiter((len(m) for m in imus), dtype=np.uint32, count=n_scans)
            tab["imus"] = np.ascontiguousarray(np.concatenate([np.asarray(m) for m in imus])) if tab["n_imu"].sum() else np.zeros(0, dtype=np.float64)
            assert tab["imus"].nbytes == 56 * int(tab["n_imu"].sum())
        if kins is not None:   # per-scan lk_kin_imu arrays (leg-fusion mode, KILO.cc:384-390)
            assert len(kins) == n_scans and imus is None
            tab["n_kin"] = np.fromiter((len(m) for m in kins), dtype=np.uint32, count=n_scans)
            tab["kins"] = np.ascontiguousarray(np.concatenate([np.asarray(m) for m in kins])) if tab["n_kin"].sum() else np.zeros(0, dtype=np.float64)
            assert tab["kins"].nbytes == 264 * int(tab["n_kin"].sum())
        return tab

    def batch_replay_ragged_dev(self, d_pts, tables, want_poses=True):
        """Ragged batch on slots [0, n_scans): `tables` from ragged_tables()."""
        t = tables
        n_scans = t["n_scans"]
        poses = (abi.lk_pose * n_scans)() if want_poses else None
        if "n_kin" in t:
            self._chk(self.L.lk_batch_replay_ragged_kin_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), _p(t["scan_off"]),
                                                            _p(t["n_buckets"]), _p(t["bucket_off"]), _p(t["bucket_dt"]), _p(t["t_begin"]),
                                                            _p(t["n_kin"]), _p(t["kins"]), poses))
        elif "n_imu" in t:
            self._chk(self.L.lk_batch_replay_ragged_imu_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), _p(t["scan_off"]),
                                                            _p(t["n_buckets"]), _p(t["bucket_off"]), _p(t["bucket_dt"]), _p(t["t_begin"]),
                                                            _p(t["n_imu"]), _p(t["imus"]), poses))
        else:
            self._chk(self.L.lk_batch_replay_ragged_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), _p(t["scan_off"]), _p(t["n_buckets"]),
                                                        _p(t["bucket_off"]), _p(t["bucket_dt"]), _p(t["t_begin"]), poses))
        return poses

    def batch_replay_scans_dev(self, d_pts, scan_off, t_begins, imus=None, kins=None, want_poses=True):
        """Ragged batch on slots [0, n_scans) with the bucket tables built ON THE DEVICE (lk_batch_replay_scans_dev): the host
        passes the scans' offsets and start times only.  imus / kins: per-scan message arrays (or None)."""
        so = np.ascontiguousarray(scan_off, dtype=np.uint64)
        n_scans = len(so) - 1
        tb = _f64(t_begins)
        assert len(tb) == n_scans and not (imus is not None and kins is not None)
        kind, n_msg, flat = 0, None, None
        msgs = imus if imus is not None else kins
        if msgs is not None:
            kind = 1 if imus is not None else 2
            n_msg = np.fromiter((len(m) for m in msgs), dtype=np.uint32, count=n_scans)
            flat = np.ascontiguousarray(np.concatenate([np.asarray(m) for m in msgs])) if n_msg.sum() else np.zeros(1, dtype=np.float64)
        poses = (abi.lk_pose * n_scans)() if want_poses else None
        self._chk(self.L.lk_batch_replay_scans_dev(self.h, C.c_void_p(d_pts), C.c_size_t(n_scans), _p(so), _p(tb), C.c_int(kind), _p(n_msg),
                                                   _p(flat), poses))
        return poses

    def batch_replay_ragged(self, scans, t_begins, xs=None, Ps=None, imus=None, kins=None, host_tables=False):
        """Convenience: host scans (lists of lk_point arrays, time-sorted) -> HBM, buckets = runs of equal curvature
        (KILO.cc:375-378), optional priors, ragged replay.  Returns the poses.  The bucket tables are built on the device
        (lk_batch_replay_scans_dev); host_tables=True builds them here and goes through lk_batch_replay_ragged(_imu/_kin)_dev
        (same results, bit for bit)."""
        from . import synth

        if xs is not None:
            self.batch_set_priors(np.asarray(xs), np.asarray(Ps))
        allpts = np.ascontiguousarray(np.concatenate(scans))
        scan_off = np.r_[0, np.cumsum([len(sc) for sc in scans])]
        d = self.device_malloc(allpts.nbytes)
        try:
            self.h2d(d, allpts)
            if not host_tables:
                return self.batch_replay_scans_dev(d, scan_off, t_begins, imus=imus, kins=kins)
            tabs = [synth.buckets_of(sc) for sc in scans]
            return self.batch_replay_ragged_dev(d, self.ragged_tables(scan_off, [t[0] for t in tabs], [t[1] for t in tabs], t_begins, imus, kins))
        finally:
            self.device_free(d)

    # ---- measurement / memory hooks ----
    def profile_enable(self, on):
        self._chk(self.L.lk_profile_enable(self.h, int(on)))

    def profile_get(self, name):
        n, ms = C.c_uint64(), C.c_double()
        self._chk(self.L.lk_profile_get(self.h, name.encode(), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_reset(self):
        self._chk(self.L.lk_profile_reset(self.h))

    def device_malloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.lk_device_malloc(self.h, C.byref(p), C.c_size_t(nbytes)))
        return p.value

    def device_free(self, ptr):
        self._chk(self.L.lk_device_free(self.h, C.c_void_p(ptr)))

    def h2d(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.lk_memcpy_h2d(self.h, C.c_void_p(d_ptr), _p(arr), C.c_size_t(arr.nbytes)))

    def d2h(self, arr, d_ptr):
        assert arr.flags["C_CONTIGUOUS"]
        self._chk(self.L.lk_memcpy_d2h(self.h, _p(arr), C.c_void_p(d_ptr), C.c_size_t(arr.nbytes)))

    def synchronize(self):
        self._chk(self.L.lk_synchronize(self.h))

    def stream(self):
        return self.L.lk_stream(self.h)

    def stream_pipeline(self, on):
        """Pipelined stream path on / off (lk_stream_pipeline)."""
        self._chk(self.L.lk_stream_pipeline(self.h, C.c_int(1 if on else 0)))

    def stream_resident(self, on):
        """Scan-resident stream kernel on / off (lk_stream_resident)."""
        self._chk(self.L.lk_stream_resident(self.h, C.c_int(1 if on else 0)))

    def stream_grid(self, mode):
        """Grid-resident stream kernel for scans of large buckets (lk_stream_grid): 0 / False never, 1 / True buckets up to 4096 points, 2 any size."""
        self._chk(self.L.lk_stream_grid(self.h, C.c_int(int(mode))))

    def stream_grid_placement(self):
        """Bit mask of the XCC ids the working blocks of the last one-XCD grid-resident launch ran on (lk_stream_grid_placement)."""
        m = C.c_uint32(0)
        self._chk(self.L.lk_stream_grid_placement(self.h, C.byref(m)))
        return int(m.value)

    def test_stall(self, bound_ms):
        """Fault injection for the LK_ERR_TIMEOUT path of the resident stream kernels (include/legkilo_hip.h: lk_test_stall); 0 = off."""
        self._chk(self.L.lk_test_stall(self.h, C.c_uint(int(bound_ms))))

    def stream_resident_stats(self):
        """(scans through the scan-resident kernel, launches beyond one per scan: its fallback rounds)."""
        out = np.zeros(2, dtype=np.uint64)
        self._chk(self.L.lk_stream_resident_stats(self.h, _p(out)))
        return int(out[0]), int(out[1])

    def stream_resident_redo(self):
        """Buckets the scan-resident kernel's filter wave evaluated again after a conflicting insert (lk_stream_stats' fourth word)."""
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.L.lk_stream_stats(self.h, _p(out)))
        return int(out[3])

    def stream_stats(self):
        """(pipelined buckets, their residual tiles, tiles the verify pass evaluated again)."""
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.L.lk_stream_stats(self.h, _p(out)))
        return int(out[0]), int(out[1]), int(out[2])
