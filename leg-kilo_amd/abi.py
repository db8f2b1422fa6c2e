"""ctypes mirrors of the PODs in include/legkilo_hip.h (keep field order identical)."""
import ctypes as C

LK_DIM_STATE = 30
LK_STATE_DOUBLES = 36
LK_BLOCK_PTS = 52
LK_BLOB_MAGIC = 0x4C4B4D50

LK_PLANE_IS_PLANE = 1
LK_PLANE_IS_INIT = 2
LK_NODE_INIT_OCTO = 1
LK_NODE_UPDATE_ENABLE = 2
LK_NODE_OCTO_STATE = 4
LK_NODE_PTS_DROPPED = 8


class lk_config(C.Structure):
    _fields_ = [
        ("vel_process_cov", C.c_double),
        ("imu_acc_process_cov", C.c_double),
        ("imu_gyr_process_cov", C.c_double),
        ("contact_process_cov", C.c_double),
        ("acc_bias_process_cov", C.c_double),
        ("gyr_bias_process_cov", C.c_double),
        ("kin_bias_process_cov", C.c_double),
        ("imu_acc_meas_noise", C.c_double),
        ("imu_acc_z_meas_noise", C.c_double),
        ("imu_gyr_meas_noise", C.c_double),
        ("kin_meas_noise", C.c_double),
        ("chd_meas_noise", C.c_double),
        ("contact_meas_noise", C.c_double),
        ("lidar_point_meas_ratio", C.c_double),
        ("max_voxel_size", C.c_double),
        ("planner_threshold", C.c_double),
        ("beam_err", C.c_double),
        ("dept_err", C.c_double),
        ("sigma_num", C.c_double),
        ("max_layer", C.c_int32),
        ("max_iterations", C.c_int32),
        ("layer_init_num", C.c_int32 * 5),
        ("max_points_num", C.c_int32),
        ("ext_R", C.c_double * 9),
        ("ext_T", C.c_double * 3),
        ("gravity", C.c_double),
        ("device_id", C.c_int32),
        ("n_slots", C.c_uint32),
        ("max_roots", C.c_uint32),
        ("max_nodes", C.c_uint32),
        ("max_point_blocks", C.c_uint32),
        ("max_scan_points", C.c_uint32),
    ]


class lk_point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("curvature", C.c_float)]


class lk_imu(C.Structure):
    _fields_ = [("stamp", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3)]


class lk_kin_imu(C.Structure):
    _fields_ = [
        ("time_stamp", C.c_double),
        ("foot_pos", (C.c_double * 3) * 4),
        ("foot_vel", (C.c_double * 3) * 4),
        ("contact", C.c_int32 * 4),
        ("acc", C.c_double * 3),
        ("gyr", C.c_double * 3),
    ]


class lk_pose(C.Structure):
    _fields_ = [
        ("rot", C.c_double * 9),
        ("pos", C.c_double * 3),
        ("vel", C.c_double * 3),
        ("n_effect", C.c_uint64),
        ("n_buckets", C.c_uint32),
        ("n_updates", C.c_uint32),
    ]


def pose_dtype():
    """numpy view of lk_pose (136 B) for vectorised access to pose buffers."""
    import numpy as np

    return np.dtype([("rot", "<f8", 9), ("pos", "<f8", 3), ("vel", "<f8", 3), ("n_effect", "<u8"), ("n_buckets", "<u4"), ("n_updates", "<u4")])


class lk_cloud_layout(C.Structure):
    _fields_ = [("point_step", C.c_uint32), ("off_x", C.c_uint32), ("off_y", C.c_uint32), ("off_z", C.c_uint32),
                ("off_time", C.c_uint32), ("lidar_type", C.c_int32)]


class lk_blob_header(C.Structure):
    _fields_ = [
        ("magic", C.c_uint32),
        ("version", C.c_uint32),
        ("n_roots", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("n_blocks", C.c_uint32),
        ("block_pts", C.c_uint32),
        ("voxel_size", C.c_double),
        ("max_layer", C.c_int32),
        ("max_points_num", C.c_int32),
        ("bytes", C.c_uint64),
    ]


def blob_header_dtype():
    """numpy view of lk_blob_header (48 B)."""
    import numpy as np

    dt = np.dtype([("magic", "<u4"), ("version", "<u4"), ("n_roots", "<u4"), ("n_nodes", "<u4"), ("n_blocks", "<u4"), ("block_pts", "<u4"),
                   ("voxel_size", "<f8"), ("max_layer", "<i4"), ("max_points_num", "<i4"), ("bytes", "<u8")])
    assert dt.itemsize == C.sizeof(lk_blob_header)
    return dt


# numpy dtypes of the blob records (for parsing exports in tests / tools)
def blob_dtypes():
    import numpy as np

    root = np.dtype([("key", "<i4", 3), ("node", "<i4")])
    plane = np.dtype(
        [
            ("center", "<f8", 3),
            ("normal", "<f8", 3),
            ("d", "<f4"),
            ("radius", "<f4"),
            ("flags", "<u4"),
            ("points_size", "<i4"),
            ("plane_var", "<f8", 21),
            ("min_ev", "<f4"),
            ("mid_ev", "<f4"),
            ("max_ev", "<f4"),
            ("pad", "<u4", 3),
        ]
    )
    node = np.dtype(
        [
            ("child", "<i4", 8),
            ("voxel_center", "<f8", 3),
            ("quater_length", "<f4"),
            ("layer", "<i4"),
            ("npts", "<i4"),
            ("new_points", "<i4"),
            ("state", "<u4"),
            ("block", "<i4"),
            ("key", "<i4", 3),
            ("list_head", "<i4"),
            ("pad", "<u4", 8),
        ]
    )
    pt = np.dtype([("pw", "<f8", 3), ("var", "<f8", 6)])
    block = np.dtype([("pts", pt, LK_BLOCK_PTS)])
    assert plane.itemsize == 256 and node.itemsize == 128 and pt.itemsize == 72
    return root, node, plane, block


def parse_blob(buf):
    """bytes -> dict(header, roots, nodes, planes, blocks) of numpy structured arrays."""
    import numpy as np

    hd = lk_blob_header.from_buffer_copy(bytes(buf[: C.sizeof(lk_blob_header)]))
    assert hd.magic == LK_BLOB_MAGIC, hex(hd.magic)
    root, node, plane, block = blob_dtypes()
    off = C.sizeof(lk_blob_header)
    mv = memoryview(buf)
    out = {"header": hd}
    for name, dt, n in (("roots", root, hd.n_roots), ("nodes", node, hd.n_nodes), ("planes", plane, hd.n_nodes),
                        ("blocks", block, hd.n_blocks)):
        out[name] = np.frombuffer(mv, dtype=dt, count=n, offset=off)
        off += dt.itemsize * n
    assert off == hd.bytes, (off, hd.bytes)
    return out
