"""Parameter sets for the path, keyed exactly like the reference's YAML files
(legkilo/config/leg_fusion.yaml, diter.yaml; read at KILO.cc:29-82).  Only keys the
path (or its synthetic inputs) consumes are listed; unused reference keys stay inert.
"""
import ctypes as C

from .abi import lk_config

# values of legkilo/config/leg_fusion.yaml (Go1 + VLP-16)
LEG_FUSION = dict(
    only_imu_use=False, gravity=9.81,
    extrinsic_T=[0.0, 0.0, 0.20], extrinsic_R=[1, 0, 0, 0, 1, 0, 0, 0, 1],
    lidar_type=1, time_scale=1.0, blind=1.5, filter_num=3, voxel_grid_resolution=0.3,
    max_layer=2, voxel_size=0.5, min_eigen_value=0.01, sigma_num=3, beam_err=0.2, dept_err=0.04,
    layer_init_num=[5, 5, 5, 5, 5], max_points_num=50,
    leg_offset_x=0.1881, leg_offset_y=0.04675, leg_calf_length=0.213, leg_thigh_length=0.213,
    leg_thigh_offset=0.08, contact_force_threshold_up=220, contact_force_threshold_down=200,
    vel_process_cov=20, imu_acc_process_cov=500, imu_gyr_process_cov=1000, contact_process_cov=20,
    acc_bias_process_cov=0.001, gyr_bias_process_cov=0.001, kin_bias_process_cov=0.001,
    imu_acc_meas_noise=0.1, imu_acc_z_meas_noise=1.0, imu_gyr_meas_noise=0.01, kin_meas_noise=0.1,
    chd_meas_noise=0.1, contact_meas_noise=0.001, lidar_point_meas_ratio=10,
)

# legkilo/config/diter.yaml (Go2 + Ouster) = leg_fusion with these differences
DITER = dict(
    LEG_FUSION,
    extrinsic_T=[0.005, 0.00056, 0.299], lidar_type=2, time_scale=1e-9, voxel_grid_resolution=0.5,
    leg_offset_x=0.1934, leg_offset_y=0.0465, leg_thigh_offset=0.0465,
    contact_force_threshold_up=40, contact_force_threshold_down=60,
    imu_acc_meas_noise=0.01, imu_acc_z_meas_noise=0.1, imu_gyr_meas_noise=0.001,
)


def load_yaml(path):
    """Read a reference-style flat YAML file into a parameter dict (missing keys -> LEG_FUSION)."""
    import yaml

    with open(path) as f:
        y = yaml.safe_load(f)
    p = dict(LEG_FUSION)
    p.update({k: v for k, v in y.items() if k in p})
    return p


def make_config(params=None, device_id=0, n_slots=1, max_roots=1 << 18, max_nodes=1 << 19,
                max_point_blocks=1 << 18, max_scan_points=1 << 17):
    p = dict(LEG_FUSION if params is None else params)
    c = lk_config()
    for k in ("vel_process_cov", "imu_acc_process_cov", "imu_gyr_process_cov", "contact_process_cov",
              "acc_bias_process_cov", "gyr_bias_process_cov", "kin_bias_process_cov", "imu_acc_meas_noise",
              "imu_acc_z_meas_noise", "imu_gyr_meas_noise", "kin_meas_noise", "chd_meas_noise",
              "contact_meas_noise", "lidar_point_meas_ratio", "beam_err", "dept_err", "sigma_num", "gravity"):
        setattr(c, k, float(p[k]))
    c.max_voxel_size = float(p["voxel_size"])
    c.planner_threshold = float(p["min_eigen_value"])
    c.max_layer = int(p["max_layer"])
    c.max_iterations = 0
    c.layer_init_num = (C.c_int32 * 5)(*[int(v) for v in p["layer_init_num"]])
    c.max_points_num = int(p["max_points_num"])
    c.ext_R = (C.c_double * 9)(*[float(v) for v in p["extrinsic_R"]])
    c.ext_T = (C.c_double * 3)(*[float(v) for v in p["extrinsic_T"]])
    c.device_id = device_id
    c.n_slots = n_slots
    c.max_roots = max_roots
    c.max_nodes = max_nodes
    c.max_point_blocks = max_point_blocks
    c.max_scan_points = max_scan_points
    return c
