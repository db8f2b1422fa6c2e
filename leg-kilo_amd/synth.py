"""Synthetic inputs for the path: the reference ships no recorded data.

World   : axis-aligned box room 40 x 30 x 8 m, ground at z=0, 6 interior wall slabs.
Sensor  : spinning-LiDAR ray model cast from a moving body (figure-eight trajectory),
          per-point time offsets, range noise.
Streams : IMU (acc, gyr) and kinematic+IMU (foot positions / velocities / contacts)
          consistent with the process/measurement models of eskf.cc:64-70 and KILO.cc:235-314.

Host-side restatements of the two steps that sit just BEFORE the path in the reference:
  preprocess_velodyne()  lidar_processing.cc:25-52 (filter_num, blind radius, 2 ms time bins)
  voxel_grid_centroid()  pcl::VoxelGrid centroid filter used at KILO.cc:356-360 (own definition of
                         the output order: ascending cell index, x fastest)
  foot_pos_vel()         kinematics.cc:54-90 (leg forward kinematics + Jacobian foot velocity)

Seeds: world 1001, scan 2002, noise 3003, trajectory 4004, batch 5005+i.
"""
import numpy as np

POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("curvature", "<f4")])
IMU_DTYPE = np.dtype([("stamp", "<f8"), ("acc", "<f8", 3), ("gyr", "<f8", 3)])
KIN_DTYPE = np.dtype([("time_stamp", "<f8"), ("foot_pos", "<f8", (4, 3)), ("foot_vel", "<f8", (4, 3)),
                      ("contact", "<i4", 4), ("acc", "<f8", 3), ("gyr", "<f8", 3)])
assert POINT_DTYPE.itemsize == 16 and IMU_DTYPE.itemsize == 56 and KIN_DTYPE.itemsize == 264

G_WORLD = np.array([0.0, 0.0, -9.81])


# --------------------------------------------------------------------------- world
class World:
    def __init__(self, seed=1001):
        rng = np.random.default_rng(seed)
        self.lo = np.array([-20.0, -15.0, 0.0])
        self.hi = np.array([20.0, 15.0, 8.0])
        base = np.array([
            [-16.0, -15.7, -8.0, 2.0, 0.0, 4.0],
            [15.5, 15.8, -3.0, 9.0, 0.0, 5.0],
            [-6.0, 4.0, 10.0, 10.3, 0.0, 3.5],
            [-3.0, 8.0, -10.3, -10.0, 0.0, 6.0],
            [4.5, 5.0, 6.5, 9.0, 0.0, 8.0],
            [-9.0, -8.5, -9.0, -6.5, 0.0, 8.0],
        ])
        jit = rng.uniform(-0.2, 0.2, size=(6, 2))
        base[:, 0:2] += jit[:, 0:1]
        base[:, 2:4] += jit[:, 1:2]
        self.slabs = base  # xmin xmax ymin ymax zmin zmax

    def raycast(self, o, d):
        """o, d: (N,3) origins (inside the room) and unit directions -> hit distance (N,)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            # room: leaving an axis-aligned box from the inside
            t_hi = (self.hi - o) * inv
            t_lo = (self.lo - o) * inv
            t_exit = np.where(d > 0, t_hi, np.where(d < 0, t_lo, np.inf))
            t = np.min(t_exit, axis=1)
            for s in self.slabs:
                lo = s[[0, 2, 4]]
                hi = s[[1, 3, 5]]
                t1 = (lo - o) * inv
                t2 = (hi - o) * inv
                tn = np.minimum(t1, t2)
                tf = np.maximum(t1, t2)
                par = d == 0
                inside = (o >= lo) & (o <= hi)
                tn = np.where(par, np.where(inside, -np.inf, np.inf), tn)
                tf = np.where(par, np.where(inside, np.inf, -np.inf), tf)
                te = np.max(tn, axis=1)
                tx = np.min(tf, axis=1)
                hit = (te < tx) & (te > 1e-6)
                t = np.where(hit & (te < t), te, t)
        return t


# --------------------------------------------------------------------------- trajectory
def _rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.empty(np.shape(yaw) + (3, 3))
    R[..., 0, 0] = cy * cp
    R[..., 0, 1] = cy * sp * sr - sy * cr
    R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp
    R[..., 1, 1] = sy * sp * sr + cy * cr
    R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp
    R[..., 2, 1] = cp * sr
    R[..., 2, 2] = cp * cr
    return R


class Trajectory:
    """Figure-eight (lemniscate of Gerono) at trotting height with small roll/pitch/bob."""

    def __init__(self, seed=4004, period=60.0, A=10.0, B=10.0):
        rng = np.random.default_rng(seed)
        self.T, self.A, self.B = period, A, B
        self.ph = rng.uniform(0, 2 * np.pi, size=3)
        self.z0 = 0.45

    def pos(self, t):
        t = np.asarray(t, dtype=np.float64)
        th = 2 * np.pi * t / self.T
        x = self.A * np.sin(th)
        y = 0.5 * self.B * np.sin(2 * th)
        z = self.z0 + 0.02 * np.sin(2 * np.pi * 2.0 * t + self.ph[0])
        return np.stack([x, y, z], axis=-1)

    def vel(self, t, h=1e-4):
        t = np.asarray(t, dtype=np.float64)
        return (self.pos(t + h) - self.pos(t - h)) / (2 * h)

    def acc(self, t, h=1e-3):
        t = np.asarray(t, dtype=np.float64)
        return (self.pos(t + h) - 2 * self.pos(t) + self.pos(t - h)) / (h * h)

    def rot(self, t):
        t = np.asarray(t, dtype=np.float64)
        th = 2 * np.pi * t / self.T
        vx = self.A * np.cos(th)
        vy = self.B * np.cos(2 * th)
        yaw = np.arctan2(vy, vx)
        roll = 0.03 * np.sin(2 * np.pi * 1.5 * t + self.ph[1])
        pitch = 0.02 * np.sin(2 * np.pi * 1.1 * t + self.ph[2])
        return _rot_zyx(yaw, pitch, roll)

    def omega_body(self, t, h=1e-5):
        """body-frame angular velocity, vee(R^T dR/dt) by central differences."""
        t = np.asarray(t, dtype=np.float64)
        R = self.rot(t)
        dR = (self.rot(t + h) - self.rot(t - h)) / (2 * h)
        W = np.swapaxes(R, -1, -2) @ dR
        return np.stack([W[..., 2, 1], W[..., 0, 2], W[..., 1, 0]], axis=-1)


# --------------------------------------------------------------------------- scans
def cast_scan(world, traj, t_begin, dirs_lidar, t_off, ext_R, ext_T, rng_noise, range_noise=0.02):
    """Cast unit directions (lidar frame) at per-ray times t_begin + t_off from the moving body.
    Returns xyz in the LIDAR frame (float64, (N,3)) and the hit mask."""
    ext_R = np.asarray(ext_R, dtype=np.float64).reshape(3, 3)
    ext_T = np.asarray(ext_T, dtype=np.float64)
    tt = t_begin + np.asarray(t_off, dtype=np.float64)
    R = traj.rot(tt)
    p = traj.pos(tt)
    o = p + np.einsum("nij,j->ni", R, ext_T)
    d = np.einsum("nij,nj->ni", R, dirs_lidar @ ext_R.T)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = world.raycast(o, d)
    ok = np.isfinite(t) & (t > 0.05) & (t < 200.0)
    rng = t + rng_noise.normal(0.0, range_noise, size=t.shape)
    return dirs_lidar * rng[:, None], ok


def vlp16_scan(world, traj, t_begin, cfg, seed_noise=3003, n_az=1800, period=0.1):
    """Config 1: 16 rings in [-15,15] deg x 1800 azimuth columns = 28 800 rays over `period` s.
    Output: raw cloud as POINT_DTYPE with curvature = time offset in seconds (f32, unquantised)."""
    rings = np.deg2rad(np.linspace(-15.0, 15.0, 16))
    az = np.arange(n_az) * (2 * np.pi / n_az)
    A, E = np.meshgrid(az, rings, indexing="ij")  # column-major in time: all rings of one azimuth together
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    t_off = np.repeat(np.arange(n_az) * (period / n_az), 16)
    rngn = np.random.default_rng(seed_noise)
    xyz, ok = cast_scan(world, traj, t_begin, dirs, t_off, cfg["extrinsic_R"], cfg["extrinsic_T"], rngn)
    out = np.zeros(int(ok.sum()), dtype=POINT_DTYPE)
    out["x"], out["y"], out["z"] = xyz[ok, 0], xyz[ok, 1], xyz[ok, 2]
    out["curvature"] = t_off[ok].astype(np.float32)
    return out


def ouster_scan(world, traj, t_begin, cfg, seed_noise=3003, n_cols=1024, n_rings=64, vfov_deg=(-22.5, 22.5), period=0.1):
    """Config 4: an OS1-64-like scan, 64 rings x 1024 azimuth columns = 65 536 rays over `period` s, column by
    column (all rings of one azimuth share a time stamp), cast from the moving body.  Output: raw cloud as POINT_DTYPE with
    curvature = the column's time offset in seconds (f32) and the same offsets as integer nanoseconds (the `t` field of the
    Ouster PointCloud2 layout, lidar_processing.cc:54-80, time_scale 1e-9)."""
    rings = np.deg2rad(np.linspace(vfov_deg[0], vfov_deg[1], n_rings))
    az = np.arange(n_cols) * (2 * np.pi / n_cols)
    A, E = np.meshgrid(az, rings, indexing="ij")
    dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], axis=-1).reshape(-1, 3)
    t_ns = np.repeat((np.arange(n_cols) * (period * 1e9 / n_cols)).astype(np.uint32), n_rings)
    t_off = t_ns.astype(np.float64) * 1e-9
    rngn = np.random.default_rng(seed_noise)
    xyz, ok = cast_scan(world, traj, t_begin, dirs, t_off, cfg["extrinsic_R"], cfg["extrinsic_T"], rngn)
    out = np.zeros(int(ok.sum()), dtype=POINT_DTYPE)
    out["x"], out["y"], out["z"] = xyz[ok, 0], xyz[ok, 1], xyz[ok, 2]
    out["curvature"] = t_off[ok].astype(np.float32)
    return out, t_ns[ok]


def dense_scan(world, traj, t_begin, cfg, n=100_000, n_buckets=5, seed_scan=2002, seed_noise=3003, period=0.1,
               blind=1.5, layout="cell"):
    """Configs 2/3/5: n points ENTERING the path (post-downsample), random ray directions
    (azimuth uniform, elevation in [-25, 40] deg), time bins by azimuth rank: n_buckets runs of
    equal curvature k * period / n_buckets... quantised like lidar_processing.cc:48 when n_buckets=51.
    n_buckets == 1 gives one state (config 2).  Returned cloud is already time-sorted."""
    rngs = np.random.default_rng(seed_scan)
    rngn = np.random.default_rng(seed_noise)
    pts = np.zeros(0, dtype=POINT_DTYPE)
    need = n
    chunks = []
    while need > 0:
        m = int(need * 1.15) + 64
        az = rngs.uniform(0, 2 * np.pi, m)
        el = np.deg2rad(rngs.uniform(-25.0, 40.0, m))
        dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1)
        t_off = az / (2 * np.pi) * period
        xyz, ok = cast_scan(world, traj, t_begin, dirs, t_off, cfg["extrinsic_R"], cfg["extrinsic_T"], rngn)
        ok &= (xyz ** 2).sum(1) > blind * blind
        c = np.zeros(int(ok.sum()), dtype=POINT_DTYPE)
        c["x"], c["y"], c["z"] = xyz[ok, 0], xyz[ok, 1], xyz[ok, 2]
        c["curvature"] = t_off[ok].astype(np.float32)
        chunks.append(c[:need])
        need -= len(chunks[-1])
    pts = np.concatenate(chunks)
    order = np.argsort(pts["curvature"], kind="stable")
    pts = pts[order]
    if n_buckets == 51:
        pts["curvature"] = np.round(pts["curvature"] * np.float32(500.0)) / np.float32(500.0)
    else:
        b = (np.arange(n) * n_buckets) // n
        pts["curvature"] = (b * (period / n_buckets)).astype(np.float32)
    if layout == "cell":
        # Inside a run of equal time stamps the reference's order is whatever pcl::VoxelGrid emitted (ascending cell
        # index, x fastest; the std::sort by time is over equal keys there).  Reproduce that spatially coherent order:
        # sort each bucket by the body-frame cell index of the yaml voxel_grid_resolution.
        leaf = float(cfg.get("voxel_grid_resolution", 0.3))
        xyz = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
        inv = np.float32(1.0) / np.float32(leaf)
        ijk = np.floor(xyz * inv).astype(np.int64)
        ijk -= ijk.min(0)
        div = ijk.max(0) + 1
        cell = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
        pts = pts[np.lexsort((cell, pts["curvature"]))]
    # stable sort again (rounding keeps monotonic order, but make the contract explicit)
    pts = pts[np.argsort(pts["curvature"], kind="stable")]
    return pts


def preprocess_velodyne(raw, filter_num=3, blind=1.5):
    """lidar_processing.cc:25-52: keep every filter_num-th point outside the blind radius,
    curvature = round((t - t_first) * 500) / 500 in f32."""
    idx = np.arange(len(raw))
    r2 = raw["x"] * raw["x"] + raw["y"] * raw["y"] + raw["z"] * raw["z"]
    keep = (idx % filter_num == 0) & ~(np.float32(blind * blind) > r2)
    out = raw[keep].copy()
    first = raw["curvature"][0]
    v = ((out["curvature"] - first) * np.float32(500.0)).astype(np.float32)
    r = (np.floor(np.abs(v.astype(np.float64)) + 0.5) * np.sign(v)).astype(np.float32)  # std::round, half away from zero
    out["curvature"] = r / np.float32(500.0)
    return out


def voxel_grid_centroid(pts, leaf):
    """Centroid of all fields (x, y, z, curvature) per leaf cell — the behaviour of pcl::VoxelGrid that the
    reference relies on at KILO.cc:356-360 (it de-quantises the time stamps).  Host definition of what
    lk_preprocess_scan computes on the device: float32 cell index floor(p * (1/leaf)) - min, cells in ascending
    index (x fastest), points of a cell summed sequentially in input order in float32."""
    x = np.stack([pts["x"], pts["y"], pts["z"]], axis=1).astype(np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = np.floor(x.min(0) * inv).astype(np.int64)
    mx = np.floor(x.max(0) * inv).astype(np.int64)
    div = mx - mn + 1
    ijk = np.floor(x * inv).astype(np.int64) - mn
    idx = ijk[:, 0] + ijk[:, 1] * div[0] + ijk[:, 2] * div[0] * div[1]
    order = np.argsort(idx, kind="stable")
    idx_s = idx[order]
    starts = np.flatnonzero(np.r_[True, idx_s[1:] != idx_s[:-1]])
    cnt = np.diff(np.r_[starts, len(idx_s)])
    out = np.zeros(len(starts), dtype=POINT_DTYPE)
    kmax = int(cnt.max())
    for name in ("x", "y", "z", "curvature"):
        v = pts[name][order].astype(np.float32)
        acc = np.zeros(len(starts), dtype=np.float32)
        for k in range(kmax):
            m = cnt > k
            acc[m] = acc[m] + v[starts[m] + k]
        out[name] = acc / cnt.astype(np.float32)
    return out


def sort_by_time(pts):
    """The std::sort of KILO.cc:369-370, made deterministic (stable)."""
    return pts[np.argsort(pts["curvature"], kind="stable")]


def buckets_of(pts):
    """Runs of exactly equal curvature (KILO.cc:375-378): (offsets[nb+1] u32, dt[nb] f64)."""
    c = pts["curvature"]
    starts = np.flatnonzero(np.r_[True, c[1:] != c[:-1]])
    off = np.r_[starts, len(c)].astype(np.uint32)
    dt = c[starts].astype(np.float64)
    return off, dt


# --------------------------------------------------------------------------- inertial / kinematic streams
def imu_stream(traj, t0, t1, rate=200.0, seed=3003, acc_noise=0.02, gyr_noise=0.002, ba=(0, 0, 0), bw=(0, 0, 0)):
    """acc = R^T (a_w - g_w) + ba + n,  gyr = omega_body + bw + n   (v' = R*imu_a + grav, eskf.cc:68)."""
    rng = np.random.default_rng(seed + 17)
    n = int(np.floor((t1 - t0) * rate))
    t = t0 + (np.arange(n) + 0.5) / rate
    R = traj.rot(t)
    a = np.einsum("nji,nj->ni", R, traj.acc(t) - G_WORLD) + np.asarray(ba) + rng.normal(0, acc_noise, (n, 3))
    w = traj.omega_body(t) + np.asarray(bw) + rng.normal(0, gyr_noise, (n, 3))
    out = np.zeros(n, dtype=IMU_DTYPE)
    out["stamp"], out["acc"], out["gyr"] = t, a, w
    return out


def foot_pos_vel(q, dq, p):
    """kinematics.cc:54-90 — leg order FR FL RR RL; q, dq: (4,3) hip/thigh/calf angles and rates."""
    lt, lc, d, ox, oy = p["leg_thigh_length"], p["leg_calf_length"], p["leg_thigh_offset"], p["leg_offset_x"], p["leg_offset_y"]
    pos = np.zeros((4, 3))
    vel = np.zeros((4, 3))
    jac = np.zeros((4, 3, 3))
    for i in range(4):
        lfoot = 1 if i in (0, 2) else -1
        ffoot = 1 if i < 2 else -1
        s1, s2, s23 = np.sin(q[i, 0]), np.sin(q[i, 1]), np.sin(q[i, 1] + q[i, 2])
        c1, c2, c23 = np.cos(q[i, 0]), np.cos(q[i, 1]), np.cos(q[i, 1] + q[i, 2])
        pos[i, 0] = -lt * s2 - lc * s23 + ffoot * ox
        pos[i, 1] = lfoot * d * c1 + lc * s1 * c23 + lt * c2 * s1 + lfoot * oy
        pos[i, 2] = lfoot * d * s1 - lc * c1 * c23 - lt * c1 * c2
        J = np.array([
            [0.0, -lc * c23 - lt * c2, -lc * c23],
            [lt * c1 * c2 - lfoot * d * s1 + lc * c1 * c23, -s1 * (lc * s23 + lt * s2), -lc * s23 * s1],
            [lt * c2 * s1 + lfoot * d * c1 + lc * s1 * c23, c1 * (lc * s23 + lt * s2), lc * s23 * c1],
        ])
        jac[i] = J
        vel[i, 0] = J[0, 1] * dq[i, 1] + J[0, 2] * dq[i, 2]
        vel[i, 1] = J[1, 0] * dq[i, 0] + J[1, 1] * dq[i, 1] + J[1, 2] * dq[i, 2]
        vel[i, 2] = J[2, 0] * dq[i, 0] + J[2, 1] * dq[i, 1] + J[2, 2] * dq[i, 2]
    return pos, vel, jac


def kin_stream(traj, t0, t1, params, rate=500.0, seed=3003, acc_noise=0.02, gyr_noise=0.002, vel_noise=0.01):
    """500 Hz KinImuMeas stream, trot gait: diagonal pairs (FR,RL)/(FL,RR) alternate, duty 0.6.
    A stance foot does not slip:  v_w + R (w x p_f + v_f) = 0  (the model behind KILO.cc:297-303);
    joint rates are solved from the leg Jacobian so that foot_vel is what kinematics.cc would output."""
    rng = np.random.default_rng(seed + 29)
    n = int(np.floor((t1 - t0) * rate))
    t = t0 + (np.arange(n) + 0.5) / rate
    R = traj.rot(t)
    vw = traj.vel(t)
    w = traj.omega_body(t)
    a = np.einsum("nji,nj->ni", R, traj.acc(t) - G_WORLD)
    out = np.zeros(n, dtype=KIN_DTYPE)
    out["time_stamp"] = t
    out["acc"] = a + rng.normal(0, acc_noise, (n, 3))
    out["gyr"] = w + rng.normal(0, gyr_noise, (n, 3))
    gait_T = 0.5
    for k in range(n):
        ph = (t[k] / gait_T) % 1.0
        ph2 = (ph + 0.5) % 1.0
        contact = np.array([ph < 0.6, ph2 < 0.6, ph2 < 0.6, ph < 0.6])  # FR FL RR RL
        sway = 0.15 * np.sin(2 * np.pi * t[k] / gait_T + np.array([0, np.pi, np.pi, 0]))
        q = np.stack([np.full(4, 0.02), 0.8 + sway, np.full(4, -1.6)], axis=1)
        pos, _, jac = foot_pos_vel(q, np.zeros((4, 3)), params)
        vb = -(R[k].T @ vw[k])  # body-frame velocity a non-slipping foot must show, before the w x p term
        dq = np.zeros((4, 3))
        for i in range(4):
            vf = vb - np.cross(w[k], pos[i])
            if not contact[i]:
                vf = vf + np.array([1.2, 0.0, 0.3 * np.cos(2 * np.pi * ph)])  # swing leg moves forward
            dq[i] = np.linalg.solve(jac[i], vf)
        pos, vel, _ = foot_pos_vel(q, dq, params)
        out["foot_pos"][k] = pos
        out["foot_vel"][k] = vel + rng.normal(0, vel_noise, (4, 3)) * contact[:, None]
        out["contact"][k] = contact.astype(np.int32)
    return out


# --------------------------------------------------------------------------- filter priors
def initial_state(traj, t, params, perturb_rng=None, sig_pos=0.0, sig_ang_deg=0.0):
    """x36 at trajectory time t (true pose, velocity, gravity; imu_a/imu_w from the true motion),
    optionally perturbed (config 5 priors: pose (+) N(0, 2 cm / 0.5 deg))."""
    R = traj.rot(t)
    p = traj.pos(t)
    v = traj.vel(t)
    x = np.zeros(36)
    if perturb_rng is not None:
        p = p + perturb_rng.normal(0, sig_pos, 3)
        ang = np.deg2rad(perturb_rng.normal(0, sig_ang_deg, 3))
        th = np.linalg.norm(ang)
        if th > 0:
            k = ang / th
            K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
            R = R @ (np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K)
    x[0:9] = R.reshape(-1)
    x[9:12] = p
    x[12:15] = v
    x[21:24] = G_WORLD * (params["gravity"] / 9.81)
    x[24:27] = R.T @ (traj.acc(t) - G_WORLD)
    x[27:30] = traj.omega_body(t)
    return x
