"""TUM trajectory I/O and ATE — the on-disk format the parity metric of the north star
("trajectory ATE delta < 1 mm") is defined on.

write_tum() follows TrajectorySaver::write (legkilo/src/common/trajectory_saver.hpp:43-50):
    `timestamp tx ty tz qx qy qz qw`, fixed notation, 9 decimals, rotation = body -> world,
    quaternion from the rotation matrix the way Eigen::Quaterniond(Matrix3d) does (Shepperd's branches).
ate() is the usual absolute trajectory error: associate by time stamp, optionally align with the
closed-form rigid (Horn / Umeyama without scale) transform, RMSE of the position differences.
Host-side utility; no device code is involved.
"""
import numpy as np


def rot_to_quat(R):
    """Eigen::Quaterniond(Matrix3d): returns (x, y, z, w)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        x, y, z = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = [0.0, 0.0, 0.0]
        q[i] = 0.5 * t
        t = 0.5 / t
        w = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
        x, y, z = q
    return x, y, z, w


def write_tum(path, stamps, rots, poss):
    with open(path, "w") as f:
        for t, R, p in zip(stamps, rots, poss):
            x, y, z, w = rot_to_quat(R)
            f.write("%.9f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (t, p[0], p[1], p[2], x, y, z, w))


def read_tum(path):
    a = np.loadtxt(path, ndmin=2)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def associate(ta, tb, max_dt=0.01):
    """Greedy nearest-stamp association (both sorted): index pairs with |dt| <= max_dt."""
    ia, ib, out = 0, 0, []
    while ia < len(ta) and ib < len(tb):
        d = ta[ia] - tb[ib]
        if abs(d) <= max_dt:
            out.append((ia, ib))
            ia += 1
            ib += 1
        elif d < 0:
            ia += 1
        else:
            ib += 1
    return np.array(out, dtype=int).reshape(-1, 2)


def ate(pa, pb, align=False):
    """RMSE of position differences; align=True removes the best rigid transform b -> a first."""
    pa, pb = np.asarray(pa, float), np.asarray(pb, float)
    if align:
        ca, cb = pa.mean(0), pb.mean(0)
        H = (pb - cb).T @ (pa - ca)
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(Vt.T @ U.T))])
        R = Vt.T @ D @ U.T
        pb = (pb - cb) @ R.T + ca
    d = pa - pb
    return float(np.sqrt((d * d).sum(1).mean()))


def ate_files(path_a, path_b, align=False, max_dt=0.01):
    ta, pa, _ = read_tum(path_a)
    tb, pb, _ = read_tum(path_b)
    m = associate(ta, tb, max_dt)
    if len(m) == 0:
        raise ValueError("no associated stamps")
    return ate(pa[m[:, 0]], pb[m[:, 1]], align), len(m)
