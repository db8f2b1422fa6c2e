"""TUM writer / ATE tool: format of trajectory_saver.hpp:43-50 and the ATE definition."""
import numpy as np

from legkilo_amd import synth, tum


def test_quaternion_convention_and_line_format(tmp_path):
    rng = np.random.default_rng(0)
    tr = synth.Trajectory()
    ts = np.linspace(1.0, 3.0, 21)
    R, p = tr.rot(ts), tr.pos(ts)
    f = tmp_path / "traj.txt"
    tum.write_tum(f, ts, R, p)
    lines = open(f).read().splitlines()
    assert len(lines) == 21
    tok = lines[3].split(" ")
    assert len(tok) == 8 and all(len(t.split(".")[1]) == 9 for t in tok)  # std::fixed << setprecision(9)
    t2, p2, q2 = tum.read_tum(f)
    assert np.allclose(t2, ts, atol=1e-9) and np.allclose(p2, p, atol=1e-9)
    for Ri, q in zip(R, q2):  # quaternion (x y z w) reproduces the rotation, unit norm, Eigen's branch for w
        x, y, z, w = q
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(Rq, Ri, atol=1e-8) and abs(np.linalg.norm(q) - 1) < 1e-8
    # trace <= 0 branch (180 degree turns)
    for ax in range(3):
        Rm = -np.eye(3)
        Rm[ax, ax] = 1.0
        q = np.array(tum.rot_to_quat(Rm))
        assert abs(abs(q[ax]) - 1) < 1e-12 and abs(q[3]) < 1e-12


def test_ate_definition(tmp_path):
    rng = np.random.default_rng(1)
    tr = synth.Trajectory()
    ts = np.arange(0, 10, 0.1)
    p = tr.pos(ts)
    assert tum.ate(p, p) == 0.0
    off = p + np.array([0.001, 0.0, 0.0])
    assert abs(tum.ate(p, off) - 0.001) < 1e-12           # 1 mm offset = 1 mm ATE (the north-star threshold)
    # a rigidly moved copy has zero ATE after alignment
    th = 0.3
    Rz = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    moved = p @ Rz.T + [1.0, -2.0, 0.5]
    assert tum.ate(p, moved) > 0.5 and tum.ate(p, moved, align=True) < 1e-9
    fa, fb = tmp_path / "a.txt", tmp_path / "b.txt"
    R = tr.rot(ts)
    tum.write_tum(fa, ts, R, p)
    tum.write_tum(fb, ts[::2] + 0.002, R[::2], off[::2])
    e, n = tum.ate_files(fa, fb)
    assert n == 50 and abs(e - 0.001) < 1e-8
