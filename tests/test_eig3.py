"""The closed-form 3 x 3 symmetric eigen-solver of the device plane fits (leg-kilo_amd/csrc/lk_eig3.h, compiled for the host from the
same header by tools/probes/eig3_host.cc) against LAPACK: covariances of noisy planar point sets at map coordinates (the input of
init_plane, voxel_map.cc:42-56), near-degenerate pairs, diagonal and rank-deficient matrices."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eig3(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("eig3") / "eig3_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", so, os.path.join(ROOT, "tools", "probes", "eig3_host.cc")])
    L = ctypes.CDLL(so)

    def run(A6):
        A6 = np.ascontiguousarray(A6, dtype=np.float64).reshape(-1, 6)
        n = len(A6)
        ev, V = np.zeros((n, 3)), np.zeros((n, 9))
        L.lk_eig_sym3_host_n(A6.ctypes.data_as(ctypes.c_void_p), ev.ctypes.data_as(ctypes.c_void_p), V.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n))
        return ev, V.reshape(n, 3, 3)

    return run


def full(a):
    return np.array([[a[0], a[1], a[2]], [a[1], a[3], a[4]], [a[2], a[4], a[5]]])


def test_plane_covariances_against_lapack(eig3):
    rng = np.random.default_rng(1)
    cases = []
    for i in range(4000):
        npts = rng.integers(6, 60)
        Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        sc = np.array([rng.uniform(1e-3, 0.05), rng.uniform(0.02, 0.3), rng.uniform(0.02, 0.3)])
        if i % 10 == 0:
            sc[1] = sc[2] * (1 + 10.0 ** rng.uniform(-12, -2))   # the two in-plane eigenvalues nearly equal
        if i % 17 == 0:
            sc[0] = sc[1] * (1 + 10.0 ** rng.uniform(-9, -2))    # the two small eigenvalues nearly equal
        P = (rng.normal(size=(npts, 3)) * sc) @ Q.T + rng.uniform(-30, 30, size=3)
        c = P.mean(0)
        C = (P.T @ P) / npts - np.outer(c, c)
        cases.append([C[0, 0], C[0, 1], C[0, 2], C[1, 1], C[1, 2], C[2, 2]])
    ev, V = eig3(np.array(cases))
    for a, e, v in zip(cases, ev, V):
        A = full(a)
        w, U = np.linalg.eigh(A)
        s = np.abs(w).max()
        assert np.abs(np.sort(e) - w).max() <= 1e-14 * s          # eigenvalues: ulps of the largest one
        assert np.abs(A @ v - v * e).max() <= 1e-11 * s           # eigen-pairs
        assert np.abs(v.T @ v - np.eye(3)).max() <= 1e-14         # orthonormal basis even where eigenvalues coincide
        if (w[1] - w[0]) > 1e-3 * w[2]:                           # the plane normal where it is well defined
            d = min(np.abs(U[:, 0] - v[:, int(np.argmin(e))]).max(), np.abs(U[:, 0] + v[:, int(np.argmin(e))]).max())
            assert d <= 1e-9


def test_special_matrices(eig3):
    for a in ([1, 0, 0, 1, 0, 1], [0, 0, 0, 0, 0, 0], [2, 0, 0, 1, 0, 3], [1, 1e-300, 0, 1, 0, 1], [1, 1, 1, 1, 1, 1], [5, 0, 0, 5, 1e-9, 5],
              [1e-12, 0, 0, 3e-12, 1e-12, 2e-12], [4e6, 1e6, 0, 2e6, 0, 1e6]):
        e, v = eig3(np.array([a], dtype=float))
        A = full(a)
        w = np.linalg.eigvalsh(A)
        s = max(np.abs(w).max(), 1e-300)
        assert np.abs(np.sort(e[0]) - w).max() <= 1e-14 * s, a
        assert np.abs(A @ v[0] - v[0] * e[0]).max() <= 1e-13 * s, a
        assert np.abs(v[0].T @ v[0] - np.eye(3)).max() <= 1e-14, a
