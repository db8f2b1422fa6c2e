"""Pins the oracle's small-math layer (oracle/smallmat.hpp, calcBodyCov, hash) against independent
numpy / LAPACK derivations.  The reference ships no golden vectors, so these derivations
are what anchors the restatement."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

import oracle_binding as ob


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def rodrigues(v):
    n = np.linalg.norm(v)
    K = skew(v / n)
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * K @ K


def test_eig_sym3_matches_lapack():
    rng = np.random.default_rng(0)
    for trial in range(300):
        A = rng.normal(size=(3, 3)) * 10 ** rng.uniform(-3, 1)
        C = A @ A.T
        if trial % 10 == 0:
            C = np.diag(rng.uniform(0.001, 1, 3))  # already diagonal
        if trial % 17 == 0:
            q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            C = q @ np.diag([1e-4, 0.5, 0.5 + 1e-9]) @ q.T  # nearly repeated pair
            C = 0.5 * (C + C.T)
        ev, V = ob.eig_sym3(C)
        w = np.linalg.eigvalsh(C)
        assert np.allclose(np.sort(ev), w, rtol=1e-12, atol=1e-15 * np.abs(w).max())
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-13)
        assert np.allclose(C @ V, V * ev, atol=1e-12 * np.abs(w).max())


def test_exp_thresholds_and_values():
    # math_utils.hpp:54-68 (1e-5) vs :19-32 (1e-7): between the two thresholds they differ
    v = np.array([3e-6, 0, 0])
    e3, ev, _ = ob.exp_log(v)
    assert np.array_equal(e3, np.eye(3)) and not np.array_equal(ev, np.eye(3))
    e3, ev, _ = ob.exp_log(np.array([3e-8, 0, 0]))
    assert np.array_equal(e3, np.eye(3)) and np.array_equal(ev, np.eye(3))
    rng = np.random.default_rng(1)
    for _ in range(100):
        v = rng.normal(size=3) * rng.uniform(1e-4, 2.5)
        e3, ev, lg = ob.exp_log(v)
        assert np.allclose(e3, rodrigues(v), atol=1e-14) and np.allclose(ev, e3, atol=1e-15)
        assert np.allclose(e3 @ e3.T, np.eye(3), atol=1e-14)
        if np.linalg.norm(v) < 3.0:
            assert np.allclose(lg, v, rtol=1e-6, atol=1e-9)  # Log is the reference's own (0.001 switch, math_utils.hpp:71-76)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.floats(-1.0, 1.0), min_size=3, max_size=3))
def test_log_exp_roundtrip_property(v):
    v = np.array(v)
    e3, _, lg = ob.exp_log(v)
    if 2e-3 < np.linalg.norm(v) < 3.0:
        assert np.allclose(lg, v, rtol=1e-7, atol=1e-10)


def body_cov_numpy(pb, range_inc, degree_inc):
    """voxel_map.cc:22-40 re-derived with numpy (float32 where the reference uses float)."""
    pb = np.array(pb, dtype=np.float64)
    if pb[2] == 0:
        pb[2] = 0.0001
    rng_ = np.float32(np.sqrt(pb @ pb))
    range_var = np.float32(range_inc) * np.float32(range_inc)
    dvar = np.sin(float(np.float32(degree_inc)) * 0.017453293) ** 2
    d = pb / np.sqrt(pb @ pb)
    dh = skew(d)
    b1 = np.array([1.0, 1.0, -(d[0] + d[1]) / d[2]])
    b1 /= np.linalg.norm(b1)
    b2 = np.cross(b1, d)
    b2 /= np.linalg.norm(b2)
    N = np.stack([b1, b2], axis=1)
    A = float(rng_) * dh @ N
    return np.outer(d, d) * float(range_var) + A @ (dvar * np.eye(2)) @ A.T


def test_calc_body_cov_matches_numpy():
    rng = np.random.default_rng(2)
    for _ in range(200):
        pb = rng.normal(size=3) * rng.uniform(0.5, 40)
        c = ob.calc_body_cov(pb, 0.04, 0.2)
        ref = body_cov_numpy(pb, 0.04, 0.2)
        assert np.allclose(c, ref, rtol=1e-10, atol=1e-16), np.abs(c - ref).max()
        w = np.linalg.eigvalsh(0.5 * (c + c.T))
        assert w.min() > 0
        # largest-variance direction is tangential for long ranges, radial (dept_err) for short ones
    c0 = ob.calc_body_cov(np.array([1.0, 2.0, 0.0]), 0.04, 0.2)  # z == 0 -> 0.0001 (voxel_map.cc:23)
    assert np.allclose(c0, body_cov_numpy([1.0, 2.0, 0.0001], 0.04, 0.2), rtol=1e-10)


def c_int32(v):
    return (v + 2 ** 31) % 2 ** 32 - 2 ** 31


def hash_ref(x, y, z):
    """eigen_types.hpp:79-82 in wrapped int32 arithmetic, C '%' (truncating), cast to size_t."""
    h = c_int32(c_int32(x * 73856093) ^ c_int32(y * 471943) ^ c_int32(z * 83492791))
    r = int(np.fmod(h, 10000000))
    return r % 2 ** 64


def test_hash_vec3_semantics():
    rng = np.random.default_rng(3)
    for _ in range(500):
        x, y, z = (int(v) for v in rng.integers(-5000, 5000, 3))
        assert ob.hash_vec3(x, y, z) == hash_ref(x, y, z)
    assert ob.hash_vec3(0, 0, 0) == 0
    assert ob.hash_vec3(-1, 0, 0) == hash_ref(-1, 0, 0) and hash_ref(-1, 0, 0) > 2 ** 63  # negative int -> huge size_t
