"""CPU tests: oracle substrate (SE3, camera, M-estimators) -- known-answer tests and the verbatim-compiled reference."""
import json
import os

import numpy as np
import pytest

CAM = (0.841906, 1.10893, 0.505171, 0.470265, -0.0133843)   # jni/ATANCamera.cc:20-24
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def expm_se3(mu):
    """independent reference: series expm of the 4x4 twist matrix"""
    u, w = np.asarray(mu[:3]), np.asarray(mu[3:])
    A = np.zeros((4, 4))
    A[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    A[:3, 3] = u
    out, term = np.eye(4), np.eye(4)
    for k in range(1, 40):
        term = term @ A / k
        out = out + term
    return out


@pytest.mark.parametrize("mu", [[0.1, -0.2, 0.3, 0.4, -0.5, 0.6], [1e-6, 2e-6, -1e-6, 1e-5, 2e-5, -3e-5],
                                [0.3, 0.1, -0.2, 3e-4, -2e-4, 5e-4], [0, 0, 0, 0, 0, 0], [0.5, 0.5, 0.5, 2.0, -1.0, 0.7]])
def test_se3_exp_matches_matrix_exponential_and_ln_roundtrip(oracle, mu):
    T = oracle.se3_exp(mu)
    E = expm_se3(mu)
    assert np.allclose(T[:9].reshape(3, 3), E[:3, :3], atol=1e-12)
    assert np.allclose(T[9:], E[:3, 3], atol=1e-12)
    assert np.allclose(oracle.se3_ln(T), mu, atol=1e-9)           # K1: ln(exp(xi)) = xi


def test_se3_ln_near_pi_branch(oracle):
    for ang in (3.0, 3.14, 2.5):
        mu = np.array([0.1, 0.2, -0.3, ang / np.sqrt(3), ang / np.sqrt(3), ang / np.sqrt(3)])
        T = oracle.se3_exp(mu)
        T2 = oracle.se3_exp(oracle.se3_ln(T))                     # K1: exp(ln(T)) = T
        assert np.allclose(T, T2, atol=1e-9)


def test_camera_project_unproject_roundtrip_and_derivs(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        cx, cy = rng.uniform(-0.5, 0.5, 2)
        im, d, invalid, lr = oracle.cam_project(CAM, 640, 480, cx, cy)
        assert not invalid and lr > 0.5
        back = oracle.cam_unproject(CAM, 640, 480, im[0], im[1])
        assert np.allclose(back, [cx, cy], atol=1e-9)
        h = 1e-6                                                   # K2: finite-difference check of the 2x2 derivatives
        fx = (oracle.cam_project(CAM, 640, 480, cx + h, cy)[0] - oracle.cam_project(CAM, 640, 480, cx - h, cy)[0]) / (2 * h)
        fy = (oracle.cam_project(CAM, 640, 480, cx, cy + h)[0] - oracle.cam_project(CAM, 640, 480, cx, cy - h)[0]) / (2 * h)
        if cx * cx + cy * cy > 1e-4:
            assert np.allclose(d[:, 0], fx, rtol=1e-5, atol=1e-4) and np.allclose(d[:, 1], fy, rtol=1e-5, atol=1e-4)


def test_camera_int_radius_quirk(oracle):
    # quirk #5 (jni/ATANCamera.cc:70-82): int-typed operands make the largest radius 0
    _, _, _, lr_q = oracle.cam_project(CAM, 640, 480, 0.1, 0.1, quirks=1)
    _, _, invalid, lr = oracle.cam_project(CAM, 640, 480, 0.1, 0.1, quirks=0)
    assert lr_q == 0.0 and lr > 0.6 and not invalid


def test_mestimators_match_reference_header_golden(oracle):
    g = json.load(open(os.path.join(GOLD, "mestimator_ref.json")))   # written by oracle/gen_mestimator_golden.py
    for c in g["sigma_cases"]:
        for est in range(4):
            want = c["sigma_squared"][est]
            got = oracle.find_sigma_squared(est, c["v"])
            assert got == want or (np.isnan(got) and np.isnan(want)) or abs(got - want) <= 1e-15 * abs(want), (est, len(c["v"]))
    for c in g["point_cases"]:
        for est in range(4):
            for fn in ("weight", "sqrt_weight", "objective"):
                assert oracle.mest(fn, est, c["e2"], c["s2"]) == c[fn][est], (fn, est, c)


def test_mestimators_match_compiled_reference_live(oracle):
    ref = oracle.ref_mestimator()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    rng = np.random.default_rng(5)
    for n in (5, 64, 999):
        v = rng.standard_normal(n) ** 2
        for est in range(4):
            assert oracle.find_sigma_squared(est, v) == ref.ref_find_sigma_squared(est, np.ascontiguousarray(v).ctypes.data, n)
