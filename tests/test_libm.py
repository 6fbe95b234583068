"""csrc/vslam_libm.h: the transcendentals of the camera model and the Lie-group maps, one source for host and device.

CPU: accuracy against numpy (glibc) and the oracle using the very same functions.  GPU: the device returns the SAME BITS as the
host for every function over dense argument sets -- the premise of the bit-exact tracker parity tests."""
import numpy as np
import pytest

from visualslam_android_amd import capi

RANGES = {"sin": (-7.0, 7.0), "cos": (-7.0, 7.0), "tan": (-1.5, 1.5), "atan": (-30.0, 30.0), "asin": (-1.0, 1.0), "acos": (-1.0, 1.0),
          "sqrt": (0.0, 1e6), "rcp": (1e-3, 1e3)}
NP = {"sin": np.sin, "cos": np.cos, "tan": np.tan, "atan": np.arctan, "asin": np.arcsin, "acos": np.arccos, "sqrt": np.sqrt,
      "rcp": lambda v: 1.0 / v}


def arguments(name, n=200000, seed=3):
    rng = np.random.default_rng(seed)
    lo, hi = RANGES[name]
    x = rng.uniform(lo, hi, n)
    small = rng.uniform(-1.0, 1.0, n // 4) * 10.0 ** rng.uniform(-12, -1, n // 4)      # the Taylor / small-angle branches
    if name in ("sqrt", "rcp"):
        small = np.abs(small) + 1e-300
    edge = np.array([0.0, 0.4375, 0.6875, 1.1875, 2.4375, 0.5, 0.975, 1.0, -1.0, 0.3, 0.78125, np.pi / 4, np.pi / 2, 1e-9]) if name not in ("sqrt", "rcp") else np.array([1.0, 4.0])
    if name in ("asin", "acos"):
        edge = edge[np.abs(edge) <= 1.0]
    return np.concatenate([x, small, edge])


# outside the documented domains: the host and the device must still agree (NaN with the same bits), ADVICE r2
NON_FINITE = np.array([np.inf, -np.inf, np.nan, 1e300, -1e300, 2.0 ** 51, 2.0 ** 63, 1.5, -7.0])


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


@pytest.mark.parametrize("name", list(RANGES))
def test_host_accuracy_against_glibc(name):
    x = arguments(name)
    y = capi.eval_transcendental(name, x, on_host=True)
    ref = NP[name](x)
    bar = 4 if name == "tan" else (0 if name in ("sqrt", "rcp") else 2)       # glibc itself is within 1 ulp
    assert ulps(y, ref).max() <= bar, (name, int(ulps(y, ref).max()))


def test_oracle_uses_the_same_functions(oracle):
    cam5 = (0.841906, 1.10893, 0.505171, 0.470265, -0.0133843)
    rng = np.random.default_rng(5)
    for _ in range(200):
        cx, cy = rng.uniform(-0.6, 0.6, 2)
        im, _d, _inv, _lr = oracle.cam_project(cam5, 640, 480, cx, cy)
        r = np.hypot(cx, cy)
        rr = np.array([np.sqrt(cx * cx + cy * cy)])
        two_tan = 2.0 * capi.eval_transcendental("tan", [cam5[4] / 2.0], on_host=True)[0]
        fac = (1.0 / cam5[4]) * capi.eval_transcendental("atan", rr * two_tan, on_host=True)[0] / rr[0] if r >= 0.001 else 1.0
        assert im[0] == (640 * cam5[2] - 0.5) + (640 * cam5[0]) * (cx * fac)


def test_domain_errors_are_nan_on_the_host():
    for name in ("sin", "cos", "tan"):
        y = capi.eval_transcendental(name, NON_FINITE, on_host=True)
        assert np.isnan(y[:7]).all() and np.isfinite(y[7:]).all(), (name, y)
    for name in ("asin", "acos"):
        y = capi.eval_transcendental(name, NON_FINITE, on_host=True)
        assert np.isnan(y).all(), (name, y)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(RANGES))
def test_device_returns_the_host_bits(name):
    x = arguments(name, n=1000000)
    if name not in ("sqrt", "rcp"):
        x = np.concatenate([x, NON_FINITE])
    yh = capi.eval_transcendental(name, x, on_host=True)
    yd = capi.eval_transcendental(name, x, on_host=False)
    bad = yh.view(np.int64) != yd.view(np.int64)
    assert not bad.any(), (name, int(bad.sum()), x[bad][:4], yh[bad][:4], yd[bad][:4])
