import os
import sys

import numpy as np
import pytest

try:                                   # torch bundles its own HIP runtime: it must initialise before libvslam_hip.so's does,
    import torch  # noqa: F401         # or a later torch.cuda call in the same process finds no device (bench.py imports it first too)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def synth_image(seed, w, h, lo=90, hi=130, nrect=None):
    """Seeded noise + rectangles image with plenty of FAST corners (test input, not a reference fixture)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(lo, hi, size=(h, w)).astype(np.int32)
    nrect = nrect or max(20, (w * h) // 400)
    for _ in range(nrect):
        x0, y0 = int(rng.integers(0, w - 8)), int(rng.integers(0, h - 8))
        ww, hh = int(rng.integers(4, 48)), int(rng.integers(4, 48))
        img[y0:y0 + hh, x0:x0 + ww] = int(rng.integers(0, 256))
    img += rng.integers(-6, 7, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding
