"""GPU test: the C++ drop-in classes (include/vslam/ptam.h) driven like jni/jni_part.cpp's SystemPTAM."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_system_ptam_example_tracks():
    exe = os.path.join(ROOT, "examples", "_build", "system_ptam")
    assert os.path.exists(exe), "examples/_build/system_ptam not built (run __graft_entry__.build())"
    out = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) == 4
    for l in lines:
        assert "Tracking Map, quality good." in l, l
        err = float(re.search(r"max = ([0-9.e+-]+)", l).group(1))
        assert err < 2e-2, l
