"""The host-side C++ of the repository under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle and
the synthetic feeder, driven by oracle/selftest_sanitize.cpp through a 46-frame sequence with keyframes, the asynchronous
map-maker model, map growth, the SmallBlurryImage prior, BundleAdjustAll and a stand-alone Bundle.  (The GPU code has no
sanitizer on this pool: GPU ASan / XNACK runs are refused.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_feeder_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"])
    out = subprocess.run([os.path.join(ROOT, "oracle", "_build", "asan", "selftest")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
    assert "46/46 frames good" in out.stdout and "stand-alone bundle" in out.stdout
