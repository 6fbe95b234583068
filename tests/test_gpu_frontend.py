"""GPU parity: HIP front-end (through the C ABI) vs the oracle -- bit-exact images, corners, LUTs."""
import glob
import os

import numpy as np
import pytest

from conftest import synth_image
from visualslam_android_amd import capi

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def run_gpu(frames, **kw):
    S, h, w = frames.shape
    sys_ = capi.System(capi.default_params(w, h, S, **kw))
    sys_.make_keyframe_lite(frames)
    return sys_


@pytest.mark.parametrize("w,h,S", [(640, 480, 3), (160, 120, 2), (800, 480, 1), (1280, 720, 1), (200, 136, 2)])
def test_keyframe_lite_bit_exact(oracle, w, h, S):
    frames = np.stack([synth_image(100 + s, w, h) for s in range(S)])
    g = run_gpu(frames)
    for s in range(S):
        want = oracle.make_keyframe_lite(frames[s])
        for l in range(4):
            img, corners, lut = want[l]
            assert np.array_equal(g.read_level_image(s, l), img), (s, l)
            assert np.array_equal(g.read_corners(s, l), corners), (s, l)
            assert np.array_equal(g.read_row_lut(s, l), lut), (s, l)
    g.close()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "fast10_tree_*.npz"))))
def test_fast10_against_reference_tree_vectors(path):
    gv = np.load(path)
    img = gv["image"]
    thr = int(gv["threshold"])
    g = run_gpu(img[None], fast_threshold=[thr] * 4)
    assert np.array_equal(g.read_corners(0, 0), gv["corners"])
    g.close()


def test_edge_cases(oracle):
    flat = np.full((1, 120, 160), 77, np.uint8)
    g = run_gpu(flat)
    assert all(len(g.read_corners(0, l)) == 0 for l in range(4))
    assert np.all(g.read_row_lut(0, 0) == 0)
    g.close()
    # same system reused across frames: stale masks/counts must not leak
    a = synth_image(1, 160, 120)[None]
    g = run_gpu(a)
    n1 = len(g.read_corners(0, 0))
    g.make_keyframe_lite(flat)
    assert len(g.read_corners(0, 0)) == 0
    g.make_keyframe_lite(a)
    assert len(g.read_corners(0, 0)) == n1 > 0
    g.close()


def test_capacity_overflow_is_reported():
    noisy = np.random.default_rng(3).integers(0, 256, size=(1, 120, 160)).astype(np.uint8)
    g = run_gpu(noisy, max_corners=[64, 64, 64, 64])
    with pytest.raises(capi.VslamError):
        g.read_corners(0, 0)
    g.close()


@pytest.mark.parametrize("quirk", [0, capi.Q_NONMAX_RIGHT_NEIGHBOUR])
def test_fast_nonmax_bit_exact(oracle, quirk):
    frames = np.stack([synth_image(300 + s, 320, 240) for s in range(2)])
    g = run_gpu(frames, quirks=quirk)
    g.fast_nonmax()
    for s in range(2):
        want = oracle.make_keyframe_lite(frames[s])
        for l in range(4):
            img, corners, _ = want[l]
            sc = oracle.fast_score(img, corners, 10)
            keep = oracle.nonmax(corners, sc, quirk=bool(quirk))
            got, gsc = g.read_max_corners(s, l)
            assert np.array_equal(gsc[:len(corners)], sc), (s, l)
            assert np.array_equal(got, keep), (s, l)
    g.close()


def test_keyframe_rest_candidates_bit_exact(oracle):
    """vslam_make_keyframe_rest: fast_nonmax + Shi-Tomasi candidates (jni/KeyFrame.cc:53-95), lists and scores bit-exact."""
    frames = np.stack([synth_image(500 + s, 640, 480) for s in range(2)])
    g = run_gpu(frames)
    g.make_keyframe_rest(70.0)
    total = 0
    for s in range(2):
        want = oracle.make_keyframe_lite(frames[s])
        for l in range(4):
            img, corners, _ = want[l]
            keep = oracle.nonmax(corners, oracle.fast_score(img, corners, 10))
            pos, sc = oracle.candidates(img, keep, 70.0, 10)
            gpos, gsc = g.read_candidates(s, l)
            assert np.array_equal(gpos, pos), (s, l)
            assert np.array_equal(gsc, sc), (s, l)          # integer gradient sums, IEEE division and sqrt: bit-exact
            total += len(pos)
    assert total > 50
    g.close()


@pytest.mark.parametrize("on_device", [0, 1])
def test_padded_row_and_stream_strides(oracle, on_device):
    """cv::Mat::step need not equal the width (jni/KeyFrame.cc:12 copies whatever `step` the caller's Mat has), and the
    frames of a batch need not be adjacent: rows 13 bytes and images 1000 bytes further apart than tight, the gaps filled
    with noise that must never be read as pixels.  Host and device source buffers."""
    import torch
    w, h, S, pad, gap = 200, 136, 3, 13, 1000
    frames = np.stack([synth_image(7 + s, w, h) for s in range(S)])
    rng = np.random.default_rng(0)
    buf = rng.integers(0, 256, size=S * (h * (w + pad) + gap), dtype=np.uint8)
    sstride = h * (w + pad) + gap
    for s in range(S):
        view = buf[s * sstride: s * sstride + h * (w + pad)].reshape(h, w + pad)
        view[:, :w] = frames[s]
    g = capi.System(capi.default_params(w, h, S))
    if on_device:
        dev = torch.from_numpy(buf).cuda()
        capi._check(g.lib.vslam_make_keyframe_lite(g.h, dev.data_ptr(), w + pad, sstride, 1))
    else:
        capi._check(g.lib.vslam_make_keyframe_lite(g.h, buf.ctypes.data, w + pad, sstride, 0))
    g.synchronize()
    for s in range(S):
        want = oracle.make_keyframe_lite(frames[s])
        for l in range(4):
            img, corners, lut = want[l]
            assert np.array_equal(g.read_level_image(s, l), img), (s, l)
            assert np.array_equal(g.read_corners(s, l), corners), (s, l)
            assert np.array_equal(g.read_row_lut(s, l), lut), (s, l)
    g.close()


def test_fast_symmetries_at_full_size():
    """Size-independent properties of FAST-10 (jni/vision/cvfast.cpp:6088-9241) checked at 1280x720 without the oracle: the
    16-pixel ring and the brighter / darker tests are symmetric, so the level-0 corner set of the grey-inverted frame is the
    same, and that of the mirrored frame is the mirror image (the pyramid's +2 rounding breaks both on the coarser levels)."""
    w, h = 1280, 720
    img = synth_image(42, w, h)
    frames = np.stack([img, 255 - img, img[:, ::-1].copy(), img[::-1, :].copy()])
    g = run_gpu(frames)
    c = [g.read_corners(s, 0) for s in range(4)]
    xy = [set(zip((a & 0xFFFF).tolist(), (a >> 16).tolist())) for a in c]
    assert len(xy[0]) > 500
    assert xy[1] == xy[0]
    assert xy[2] == {(w - 1 - x, y) for (x, y) in xy[0]}
    assert xy[3] == {(x, h - 1 - y) for (x, y) in xy[0]}
    for a in c:                                                      # raster order, strictly increasing packed positions
        assert np.all(np.diff(a.astype(np.int64)) > 0)
    g.close()
