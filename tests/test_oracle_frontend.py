"""CPU tests: the oracle's front-end restatement against the golden vectors and brute force."""
import glob
import json
import os

import numpy as np
import pytest

from conftest import synth_image

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_fast10_tree_pin_is_exhaustive():
    # written by oracle/pin_fast_tree.py from a tree-walk of jni/vision/cvfast.cpp:6123-9236
    pin = json.load(open(os.path.join(GOLD, "fast10_tree_pin.json")))
    assert pin["states"] == 3 ** 16 and pin["mismatches_vs_run_of_10"] == 0


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "fast10_tree_*.npz"))))
def test_fast10_matches_reference_tree_vectors(oracle, path):
    g = np.load(path)
    got = oracle.fast10(g["image"], int(g["threshold"]))
    assert np.array_equal(got, g["corners"])


def brute_fast10(img, thr):
    h, w = img.shape
    out = []
    im = img.astype(np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            c = im[y, x]
            ring = [im[y + dy, x + dx] for dx, dy in RING]
            for sign in (1, -1):
                flags = [(v > c + thr) if sign > 0 else (v < c - thr) for v in ring]
                f2 = flags + flags
                run = best = 0
                for f in f2:
                    run = run + 1 if f else 0
                    best = max(best, run)
                if best >= 10:
                    out.append(x | (y << 16))
                    break
    return np.array(out, np.uint32)


def test_fast10_bruteforce_small(oracle):
    img = synth_image(5, 48, 40)
    assert np.array_equal(oracle.fast10(img, 10), brute_fast10(img, 10))


def test_fast10_edge_cases(oracle):
    assert len(oracle.fast10(np.full((32, 32), 77, np.uint8), 10)) == 0          # flat image: no corners
    assert len(oracle.fast10(np.zeros((6, 6), np.uint8), 10)) == 0               # smaller than the border
    img = np.zeros((9, 9), np.uint8); img[4, 4] = 255                            # isolated bright pixel
    assert np.array_equal(oracle.fast10(img, 10), np.array([4 | (4 << 16)], np.uint32))
    sat = np.full((16, 16), 250, np.uint8); sat[8, 8] = 255                      # c+t > 255 can never be exceeded
    assert len(oracle.fast10(sat, 10)) == 0


def test_halfsample_and_lut(oracle):
    img = synth_image(7, 64, 48)
    half = oracle.halfsample(img)
    i = img.astype(np.int32)
    want = (i[0::2, 0::2] + i[0::2, 1::2] + i[1::2, 0::2] + i[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, want.astype(np.uint8))
    c = oracle.fast10(img, 10)
    lut = oracle.row_lut(c, 48)
    ys = (c >> 16).astype(np.int64)
    for y in range(48):
        assert lut[y] == np.searchsorted(ys, y, side="left")   # first index with corner.y >= y


def test_nonmax_semantics(oracle):
    img = synth_image(9, 96, 80)
    c = oracle.fast10(img, 10)
    sc = oracle.fast_score(img, c, 10)
    keep = oracle.nonmax(c, sc, quirk=False)
    # intended semantics: survive unless an 8-neighbour corner has a strictly greater score
    pos = {int(v): int(s) for v, s in zip(c, sc)}
    want = []
    for v, s in zip(c, sc):
        x, y = int(v) & 0xFFFF, int(v) >> 16
        ok = True
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if dx == 0 and dy == 0:
                    continue
                k = (x + dx) | ((y + dy) << 16)
                if k in pos and pos[k] > s:
                    ok = False
        if ok:
            want.append(int(v))
    assert np.array_equal(keep, np.array(want, np.uint32))
    # quirk #7 mode differs only through the right-neighbour test and never reads out of bounds
    kq = oracle.nonmax(c, sc, quirk=True)
    assert 0 < len(kq) <= len(c)
    assert len(oracle.nonmax(c[:0], sc[:0])) == 0


def test_make_keyframe_lite_levels(oracle):
    img = synth_image(21, 160, 120)
    lv = oracle.make_keyframe_lite(img)
    assert [l[0].shape for l in lv] == [(120, 160), (60, 80), (30, 40), (15, 20)]
    assert np.array_equal(lv[1][0], oracle.halfsample(img))
    for l, thr in enumerate((10, 15, 15, 10)):
        assert np.array_equal(lv[l][1], oracle.fast10(lv[l][0], thr))


def test_shi_tomasi_known_answer(oracle):
    img = np.zeros((32, 32), np.uint8)
    assert oracle.shi_tomasi(img, 3, 16, 16) == 0.0
    img[:, 16:] = 200                                    # vertical edge: one zero eigenvalue
    assert abs(oracle.shi_tomasi(img, 3, 16, 16)) < 1e-9
    img[16:, :] = 200                                    # corner: both eigenvalues > 0
    assert oracle.shi_tomasi(img, 3, 16, 16) > 70


def test_candidates_and_thinning_semantics(oracle):
    """KeyFrame::MakeKeyFrame_Rest candidate loop (jni/KeyFrame.cc:66-95) + MapMaker::ThinCandidates (jni/MapMaker.cc:393-422)."""
    img = np.zeros((64, 64), np.uint8)
    img[20:, 20:] = 200                                  # one strong corner at (20, 20)
    pack = lambda x, y: np.uint32(x | (y << 16))
    mc = np.array([pack(5, 5), pack(20, 20), pack(40, 20), pack(58, 40)], np.uint32)   # border, corner, edge, border
    pos, sc = oracle.candidates(img, mc, 70.0, 10)
    assert list(pos) == [pack(20, 20)] and sc[0] == oracle.shi_tomasi(img, 3, 20, 20) and sc[0] > 70
    # strict threshold: a candidate at exactly the minimum score is dropped (:81 uses >)
    assert len(oracle.candidates(img, mc, sc[0], 10)[0]) == 0
    # thinning on level 1: a measurement at level 1 or 2 within 10 level-pixels kills the candidate, others do not
    cand = np.array([pack(20, 20), pack(50, 50)], np.uint32); csc = np.array([80.0, 90.0])
    root = np.array([[2 * 20 + 12.0, 2 * 20.0], [200.0, 200.0], [2 * 20.0, 2 * 20.0]])   # L0 coordinates
    for lev, want in (([1, 1, 0], [pack(20, 20), pack(50, 50)][1:] ), ([0, 3, 0], [pack(20, 20), pack(50, 50)])):
        got, gs = oracle.thin_candidates(cand, csc, 1, root, np.array(lev, np.int32))
        assert list(got) == list(want)
    # distance exactly 10 survives (< 100 is the kill test), 9.x does not; positions are rounded() half away from zero
    got, _ = oracle.thin_candidates(cand[:1], csc[:1], 0, np.array([[30.0, 20.0]]), np.array([0], np.int32))
    assert len(got) == 1
    got, _ = oracle.thin_candidates(cand[:1], csc[:1], 0, np.array([[29.49, 20.0]]), np.array([1], np.int32))
    assert len(got) == 0
    got, _ = oracle.thin_candidates(cand[:1], csc[:1], 0, np.array([[29.5, 20.0]]), np.array([1], np.int32))
    assert len(got) == 1
