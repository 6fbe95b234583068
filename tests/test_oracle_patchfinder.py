"""CPU tests: oracle PatchFinder pieces -- known-answer tests K3/K4 of SURVEY.md 8(c)."""
import numpy as np

from conftest import synth_image


def smooth_image(w, h, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = 128 + 60 * np.sin(xx / 7.0 + rng.uniform(0, 3)) * np.cos(yy / 9.0) + 40 * np.sin((xx + yy) / 13.0)
    return img


def test_zmssd_identities(oracle):
    img = synth_image(3, 64, 48)
    for P in (8, 11):
        x, y = 30, 20
        b = P // 2
        t = img[y - b:y - b + P, x - b:x - b + P].copy()
        assert oracle.zmssd(t, img, x, y) == 0                                  # ZMSSD(T, T) = 0
        t2 = np.clip(t.astype(int) // 2 + 7, 0, 255).astype(np.uint8)
        im2 = np.clip(img.astype(int) // 2 + 40, 0, 255).astype(np.uint8)       # constant offset cancels (zero mean)
        n = P * P
        a = t2.astype(np.int64).ravel(); bimg = im2[y - b:y - b + P, x - b:x - b + P].astype(np.int64).ravel()
        sa, sb = a.sum(), bimg.sum()
        num = 2 * sa * sb - sa * sa - sb * sb
        want = int(np.trunc(num / n)) + int((bimg ** 2).sum()) + int((a ** 2).sum()) - 2 * int((a * bimg).sum())   # C++ truncating division
        assert oracle.zmssd(t2, im2, x, y) == want
        assert oracle.zmssd(t, img, 1, 1) == 500 * n + 1                        # border -> mnMaxSSD + 1


def test_transform_image_identity_and_outside_count(oracle):
    img = synth_image(4, 64, 48)
    for P in (8, 11):
        M = [1.0, 0.0, 0.0, 1.0]
        out, n = oracle.transform_image(img, P, M, [30.0, 20.0], [P // 2, P // 2])
        b = P // 2
        assert n == 0 and np.array_equal(out, img[20 - b:20 - b + P, 30 - b:30 - b + P])
        out, n = oracle.transform_image(img, P, M, [1.0, 1.0], [P // 2, P // 2])   # hangs off the top-left corner
        assert n > 0
        out, n = oracle.transform_image(img, P, [2.0, 0.0, 0.0, 2.0], [30.0, 20.0], [P // 2, P // 2])
        assert n == 0 and out[b, b] == img[20, 30]                                   # centre pixel maps to the centre


def test_subpixel_recovers_known_shift(oracle):
    # K4: inverse-compositional refinement recovers a +-0.3 px shift on a smooth patch
    P = 11
    for dx, dy in [(0.3, -0.2), (-0.25, 0.3), (0.0, 0.0)]:
        base = smooth_image(64, 48, 1)
        yy, xx = np.mgrid[0:48, 0:64].astype(np.float64)
        rng = np.random.default_rng(1)
        ph = rng.uniform(0, 3)
        shifted = 128 + 60 * np.sin((xx - dx) / 7.0 + ph) * np.cos((yy - dy) / 9.0) + 40 * np.sin((xx - dx + yy - dy) / 13.0)
        img = np.clip(np.round(shifted), 0, 255).astype(np.uint8)
        ref = np.clip(np.round(base), 0, 255).astype(np.uint8)
        b = P // 2
        tmpl = ref[24 - b:24 - b + P, 32 - b:32 - b + P]
        ok, pos = oracle.subpix_refine(tmpl, img, 32, 24, 10)
        assert ok
        assert abs(pos[0] - (32 + dx)) < 0.08 and abs(pos[1] - (24 + dy)) < 0.08
