"""Oracle: SmallBlurryImage and the rotation prior (jni/SmallBlurryImage.cc, jni/Tracker.cc:86-105, 781-798, 885-893).

cv::resize and cv::GaussianBlur are third-party arithmetic restated in oracle/sbi.cpp (parity unpinned); these are
known-answer tests of the restatement."""
import numpy as np

from helpers import make_oracle, make_scene, pose_err
from oracle import binding as orc
from visualslam_android_amd import capi
from visualslam_android_amd.feeder import REF_CAM


def smooth_image(w=80, h=60, seed=5):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = 120 + 50 * np.sin(x / 7.0 + 0.3) * np.cos(y / 5.0) + 30 * np.sin((x + 2 * y) / 11.0)
    return np.clip(img + rng.normal(0, 1.0, img.shape), 0, 255).astype(np.uint8)


def test_make_from_keyframe_properties():
    l3 = smooth_image()
    small, tmpl = orc.sbi_make(l3)
    assert small.shape == (30, 40) and tmpl.shape == (30, 40) and tmpl.dtype == np.float32
    assert np.array_equal(small, orc.halfsample(l3))                       # the restated cv::resize
    assert abs(float(tmpl.mean())) < 0.05                                  # zero mean before the blur; the blur keeps it to rounding
    flat = np.full((60, 80), 77, np.uint8)
    assert np.abs(orc.sbi_make(flat)[1]).max() == 0.0                      # constant image -> zero template
    # the blur is a 9-tap separable Gaussian with sigma 0.75, replicated border: check one interior pixel by hand
    z = small.astype(np.float32) - np.float32(small.sum(dtype=np.uint32)) / np.float32(small.size)
    x = np.arange(9) - 4.0
    k = np.exp(-0.5 / 0.75 ** 2 * x * x).astype(np.float32)
    k = (k * (1.0 / float(k.sum(dtype=np.float64)))).astype(np.float32)
    want = float((np.outer(k, k).astype(np.float64) * z[6:15, 16:25].astype(np.float64)).sum())
    assert abs(float(tmpl[10, 20]) - want) < 1e-3


def test_rotation_prior_known_answers():
    l3 = smooth_image(seed=9)
    r, score = orc.sbi_rotation(l3, l3, REF_CAM)
    assert np.abs(r).max() < 1e-15 and score == 0.0                       # identical frames: no motion (project/unproject round trips are not the identity in the last bit)
    # the current frame shows the scene shifted by +2 level-3 pixels in x: with
    # this = last(x + t) the ESM translation is -t/2 small-image pixels and the equivalent camera rotation is about the y axis
    y, x = np.mgrid[0:60, 0:80]
    base = lambda xx: np.clip(120 + 50 * np.sin(xx / 7.0 + 0.3) * np.cos(y / 5.0) + 30 * np.sin((xx + 2 * y) / 11.0), 0, 255).astype(np.uint8)
    r, score = orc.sbi_rotation(base(x + 2.0), base(x + 0.0), REF_CAM)
    fx = 40 * REF_CAM[0]                                                    # focal length of the 40x30 camera in pixels
    assert abs(r[4] + (1.0 / fx)) < 0.25 / fx                                # one small-image pixel of shift ~ atan(1 / fx) about the y axis
    assert abs(r[3]) < 0.2 / fx and abs(r[5]) < 0.02 and np.abs(r[:3]).max() == 0.0


def test_tracker_with_rotation_prior_follows_ground_truth():
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=24, per_level=(120, 50, 20, 8))
    vp = capi.default_params(w, h, 1, use_sbi=1)
    o = make_oracle(vp, m, f.pose(-1))
    vp0 = capi.default_params(w, h, 1)
    o0 = make_oracle(vp0, m, f.pose(-1))
    worst, differs = 0.0, False
    for i in range(24):
        o.track_frame(frames[i]); o0.track_frame(frames[i])
        st = o.state()
        assert st.quality == 2, i
        worst = max(worst, pose_err(st.pose, f.pose(i)))
        differs |= pose_err(st.pose, o0.state().pose) > 0
    assert worst < 5e-3
    assert differs                                                          # the prior changes the predicted pose, hence the search
