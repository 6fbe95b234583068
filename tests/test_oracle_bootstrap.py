"""Map bootstrap in the oracle: HomographyInit, CalcPlaneAligner, the trails and InitFromStereo.

PARITY UNPINNED: the reference holds no fixture for any of this (and draws from rand()); the checks are against the ground truth
of synthetic planar scenes."""
import numpy as np

import oracle.binding as orc
from visualslam_android_amd import capi, feeder


def _rot(ax, a):
    c, s = np.cos(a), np.sin(a)
    return {0: np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 1: np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]), 2: np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def _planar_matches(seed, n=300, outliers=30, noise=1e-4):
    rng = np.random.default_rng(seed)
    P = np.c_[rng.uniform(-1, 1, n), rng.uniform(-0.8, 0.8, n), np.zeros(n)]
    P[:, 2] = 2.0 + 0.2 * P[:, 0] - 0.1 * P[:, 1]                      # a tilted plane in front of the first camera
    R = _rot(1, 0.05) @ _rot(0, -0.03) @ _rot(2, 0.02)
    t = np.array([0.2, 0.01, -0.02])
    Q = (R @ P.T).T + t
    first, second = P[:, :2] / P[:, 2:], Q[:, :2] / Q[:, 2:]
    second = second + rng.normal(0, noise, second.shape)
    second[:outliers] += rng.uniform(-0.1, 0.1, (outliers, 2))
    return np.c_[first, second, np.tile([500.0, 0, 0, 500.0], (n, 1))], R, t


def test_homography_init_recovers_rotation_and_translation_direction():
    """HomographyInit::Compute (jni/HomographyInit.cc:43-71): MLESAC over 300 minimal sets, refinement, Faugeras-Lustman
    decomposition, choice by visibility -- against the motion that generated the matches, with 10 % gross outliers."""
    for seed in (3, 4, 5):
        m8, R, t = _planar_matches(seed)
        ok, T, ninl = orc.homography_init(m8, 5.0, seed)
        assert ok and 260 <= ninl <= 275, (seed, ninl)
        Re, te = T[:9].reshape(3, 3), T[9:]
        assert np.abs(Re - R).max() < 2e-3, seed
        assert np.abs(te / np.linalg.norm(te) - t / np.linalg.norm(t)).max() < 2e-2, seed
        assert abs(np.linalg.det(Re) - 1.0) < 1e-9 and np.abs(Re @ Re.T - np.eye(3)).max() < 1e-9
    ok2, T2, _ = orc.homography_init(m8, 5.0, 5)
    assert ok2 and np.array_equal(T, T2)                                # reproducible for a seed
    assert not orc.homography_init(m8[:3], 5.0, 1)[0]                   # fewer than four matches
    few, Rf, tf = _planar_matches(9, n=8, outliers=0, noise=0.0)        # fewer than ten: all matches in one fit (:226-229)
    okf, Tf, _ = orc.homography_init(few, 5.0, 1)
    assert okf and np.abs(Tf[:9].reshape(3, 3) - Rf).max() < 1e-6


def test_plane_aligner_puts_the_dominant_plane_at_z_zero():
    """MapMaker::CalcPlaneAligner (jni/MapMaker.cc:1104-1231): 100 three-point hypotheses, inliers within 0.05, the direction of
    least variance as the normal (towards the camera), x axis kept as close to the old one as the plane allows."""
    rng = np.random.default_rng(1)
    pts = np.c_[rng.uniform(-1, 1, 500), rng.uniform(-1, 1, 500), rng.normal(0, 0.002, 500)]
    pts[:25, 2] += rng.uniform(0.2, 0.6, 25)                            # clutter off the plane
    Rw, tw = _rot(0, 0.4) @ _rot(1, -0.3), np.array([0.3, -0.2, 1.5])
    W = (Rw @ pts.T).T + tw
    ok, A = orc.calc_plane_aligner(W, 1)
    assert ok
    Ra, ta = A[:9].reshape(3, 3), A[9:]
    Z = (Ra @ W.T).T + ta
    assert np.abs(Z[25:, 2]).max() < 0.01 and abs(np.median(Z[25:, 2])) < 1e-3
    assert np.abs(Ra @ Ra.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Ra) - 1) < 1e-12
    assert Ra[2, 2] < 0                                                  # the normal (third row) points back towards the camera at the origin
    assert not orc.calc_plane_aligner(W[:9], 1)[0]                       # fewer than ten points: identity (:1107-1110)


def test_trails_and_init_from_stereo_make_a_trackable_map():
    """Tracker::TrackForInitialMap (jni/Tracker.cc:247-288): the first spacebar starts the trails (MiniPatch forward / backward
    matching), the second runs MapMaker::InitFromStereo; the map it leaves -- plane at z = 0, second camera wiggle_scale from the
    first -- is then tracked with good quality and grows keyframes by itself."""
    w, h = 640, 480
    f = feeder.Feeder(w, h, seed=1234, noise=2)
    frames = f.render(0, 30)
    vp = capi.default_params(w, h, 1, patch_size=8, grow_map=3)
    o = orc.OracleSystem(orc.params_from_vslam(vp))
    for t in range(30):
        if t in (0, 12):
            o.press_spacebar()
        o.track_frame(frames[t])
        info = o.init_info()
        if t == 0:
            assert info["stage"] == 1 and info["trails"] > 100
            tr0 = o.trails()
            assert np.array_equal(tr0[:, :2], tr0[:, 2:])                 # irCurrentPos = irInitialPos
        elif t < 12:
            assert info["stage"] == 1 and 100 < info["trails"] <= len(tr0) and not info["map_good"]
        else:
            assert info["stage"] == 2 and info["init_ok"] and info["map_good"]
        if t == 11:
            tr = o.trails()
            d = np.linalg.norm((tr[:, 2:] - tr[:, :2]).astype(float), axis=1)
            assert 2.0 < np.median(d) < 40.0                                # the feeder's sideways motion, in pixels
    assert info["hom_inliers"] > 100 and info["stereo_points"] > 100
    s = o.state()
    assert s.n_keyframes >= 3 and s.n_points > info["stereo_points"] and s.quality == 2 and sum(s.found) > 150
    k0, k1 = np.array(o.keyframe_pose(0)[:]), np.array(o.keyframe_pose(1)[:])
    c0 = -k0[:9].reshape(3, 3).T @ k0[9:]; c1 = -k1[:9].reshape(3, 3).T @ k1[9:]
    assert abs(np.linalg.norm(c1 - c0) - vp.wiggle_scale) < 0.02            # the scale of the map (jni/MapMaker.cc:250), up to the adjustments
    pts = o.points()
    z = pts["pos"][pts["bad"] == 0][:, 2]
    assert abs(np.median(z)) < 0.01 and np.percentile(np.abs(z), 90) < 0.08 * abs(c0[2])   # the plane sits at z = 0
    assert c0[2] > 0                                                          # the cameras are on the side the plane normal (+z) points to
