"""CPU test: the oracle's TrackFrame + AddKeyFrame + BundleAdjustRecent on a small synthetic sequence
(BASELINE.json configs[0]: plumbing of the reference CPU path, no GPU)."""
import os

import numpy as np

from helpers import make_oracle, make_scene, pose_err
from visualslam_android_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tracks_ground_truth_and_adds_keyframes():
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=26, per_level=(120, 50, 20, 8))
    vp = capi.default_params(w, h, 1)
    o = make_oracle(vp, m, f.pose(-1))
    n_kf0 = len(m["keyframes"])
    worst = 0.0
    for i in range(26):
        o.track_frame(frames[i])
        st = o.state()
        assert st.quality == 2, (i, list(st.attempted), list(st.found))
        worst = max(worst, pose_err(st.pose, f.pose(i)))
        assert sum(st.found) > 0.5 * sum(st.attempted) > 50
    assert worst < 5e-3                       # pixel-quantised measurements: millimetre-level pose error
    st = o.state()
    assert st.n_keyframes >= n_kf0 + 2        # frame 0 and frame 21 (jni/Tracker.cc:128: > 20 frames apart)
    assert st.ba_accepted >= 0 and st.n_ba_trials > 0
    km = o.keyframe_meas(st.n_keyframes - 1)
    assert len(km["pt"]) > 100 and set(km["source"]) == {0}


def test_quirk_cam_int_radius_disables_tracking():
    # quirk #5: LargestRadiusInImage == 0 -> TrackerData::Project rejects every point -> nothing is searched
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=78, n_frames=2, per_level=(60, 20, 8, 4))
    o = make_oracle(capi.default_params(w, h, 1, quirks=capi.Q_CAM_INT_RADIUS), m, f.pose(-1))
    o.track_frame(frames[0])
    st = o.state()
    assert sum(st.attempted) == 0 and st.quality == 0


def test_reproject_point_recovers_a_known_point():
    """MapMaker::ReprojectPoint (jni/MapMaker.cc:174-200) on exact projections; Eigen::JacobiSVD of the 4x4 A is restated as
    the two-sided Jacobi SVD of A itself (parity unpinned: the reference's Eigen is not vendored)."""
    from oracle import binding as orc
    rng = np.random.default_rng(3)
    for _ in range(20):
        w = rng.normal(0, 0.2, 3)
        T = orc.se3_exp(np.concatenate([rng.normal(0, 0.3, 3), w]))          # A from B
        R, t = np.array(T[:9]).reshape(3, 3), np.array(T[9:])
        XB = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(1.0, 3.0)])
        XA = R @ XB + t
        got = orc.reproject_point(T, XA[:2] / XA[2], XB[:2] / XB[2])
        assert np.abs(got - XB).max() < 1e-9


def test_reproject_point_small_baseline_and_numpy_svd():
    """Known answers for the restated JacobiSVD: (1) low-parallax pairs (baseline 1e-3 .. 1e-5 of the depth), where an
    eigen-solve of A^T A would square the condition number, still return the point to the accuracy the geometry allows;
    (2) on random and on triangulation matrices the chosen vector is numpy's last right singular vector (LAPACK) up to sign."""
    from oracle import binding as orc
    rng = np.random.default_rng(11)
    for base in (1e-3, 1e-4, 1e-5):
        for _ in range(10):
            T = orc.se3_exp(np.concatenate([rng.normal(0, 1.0, 3) * base, rng.normal(0, 0.3, 3) * base]))
            R, t = np.array(T[:9]).reshape(3, 3), np.array(T[9:])
            XB = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(1.0, 3.0)])
            XA = R @ XB + t
            vA, vB = XA[:2] / XA[2], XB[:2] / XB[2]
            got = orc.reproject_point(T, vA, vB)
            A = np.zeros((4, 4)); PD = np.hstack([R, t[:, None]])
            A[0] = [-1, 0, vB[0], 0]; A[1] = [0, -1, vB[1], 0]; A[2] = vA[0] * PD[2] - PD[0]; A[3] = vA[1] * PD[2] - PD[1]
            v = np.linalg.svd(A)[2][-1]
            want = v[:3] / v[3]
            assert np.abs(got - want).max() < 1e-6 * max(1.0, 1e-4 / base), (base, got, want)    # the two SVDs agree far inside the geometric error
            assert np.abs(got - XB).max() < 2e-11 / base ** 1.0 * 10, (base, got, XB)            # error ~ eps * depth / parallax


def test_map_growth_adds_points_on_the_scene_plane():
    """grow_map = 1 (AddSomeMapPoints, jni/MapMaker.cc:424-437, 525-703): new points come from candidates away from the
    existing measurements, triangulate close to the feeder's z = 0 plane, and are tracked afterwards."""
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=46, per_level=(120, 50, 20, 8))
    n0 = len(m["points"])
    o = make_oracle(capi.default_params(w, h, 1, grow_map=1), m, f.pose(-1))
    worst = 0.0
    for i in range(46):
        o.track_frame(frames[i])
        st = o.state()
        assert st.quality == 2
        worst = max(worst, pose_err(st.pose, f.pose(i)))
    assert worst < 6e-3
    P = o.points()
    new = P["pos"][n0:]
    assert len(new) > 30
    assert np.median(np.abs(new[:, 2])) < 0.05                 # scene depth is about 1: a few per cent from a short baseline
    assert P["n_in"][n0:].sum() > 3 * len(new) > P["n_out"][n0:].sum()   # the tracker uses them as inliers
    kf_new = o.keyframe_meas(o.state().n_keyframes - 1)
    assert (kf_new["source"] == 2).sum() > 0                   # SRC_ROOT measurements of the points this keyframe created


def test_refind_in_single_keyframe_adds_measurements():
    """grow_map bit 1 (ReFindInSingleKeyFrame / ReFind_Common, jni/MapMaker.cc:967-1056): the new keyframe gains SRC_REFIND
    measurements of points the tracker did not measure this frame (its 1000-point budget, PVS sampling), close to where the
    keyframe's pose projects them, and never a second measurement of a point the tracker already measured."""
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=46, per_level=(120, 50, 20, 8))
    o = make_oracle(capi.default_params(w, h, 1, grow_map=2), m, f.pose(-1))
    o_plain = make_oracle(capi.default_params(w, h, 1), m, f.pose(-1))
    refound = 0
    for i in range(46):
        o.track_frame(frames[i]); o_plain.track_frame(frames[i])
        st = o.state()
        assert st.quality == 2
        if st.kf_added:
            km = o.keyframe_meas(st.n_keyframes - 1)
            assert len(np.unique(km["pt"])) == len(km["pt"])
            refound += int((km["source"] == 1).sum())
            assert set(km["source"].tolist()) <= {0, 1}
    assert refound > 0
    assert o.state().n_points == o_plain.state().n_points == len(m["points"])
    assert pose_err(o.state().pose, f.pose(45)) < 6e-3


def test_idle_jobs_of_the_map_maker():
    """MapMaker::run's idle jobs (jni/MapMaker.cc:94-117) in the oracle: ReFindNewlyMade consumes the new queue, BundleAdjustAll runs
    once per new keyframe until it has converged, the outliers of the adjustments wait in the failure queue until the 20th pass
    re-finds most of them; one pass = the four jobs one by one (the split the device test uses)."""
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=32, per_level=(120, 50, 20, 8))
    kw = dict(patch_size=8, grow_map=3)
    a = make_oracle(capi.default_params(w, h, 1, idle_iterations=2, **kw), m, f.pose(-1))
    b = make_oracle(capi.default_params(w, h, 1, idle_iterations=0, **kw), m, f.pose(-1))
    seen_queue = 0
    for t in range(32):
        a.track_frame(frames[t]); b.track_frame(frames[t])
        for _ in range(2):
            for job in range(4):
                b.idle_job(job)
        sa, sb = a.idle_stats(), b.idle_stats()
        assert sa == sb, (t, sa, sb)                                   # a pass = its four jobs
        assert pose_err(a.state().pose, b.state().pose) == 0.0
        assert sa["new_queue"] == 0                                    # ReFindNewlyMade ran behind every AddKeyFrame
        seen_queue = max(seen_queue, sa["failure_queue"])
        assert pose_err(a.state().pose, f.pose(t)) < 5e-3
    assert sa["ba_all"] == a.state().n_keyframes - len(m["keyframes"])
    assert seen_queue > 100 and sa["failure_queue"] == 0 and sa["refound_failed"] > seen_queue // 2
    assert sa["refound_new"] > 0


def test_std_libm_variant_follows_the_default_build_within_the_free_running_bars():
    """the oracle evaluates sin / cos / tan / atan / asin / acos with the product's libm-free kernels, so device-vs-oracle bit
    parity of those functions holds by construction and says nothing about the reference's own libm.  This measures the substitution:
    the same free-running sequence (45 frames, two keyframes with their bundle adjustments) through the default build and through the
    variant built on glibc's libm (what a reference build on this host calls; oracle/Makefile, -DORC_STD_LIBM), in a child process.
    Bars: the north_star pose tolerance in every frame, the same keyframe frames, found counts within 3."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from helpers import make_scene, make_oracle\n"
        "from visualslam_android_amd import capi\n"
        "f, m, frames = make_scene(320, 240, seed=21, n_frames=45, per_level=(120, 50, 20, 8))\n"
        "o = make_oracle(capi.default_params(320, 240, 1), m, f.pose(-1))\n"
        "out = []\n"
        "for i in range(45):\n"
        "    o.track_frame(frames[i]); st = o.state()\n"
        "    out.append([list(st.pose), list(st.found), st.kf_added, st.n_keyframes, st.quality])\n"
        "print(json.dumps(out))\n") % (ROOT, os.path.join(ROOT, "tests"))
    runs = {}
    for variant in ("", "stdlibm"):
        env = dict(os.environ, ORC_ORACLE_VARIANT=variant)
        runs[variant] = json.loads(subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip().splitlines()[-1])
    worst, bits_equal = 0.0, 0
    for a, b in zip(runs[""], runs["stdlibm"]):
        d = float(np.abs(np.array(a[0]) - np.array(b[0])).max())
        worst = max(worst, d)
        bits_equal += a[0] == b[0]
        assert d < 1e-4, d
        assert a[2:] == b[2:] and np.abs(np.array(a[1]) - np.array(b[1])).max() <= 3, (a[1:], b[1:])
    assert runs[""][-1][3] >= 10                                  # two keyframes were added on top of the map's eight
    print("std-libm variant: worst pose difference %.3g, %d of %d frames bit-equal" % (worst, bits_equal, len(runs[""])))
