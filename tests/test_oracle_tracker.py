"""CPU test: the oracle's TrackFrame + AddKeyFrame + BundleAdjustRecent on a small synthetic sequence
(BASELINE.json configs[0]: plumbing of the reference CPU path, no GPU)."""
import numpy as np

from helpers import make_oracle, make_scene, pose_err
from visualslam_android_amd import capi


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
