"""Diagnostic: one seed of tools/parity_sweep.py's grow_map = 3 case, every frame printed (python tools/parity_case.py SEED)."""
import os, sys, traceback
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import make_oracle, make_scene, pose_err
from test_gpu_tracking import compare_frame, Drift
from visualslam_android_amd import capi
seed = int(sys.argv[1]); cfg = dict(patch_size=8, grow_map=3)
w, h, n = 320, 240, 34
f, m, frames = make_scene(w, h, seed=seed, n_frames=n, per_level=(120, 50, 20, 8))
g = capi.System(capi.default_params(w, h, 1, **cfg)); g.load_map(0, m); g.set_pose(0, f.pose(-1))
o = make_oracle(capi.default_params(w, h, 1, **cfg), m, f.pose(-1))
drift = Drift()
for t in range(n):
    g.track_frame(frames[t][None]); o.track_frame(frames[t])
    so, sg = o.state(), g.state(0)
    to, tg = o.point_tracks(), g.point_tracks(0)
    fm = (to["found"] == 1) & (tg["found"] == 1) & (tg["level"] >= 0)
    print(t, "pose diff %.2e" % pose_err(so.pose, sg.pose), "pts", so.n_points, sg.n_points, "found", list(so.found), list(sg.found), "zm", so.n_zmssd, sg.n_zmssd, "ba", so.ba_accepted, sg.ba_accepted, so.n_ba_trials, sg.n_ba_trials,
          "vfound max diff %.2e" % (np.abs(to["vfound"][fm] - tg["vfound"][fm]).max() if fm.any() else 0), "kf", so.kf_added, "drift", drift.seen)
    try:
        compare_frame(o, g, 0, "frame %d" % t, drift, tight=1e-6)
    except AssertionError:
        traceback.print_exc(limit=2)
        break
