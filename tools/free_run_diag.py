"""Diagnostic: per-frame pose difference device vs oracle in the free-running mode (no re-synchronisation)."""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import make_scene, make_oracle, pose_err
from visualslam_android_amd import capi

w, h, patch, grow, seed, n = 640, 480, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
f, m, frames = make_scene(w, h, seed=seed, n_frames=n)
vp = capi.default_params(w, h, 1, patch_size=patch, grow_map=grow)
g = capi.System(vp); g.load_map(0, m); g.set_pose(0, f.pose(-1))
o = make_oracle(vp, m, f.pose(-1))
for t in range(n):
    g.track_frame(frames[t][None]); o.track_frame(frames[t])
    so, sg = o.state(), g.state(0)
    to, tg = o.point_tracks(), g.point_tracks(0)
    pv = tg["level"] >= 0
    fd = int((to["found"][pv] != tg["found"][pv]).sum())
    both = pv & (to["found"] == 1) & (tg["found"] == 1)
    dv = np.abs(to["vfound"][both] - tg["vfound"][both]).max(1) if both.any() else np.zeros(1)
    po, pg = o.points(), g.points(0)
    npt = min(len(po["pos"]), len(pg["pos"]))
    print("frame %2d pose %.2e  found-diff %d  vfound>1e-6: %d (max %.3g)  map %.2e  kf %d/%d pts %d/%d ba %d/%d" % (
        t, pose_err(so.pose, sg.pose), fd, int((dv > 1e-6).sum()), dv.max(), np.abs(po["pos"][:npt] - pg["pos"][:npt]).max(),
        so.n_keyframes, sg.n_keyframes, so.n_points, sg.n_points, so.n_ba_trials, sg.n_ba_trials))
