"""Diagnostic: frame-by-frame parity of the map-growth path against the oracle at 640x480 (python tools/grow_parity_check.py)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import make_scene, make_oracle, pose_err
from visualslam_android_amd import capi
w, h = 640, 480
f, m, frames = make_scene(w, h, seed=1234, n_frames=48)
n0 = len(m["points"])
vp = capi.default_params(w, h, 1, patch_size=8, grow_map=1)
g = capi.System(vp); g.load_map(0, m); g.set_pose(0, f.pose(-1))
o = make_oracle(capi.default_params(w, h, 1, patch_size=8, grow_map=1), m, f.pose(-1))
for t in range(48):
    g.track_frame(frames[t][None]); o.track_frame(frames[t])
    so, sg = o.state(), g.state(0)
    d = pose_err(so.pose, sg.pose)
    line = "t %2d pose diff %.2e points %d/%d found %s/%s zm %d/%d ba %d/%d" % (t, d, so.n_points, sg.n_points, sum(so.found), sum(sg.found), so.n_zmssd, sg.n_zmssd, so.ba_accepted, sg.ba_accepted)
    if so.kf_added:
        po, pg = o.points(), g.points(0)
        n1 = min(so.n_points, sg.n_points)
        dp = np.abs(po["pos"][n0:n1] - pg["pos"][n0:n1])
        line += "  KF: new-point pos diff max %.2e (n=%d) all-point diff %.2e" % (dp.max() if len(dp) else 0, n1 - n0, np.abs(po["pos"][:n0] - pg["pos"][:n0]).max())
    print(line)
    if so.n_points != sg.n_points:
        k = so.n_keyframes - 1
        mo, mg = o.keyframe_meas(k), g.keyframe_meas(0, k)
        ro = {(round(x, 3), round(y, 3), int(l)) for (x, y), l, s_ in zip(mo["root"], mo["level"], mo["source"]) if s_ == 2}
        rg = {(round(x, 3), round(y, 3), int(l)) for (x, y), l, s_ in zip(mg["root"], mg["level"], mg["source"]) if s_ == 2}
        print("  roots only in oracle:", sorted(ro - rg)); print("  roots only on device:", sorted(rg - ro))
        from oracle import binding as orc
        for kf in range(so.n_keyframes):
            print('  kf', kf, 'device corners per level', [len(g.keyframe_corners(0, kf, l)) for l in range(4)])
        for t2 in (0, 21, 42):
            print('  oracle corners of frame', t2, [len(x[1]) for x in orc.make_keyframe_lite(frames[t2])])
        log = o.grow_log()
        for l in range(4):
            pos, sc = g.read_candidates(0, l)
            dev = {int(pp): (0 if c > 0 else int(-c)) for pp, c in zip(pos, sc)}
            # the oracle log holds every keyframe's calls; keep the last call per (level, position)
            orc_ = {}
            for lv, pp, why in log:
                if lv == l: orc_[int(pp)] = int(why)
            diff = [(pp & 0xFFFF, pp >> 16, orc_.get(pp), dev[pp]) for pp in dev if orc_.get(pp) != dev[pp]]
            print("  level", l, "candidates", len(dev), "differing (x, y, oracle stage, device stage):", diff[:10])
        break
