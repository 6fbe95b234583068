"""Diagnostic: frame-by-frame parity of grow_map (ReFindInSingleKeyFrame / AddSomeMapPoints) against the oracle
(python tools/refind_parity_check.py [grow_map] [patch])."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
from helpers import make_scene, make_oracle, pose_err
from visualslam_android_amd import capi
grow = int(sys.argv[1]) if len(sys.argv) > 1 else 2
patch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if len(sys.argv) > 3 and sys.argv[3] == "vga":
    w, h = 640, 480
    f, m, frames = make_scene(w, h, seed=1234, n_frames=46)
else:
    w, h = 320, 240
    f, m, frames = make_scene(w, h, seed=77, n_frames=46, per_level=(120, 50, 20, 8))
vp = capi.default_params(w, h, 1, patch_size=patch, grow_map=grow)
g = capi.System(vp); g.load_map(0, m); g.set_pose(0, f.pose(-1))
o = make_oracle(capi.default_params(w, h, 1, patch_size=patch, grow_map=grow), m, f.pose(-1))
for t in range(46):
    g.track_frame(frames[t][None]); o.track_frame(frames[t])
    so, sg = o.state(), g.state(0)
    to, tg = o.point_tracks(), g.point_tracks(0)
    fm = (to["found"] == 1) & (tg["found"] == 1) & (tg["level"] >= 0)
    dv = np.abs(to["vfound"][fm] - tg["vfound"][fm]).max() if fm.any() else 0
    print("t %2d pose diff %.2e points %d/%d found %s/%s zm %d/%d ba %d/%d vfound diff %.2e found-mismatch %d" % (
        t, pose_err(so.pose, sg.pose), so.n_points, sg.n_points, sum(so.found), sum(sg.found), so.n_zmssd, sg.n_zmssd, so.ba_accepted, sg.ba_accepted,
        dv, int((to["found"] != tg["found"]).sum())))
    if so.kf_added:
        po, pg = o.points(), g.points(0)
        n = min(so.n_points, sg.n_points)
        print("   KF: point pos diff %.2e  bad %d/%d" % (np.abs(po["pos"][:n] - pg["pos"][:n]).max(), int(po["bad"].sum()), int(pg["bad"].sum())))
        for k in range(so.n_keyframes):
            mo, mg = o.keyframe_meas(k), g.keyframe_meas(0, k)
            same = np.array_equal(mo["pt"], mg["pt"])
            print("   kf %d meas %d/%d same points %s root diff %.2e pose diff %.2e" % (
                k, len(mo["pt"]), len(mg["pt"]), same, np.abs(mo["root"] - mg["root"]).max() if same and len(mo["pt"]) else -1,
                pose_err(o.keyframe_pose(k), g.keyframe_pose(0, k))))
            if same and len(mo["pt"]):
                for j in np.where(np.abs(mo["root"] - mg["root"]).max(1) > 1e-6)[0][:5]:
                    print("     meas of point %d source %d level %d root %s / %s" % (mo["pt"][j], mo["source"][j], mo["level"][j], mo["root"][j], mg["root"][j]))
            if not same:
                print("     only oracle", sorted(set(mo["pt"].tolist()) - set(mg["pt"].tolist()))[:20], "only device", sorted(set(mg["pt"].tolist()) - set(mo["pt"].tolist()))[:20])
    bad = np.where(fm & (np.abs(to["vfound"] - tg["vfound"]).max(1) > 1e-6))[0]
    for i in bad[:4]:
        ot, gt = o.template(int(i)), g.template(0, int(i))
        print("   point %d level %d/%d subpix %d/%d vfound %s / %s image %s / %s" % (i, to["level"][i], tg["level"][i], to["subpix"][i], tg["subpix"][i], to["vfound"][i], tg["vfound"][i], to["image"][i], tg["image"][i]))
        print("   template differs in %d pixels; sums %s / %s" % (int((np.asarray(ot["tmpl"]).reshape(-1) != gt["tmpl"].reshape(-1)).sum()), (ot["sum"], ot["sumsq"]), (gt["sum"], gt["sumsq"])))
