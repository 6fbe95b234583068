"""Shared scene/setup helpers for the whole-path parity tests."""
import numpy as np

from oracle import binding as orc
from visualslam_android_amd import capi, feeder


def oracle_corner_fn(thr=(10, 15, 15, 10), barrier=10):
    def fn(gray):
        out = []
        for img, c, _lut in orc.make_keyframe_lite(gray, thr):
            out.append(orc.nonmax(c, orc.fast_score(img, c, barrier)))
        return out
    return fn


def make_scene(w=640, h=480, seed=1234, n_frames=30, n_keyframes=8, noise=2, rects=None, **map_kw):
    """rects: rectangles of the feeder's texture (VSLAM_FEEDER_NRECT; default 1000 ~ 1100 FAST corners at level 0 of 640x480,
    1500 ~ 2000 at 1280x720 = BASELINE configs[3])"""
    import os
    old = os.environ.get("VSLAM_FEEDER_NRECT")
    if rects:
        os.environ["VSLAM_FEEDER_NRECT"] = str(rects)
    try:
        f = feeder.Feeder(w, h, seed=seed, noise=noise)
    finally:
        if rects:
            if old is None:
                os.environ.pop("VSLAM_FEEDER_NRECT", None)
            else:
                os.environ["VSLAM_FEEDER_NRECT"] = old
    m = feeder.build_map(f, oracle_corner_fn(), n_keyframes=n_keyframes, **map_kw)
    frames = f.render(0, n_frames)
    return f, m, frames


def make_oracle(vparams, m, start_pose):
    o = orc.OracleSystem(orc.params_from_vslam(vparams))
    o.load_map(m)
    o.set_pose(start_pose)
    return o


def pose_err(a, b):
    return float(np.abs(np.asarray(a[:]) - np.asarray(b[:])).max())


# ---- parity checks of the tracking path -----------------------------------------------------------------------------
POSE_TOL = 1e-4          # north_star tolerance on pose SE3 (free-running comparison)


def assert_tracker_exact(o, g, s, tag, templates=True):
    """Everything Tracker::TrackFrame produced for this frame, BIT FOR BIT: counters, found sets, chosen corners and
    sub-pixel positions, ZMSSD evaluation count, cached templates, pose, velocity, scene depth.  Holds whenever the two
    sides entered the frame with the same map and pose bits: the device evaluates the same expressions in the same order
    (vslam_libm.h transcendentals, sequential-order sums in the pose update and the sub-pixel iterations)."""
    so, sg = o.state(), g.state(s)
    to, tg = o.point_tracks(), g.point_tracks(s)
    assert (so.frame, so.quality, so.lost_frames, so.did_coarse, so.kf_added, so.n_keyframes, so.n_points) == \
           (sg.frame, sg.quality, sg.lost_frames, sg.did_coarse, sg.kf_added, sg.n_keyframes, sg.n_points), tag
    assert list(so.attempted) == list(sg.attempted) and list(so.found) == list(sg.found), (tag, list(so.attempted), list(sg.attempted), list(so.found), list(sg.found))
    assert so.n_zmssd == sg.n_zmssd, (tag, so.n_zmssd, sg.n_zmssd)
    pv = tg["level"] >= 0                                        # bFound / the level are stale for points outside this frame's PVS
    assert np.array_equal(to["searched"], tg["searched"]), tag
    assert np.array_equal(to["found"][pv], tg["found"][pv]), tag
    f = pv & (tg["found"] == 1)
    assert np.array_equal(to["level"][f], tg["level"][f]) and np.array_equal(to["subpix"][f], tg["subpix"][f]), tag
    assert np.array_equal(to["vfound"][f], tg["vfound"][f]), (tag, np.abs(to["vfound"][f] - tg["vfound"][f]).max())
    assert np.array_equal(np.array(so.pose[:]), np.array(sg.pose[:])), (tag, pose_err(so.pose, sg.pose))
    assert np.array_equal(np.array(so.velocity[:]), np.array(sg.velocity[:])), tag
    assert (so.msd_velocity, so.depth_mean, so.depth_sigma) == (sg.msd_velocity, sg.depth_mean, sg.depth_sigma), tag
    if templates:
        n = so.n_points
        ao, ag = o.templates(n), g.templates(s, n)
        srch = pv & (tg["searched"] == 1)
        assert np.array_equal(ao["have"][srch], ag["have"][srch]) and np.array_equal(ao["bad"][srch], ag["bad"][srch]), tag
        h = srch & (ag["have"] == 1)
        assert np.array_equal(ao["tmpl"][h], ag["tmpl"][h]), (tag, int((ao["tmpl"][h] != ag["tmpl"][h]).sum()))
        assert np.array_equal(ao["sum"][h], ag["sum"][h]) and np.array_equal(ao["sumsq"][h], ag["sumsq"][h]), tag


def assert_tracker_close(o, g, s, tag, tol=POSE_TOL):
    """Free-running comparison once the maps differ in the last bits (the bundle adjustment's parallel sums are not taken
    in the reference's order): the north_star bar on the pose, the same decisions, nearly the same found set."""
    so, sg = o.state(), g.state(s)
    d = pose_err(so.pose, sg.pose)
    assert d < tol, (tag, d)
    assert (so.frame, so.quality, so.lost_frames, so.did_coarse, so.kf_added, so.n_keyframes) == \
           (sg.frame, sg.quality, sg.lost_frames, sg.did_coarse, sg.kf_added, sg.n_keyframes), tag
    assert np.abs(np.array(so.attempted[:]) - np.array(sg.attempted[:])).max() <= 3, tag
    assert np.abs(np.array(so.found[:]) - np.array(sg.found[:])).max() <= 3, tag
    return d


def is_tracker_exact(o, g, s):
    try:
        assert_tracker_exact(o, g, s, "", templates=False)
        return True
    except AssertionError:
        return False


def assert_map_close(o, g, s, tag, tol):
    """Map after a bundle adjustment: same structure (keyframes, points, bad flags, measurement rows), positions and
    poses within tol (the adjustment's sums are tree reductions on the device, sequential in the oracle)."""
    so, sg = o.state(), g.state(s)
    assert (so.n_keyframes, so.n_points, so.ba_accepted, so.n_ba_trials) == (sg.n_keyframes, sg.n_points, sg.ba_accepted, sg.n_ba_trials), (tag, so.ba_accepted, sg.ba_accepted, so.n_ba_trials, sg.n_ba_trials)
    po, pg = o.points(), g.points(s)
    assert np.array_equal(po["bad"], pg["bad"]) and np.array_equal(po["n_in"], pg["n_in"]) and np.array_equal(po["n_out"], pg["n_out"]), tag
    dp = np.abs(po["pos"] - pg["pos"]).max() if len(po["pos"]) else 0.0
    assert dp < tol, (tag, dp)
    for k in range(so.n_keyframes):
        dk = pose_err(o.keyframe_pose(k), g.keyframe_pose(s, k))
        assert dk < tol, (tag, k, dk)


def assert_map_exact(o, g, s, tag):
    """The map bit for bit: structure, counters, point positions and keyframe poses == the oracle's (the bundle adjustment in its
    reference-order summation mode, vslam_params.ba_sum_order = 1)."""
    so, sg = o.state(), g.state(s)
    assert (so.n_keyframes, so.n_points, so.ba_accepted, so.n_ba_trials) == (sg.n_keyframes, sg.n_points, sg.ba_accepted, sg.n_ba_trials), (tag, so.ba_accepted, sg.ba_accepted, so.n_ba_trials, sg.n_ba_trials)
    po, pg = o.points(), g.points(s)
    assert np.array_equal(po["bad"], pg["bad"]) and np.array_equal(po["n_in"], pg["n_in"]) and np.array_equal(po["n_out"], pg["n_out"]), tag
    assert np.array_equal(po["pos"], pg["pos"]), (tag, np.abs(po["pos"] - pg["pos"]).max())
    for k in range(so.n_keyframes):
        assert np.array_equal(np.asarray(o.keyframe_pose(k)), np.asarray(g.keyframe_pose(s, k))), (tag, k, pose_err(o.keyframe_pose(k), g.keyframe_pose(s, k)))


def resync(o, g, s):
    """Copy the oracle's bits over the device's: tracker pose and velocity, map point positions, keyframe poses.  After
    this the next frame starts from identical state on both sides, so its tracking is reproducible bit for bit."""
    so = o.state()
    g.set_pose(s, so.pose[:])
    g.set_velocity(s, so.velocity[:])
    g.set_point_positions(s, o.points()["pos"])
    for k in range(so.n_keyframes):
        g.set_keyframe_pose(s, k, o.keyframe_pose(k))


def check_and_resync(o, g, s, tag, map_tol=1e-7, templates=True):
    """The per-frame check of the re-synchronised mode: tracker == oracle, the map within map_tol after a keyframe's
    bundle adjustment, then the oracle's bits copied over the device's."""
    assert_tracker_exact(o, g, s, tag, templates=templates)
    if o.state().kf_added:
        assert_map_close(o, g, s, tag, map_tol)
    resync(o, g, s)
