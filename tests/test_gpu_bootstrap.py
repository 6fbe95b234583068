"""GPU: the map bootstrap (vslam_params.bootstrap) against the oracle.

The oracle's HomographyInit / CalcPlaneAligner mathematics is its OWN restatement of the reference (oracle/homography.cpp: Eigen's
two-sided Jacobi SVD, a closed-form symmetric eigen-solver), independent of the product's csrc/bootstrap_math.h (one-sided Jacobi,
cyclic Jacobi); only the definition of the random draws is common.  So the integer results -- the trails (MiniPatch forward / backward
matching, list order), the homography's inlier count, which trails become stereo points, their sub-pixel positions -- are compared
exactly, and everything computed FROM the homography (second camera pose, triangulated points, the adjusted map) to a tolerance; the
tracking that follows to the pose tolerance in units of the map's own baseline.  PARITY UNPINNED against the reference: it draws
from rand() and holds no fixture for this."""
import numpy as np
import pytest

import oracle.binding as orc
from helpers import pose_err
from visualslam_android_amd import capi, feeder

pytestmark = pytest.mark.gpu


def _mat(p):
    p = np.array(p[:])
    m = np.eye(4)
    m[:3, :3] = p[:9].reshape(3, 3); m[:3, 3] = p[9:]
    return m


@pytest.mark.parametrize("patch", [8, 11])
def test_trails_and_init_from_stereo_match_the_oracle(patch):
    w, h = 640, 480
    f = feeder.Feeder(w, h, seed=1234, noise=2)
    frames = f.render(0, 26)
    f2 = feeder.Feeder(w, h, seed=77, noise=2)
    frames2 = f2.render(0, 26)
    vp = capi.default_params(w, h, 2, patch_size=patch, grow_map=3, bootstrap=1)
    g = capi.System(vp)
    os_ = [orc.OracleSystem(orc.params_from_vslam(capi.default_params(w, h, 1, patch_size=patch, grow_map=3))) for _ in range(2)]
    fr = [frames, frames2]
    base = {}
    press = {0: (0, 14), 1: (2, 12)}                                   # stream -> frames of the two spacebar presses (independent streams)
    for t in range(26):
        for s in range(2):
            if t in press[s]:
                g.press_spacebar(s); os_[s].press_spacebar()
        g.track_frame(np.stack([fr[0][t], fr[1][t]]))
        for s in range(2):
            o = os_[s]
            o.track_frame(fr[s][t])
            io, ig = o.init_info(), g.init_info(s)
            tag = "patch %d stream %d frame %d" % (patch, s, t)
            assert (io["stage"], io["trails"], io["init_ok"], io["map_good"]) == (ig["stage"], ig["trails"], ig["init_ok"], ig["map_good"]), (tag, io, ig)
            if io["stage"] == 1:
                assert np.array_equal(o.trails(), g.trails(s)), tag      # same trails, same order, same positions
            if t == press[s][1]:                                          # the frame InitFromStereo ran in
                assert io["hom_inliers"] == ig["hom_inliers"] and io["stereo_points"] == ig["stereo_points"] > 100, (tag, io, ig)
                so, sg = o.state(), g.state(s)
                assert so.n_keyframes == sg.n_keyframes == 2
                n0 = io["stereo_points"]
                mo, mg = o.keyframe_meas(1), g.keyframe_meas(s, 1)
                trail_o, trail_g = mo["source"] == 3, mg["source"] == 3
                assert np.array_equal(mo["pt"][trail_o], mg["pt"][trail_g]) and np.array_equal(mo["root"][trail_o], mg["root"][trail_g]), tag   # sub-pixel positions of the stereo points: exact
                assert abs(so.n_points - sg.n_points) <= max(3, so.n_points // 50), (tag, so.n_points, sg.n_points)   # the epipolar growth works from adjusted poses
                # The finished map, in quantities the global alignment does not touch: the second camera relative to the first and the
                # stereo points in the first camera's frame.  When the adjustments took the same Levenberg-Marquardt path on both sides
                # (same number of trials) they agree like any adjustment does; a borderline accept / reject (observed: 63 against 64 trials
                # in one of the four cases) ends in a neighbouring optimum of the same quality.
                same_path = so.n_ba_trials == sg.n_ba_trials
                rel_o = _mat(o.keyframe_pose(1)) @ np.linalg.inv(_mat(o.keyframe_pose(0)))
                rel_g = _mat(g.keyframe_pose(s, 1)) @ np.linalg.inv(_mat(g.keyframe_pose(s, 0)))
                assert np.abs(rel_o - rel_g).max() < (1e-5 if same_path else 5e-3), (tag, np.abs(rel_o - rel_g).max())
                # (with ONE fixed camera the scale of the map is held only by the starting baseline: two adjustment paths end at
                # slightly different scales, so the other path's positions are compared in units of the baseline)
                bo, bg = (1.0, 1.0) if same_path else (np.linalg.norm(rel_o[:3, 3]), np.linalg.norm(rel_g[:3, 3]))
                po, pg = o.points(), g.points(s)
                assert np.array_equal(po["bad"][:n0], pg["bad"][:n0]) or not same_path, tag
                co = (_mat(o.keyframe_pose(0)) @ np.c_[po["pos"][:n0], np.ones(n0)].T).T
                cg = (_mat(g.keyframe_pose(s, 0)) @ np.c_[pg["pos"][:n0], np.ones(n0)].T).T
                assert np.abs(co[:, :3] / bo - cg[:, :3] / bg).max() < (1e-4 if same_path else 1.5), (tag, np.abs(co[:, :3] / bo - cg[:, :3] / bg).max())   # 1.5 baselines = 4 % of the scene depth
                for side_pts, side_k0 in ((po, o.keyframe_pose(0)), (pg, g.keyframe_pose(s, 0))):   # the dominant plane is z = 0, the cameras above it
                    z = side_pts["pos"][side_pts["bad"] == 0][:, 2]
                    c0 = -np.array(side_k0[:9]).reshape(3, 3).T @ np.array(side_k0[9:])
                    assert abs(np.median(z)) < 0.01 and c0[2] > 1.0, tag
                base[s] = (_mat(o.keyframe_pose(1)), _mat(g.keyframe_pose(s, 1)), np.linalg.norm(rel_o[:3, 3]), np.linalg.norm(rel_g[:3, 3]))
            if t > press[s][1]:                                           # both sides track the map they made, relative to its second keyframe
                so, sg = o.state(), g.state(s)
                assert so.quality == sg.quality == 2 and sum(sg.found) > 100, (tag, list(so.found), list(sg.found))
                assert abs(sum(so.found) - sum(sg.found)) <= max(5, sum(so.found) // 20), (tag, list(so.found), list(sg.found))
                ro, rg = _mat(so.pose) @ np.linalg.inv(base[s][0]), _mat(sg.pose) @ np.linalg.inv(base[s][1])
                # (rotation absolutely; the translation in units of each side's own baseline -- with one fixed camera the scale of a map is
                # whatever its adjustments left of the starting baseline, and the two sides' adjustments are independent from the homography on)
                assert np.abs(ro[:3, :3] - rg[:3, :3]).max() < 1e-3, (tag, np.abs(ro[:3, :3] - rg[:3, :3]).max())
                assert np.abs(ro[:3, 3] / base[s][2] - rg[:3, 3] / base[s][3]).max() < 2e-2, (tag, ro[:3, 3] / base[s][2], rg[:3, 3] / base[s][3])
    for s in range(2):
        assert g.state(s).n_keyframes >= 2 and os_[s].state().n_keyframes == g.state(s).n_keyframes
    g.close()


def test_init_from_stereo_with_host_keyframes_and_matches():
    """vslam_init_from_stereo = MapMaker::InitFromStereo(kFirst, kSecond, vMatches, se3) for a caller that owns the frames and the matches
    (jni/MapMaker.h:38, jni/MapMaker.cc:204-376) against the oracle's InitFromStereo on the same two images and matches.  The matches
    are the oracle's own trails after 12 frames.  Integers exactly (inlier count, which matches become points, their sub-pixel
    positions); the map in alignment-invariant quantities to the tolerance of the two sides' independent homography mathematics; then
    both sides track the following frames of the sequence."""
    w, h = 640, 480
    f = feeder.Feeder(w, h, seed=1234, noise=2)
    frames = f.render(0, 20)
    kw = dict(patch_size=8, grow_map=3)
    t_o = orc.OracleSystem(orc.params_from_vslam(capi.default_params(w, h, 1, **kw)))
    t_o.press_spacebar()
    for t in range(12):
        t_o.track_frame(frames[t])
    matches = t_o.trails()                                             # (x0, y0, x1, y1): frame 0 -> frame 11
    assert len(matches) > 150
    g = capi.System(capi.default_params(w, h, 1, bootstrap=1, **kw))
    o = orc.OracleSystem(orc.params_from_vslam(capi.default_params(w, h, 1, **kw)))
    ok_g, pose_g = g.init_from_stereo(frames[0], frames[11], matches)
    ok_o, pose_o = o.init_from_stereo(frames[0], frames[11], matches)
    assert ok_g and ok_o
    io, ig = o.init_info(), g.init_info(0)
    assert (io["stage"], io["init_ok"], io["map_good"], io["hom_inliers"], io["stereo_points"]) == (ig["stage"], ig["init_ok"], ig["map_good"], ig["hom_inliers"], ig["stereo_points"]), (io, ig)
    assert io["stereo_points"] > 100
    mo, mg = o.keyframe_meas(1), g.keyframe_meas(0, 1)
    tr_o, tr_g = mo["source"] == 3, mg["source"] == 3
    assert np.array_equal(mo["pt"][tr_o], mg["pt"][tr_g]) and np.array_equal(mo["root"][tr_o], mg["root"][tr_g])      # sub-pixel positions of the stereo points
    rel_o = _mat(o.keyframe_pose(1)) @ np.linalg.inv(_mat(o.keyframe_pose(0)))
    rel_g = _mat(g.keyframe_pose(0, 1)) @ np.linalg.inv(_mat(g.keyframe_pose(0, 0)))
    bo, bg = np.linalg.norm(rel_o[:3, 3]), np.linalg.norm(rel_g[:3, 3])
    assert np.abs(rel_o[:3, :3] - rel_g[:3, :3]).max() < 5e-3 and np.abs(rel_o[:3, 3] / bo - rel_g[:3, 3] / bg).max() < 2e-2
    assert pose_err(pose_g, g.state(0).pose) == 0.0
    with pytest.raises(capi.VslamError):
        g.init_from_stereo(frames[0], frames[11], matches)             # the stream has a map now
    for t in range(12, 20):
        g.track_frame(frames[t][None]); o.track_frame(frames[t])
        so, sg = o.state(), g.state(0)
        assert so.quality == sg.quality == 2 and sum(sg.found) > 100, (t, list(so.found), list(sg.found))
    g.close()


def test_bootstrap_edge_cases_and_map_dump_round_trip(tmp_path):
    """A press with too few trails left resets the stage (jni/Tracker.cc:266-269); a stream with a map ignores the key; the
    map InitFromStereo made survives vslam_save_map -> vslam_read_map_dump to the 6 digits the format keeps."""
    w, h = 320, 240
    f = feeder.Feeder(w, h, seed=5, noise=2)
    frames = f.render(0, 14)
    with pytest.raises(capi.VslamError):
        capi.System(capi.default_params(w, h, 1, bootstrap=1))           # needs grow_map
    g = capi.System(capi.default_params(w, h, 1, grow_map=3, bootstrap=1))
    o = orc.OracleSystem(orc.params_from_vslam(capi.default_params(w, h, 1, grow_map=3)))
    for t in range(14):
        if t in (1, 9):
            g.press_spacebar(0); o.press_spacebar()
        g.track_frame(frames[t][None]); o.track_frame(frames[t])
        assert g.init_info(0)["stage"] == o.init_info()["stage"] and g.init_info(0)["trails"] == o.init_info()["trails"], t
    assert g.init_info(0)["map_good"] == o.init_info()["map_good"]
    if g.init_info(0)["map_good"]:
        import ctypes as C
        (tmp_path / "keyframes").mkdir()
        g.save_map(0, str(tmp_path))
        n, nk = C.c_int(0), C.c_int(0)
        pos = np.zeros((4096, 3)); lev = np.zeros(4096, np.int32); poses = np.zeros((8, 12))
        assert g.lib.vslam_read_map_dump(str(tmp_path).encode(), pos.ctypes.data, lev.ctypes.data, 4096, C.byref(n), poses.ctypes.data, 8, C.byref(nk)) == 0
        pts = g.points(0)
        good = pts["bad"] == 0
        assert n.value == int(good.sum()) and nk.value == g.state(0).n_keyframes
        assert np.allclose(pos[:n.value], pts["pos"][good], rtol=2e-5, atol=1e-6)
        for k in range(nk.value):
            assert np.allclose(poses[k], np.array(g.keyframe_pose(0, k)[:]), rtol=2e-5, atol=1e-6)
        # vslam_load_map: the dumped estimate written back over a disturbed map
        g.set_point_positions(0, pts["pos"] + 0.01)
        g.set_keyframe_pose(0, 1, np.array(g.keyframe_pose(0, 0)[:]))
        g.load_map_dump(0, tmp_path)
        after = g.points(0)
        assert np.allclose(after["pos"][good], pts["pos"][good], rtol=2e-5, atol=1e-6)
        assert np.allclose(np.array(g.keyframe_pose(0, 1)[:]), poses[1], rtol=1e-12, atol=1e-12)
    g.close()
