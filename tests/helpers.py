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


def make_scene(w=640, h=480, seed=1234, n_frames=30, n_keyframes=8, noise=2, **map_kw):
    f = feeder.Feeder(w, h, seed=seed, noise=noise)
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
