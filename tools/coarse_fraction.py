import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from visualslam_android_amd import capi, feeder
W, H, S, T = 640, 480, 64, 30
seeds = [1234 + 16 * s for s in range(S)]
feeders = [feeder.Feeder(W, H, seed=sd) for sd in seeds]
fe = capi.System(capi.default_params(W, H, 1))
def corner_fn(gray):
    fe.make_keyframe_lite(gray[None]); fe.fast_nonmax()
    return [fe.read_max_corners(0, l)[0] for l in range(4)]
maps = [feeder.build_map(f, corner_fn) for f in feeders]
g = capi.System(capi.default_params(W, H, S))
for s in range(S):
    g.load_map(s, maps[s]); g.set_pose(s, feeders[s].pose(-1))
frames = np.stack([f.render(0, T) for f in feeders], 1)   # [T, S, H, W]
tot = 0
for t in range(T):
    g.track_frame(frames[t])
    c = sum(g.state(s).did_coarse for s in range(S))
    tot += c
    print(t, c)
print("fraction of stream-frames with a coarse stage: %.3f" % (tot / (S * T)))
