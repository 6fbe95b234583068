"""One bundle-adjustment round of S identical problems (the synchronous map-maker on a keyframe frame) for a PMC pass:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/pmc_ba_micro.py 256
prints the LM trials of the round and the problem's size, so that the counters of k_ba_compute can be put per trial."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from visualslam_android_amd import capi
from helpers import make_scene
W, H = 640, 480
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f, m, frames = make_scene(W, H, n_frames=2)
g = capi.System(capi.default_params(W, H, S, patch_size=8))
for s in range(S):
    g.load_map(s, m); g.set_pose(s, f.pose(-1))
g.track_frame(np.stack([frames[0]] * S)); g.synchronize()
st = g.state(0)
bs = g.bundle_stats(0) if hasattr(g, "bundle_stats") else None
print(json.dumps({"streams": S, "trials_per_problem": int(st.n_ba_trials), "kf": st.n_keyframes, "bundle_stats": bs}))
