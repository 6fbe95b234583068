"""Phase profile of k_pose (diagnostic library from tools/build_baprof.sh): python tools/pose_phase_profile.py [streams]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from visualslam_android_amd import capi
capi.load_library(os.environ.get('VSLAM_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'visualslam_android_amd', 'libvslam_hip_baprof.so'))
from helpers import *
W, H = 640, 480
f, m, frames = make_scene(W, H, n_frames=6)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = capi.System(capi.default_params(W, H, S, patch_size=8))
for s in range(S):
    g.load_map(s, m); g.set_pose(s, f.pose(-1))
lib = capi.load_library()
out = (C.c_ulonglong * 16)()
g.track_frame(np.stack([frames[0]] * S)); g.synchronize()
lib.vslam_debug_pose_prof(out, 1)
for t in range(1, 5):
    g.track_frame(np.stack([frames[t]] * S))
g.synchronize()
lib.vslam_debug_pose_prof(out, 1)
names = ['gather', 'step(project/linear)+e2', 'count', 'radix_select', 'accumulate', 'wave_reduce', 'solve(thread0)', 'exp(thread0)', '-', 'scatter+export+tail']
tot = sum(out[i] for i in range(0, 10))
st = g.state(0)
print('S', S, 'found', sum(st.found), 'total kcycles per k_pose launch %.1f' % (tot / 1e3 / 8))
for i in range(0, 10):
    print('%-26s %8.1f kcyc/launch %5.1f%%' % (names[i], out[i] / 1e3 / 8, 100.0 * out[i] / tot))
print('chain loop, wavefront 0 (chain): working %.1f, at the barrier %.1f kcyc/launch; wavefront 1 (producer): working %.1f, at the barrier %.1f' % tuple(out[i] / 1e3 / 8 for i in (10, 11, 12, 13)))
