"""Stage profile of k_epipolar (diagnostic library from tools/build_baprof.sh): python tools/grow_phase_profile.py [streams]"""
import os, sys, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from visualslam_android_amd import capi
capi.load_library(os.path.join(ROOT, 'visualslam_android_amd', 'libvslam_hip_baprof.so'))
from helpers import make_scene
W, H = 640, 480
f, m, frames = make_scene(W, H, n_frames=26)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = capi.System(capi.default_params(W, H, S, patch_size=8, grow_map=3))
for s in range(S):
    g.load_map(s, m); g.set_pose(s, f.pose(-1))
lib = capi.load_library()
out = (C.c_ulonglong * 16)()
names = ['geometry', 'template', 'filter+zmssd', 'subpix', 'triangulation', 'wait for chunk', 'commit']
kf = 0
for t in range(26):
    lib.vslam_debug_grow_prof(out, 1)
    g.track_frame(np.stack([frames[t]] * S)); g.synchronize()
    lib.vslam_debug_grow_prof(out, 0)
    tot = sum(out[i] for i in range(7))
    if tot:
        kf += 1
        print('frame %d: keyframe event, k_epipolar block 0 total %.1f kcycles (4 levels), points now %d' % (t, tot / 1e3, g.state(0).n_points))
        for i, n in enumerate(names): print('  %-16s %9.1f kcyc %5.1f%%' % (n, out[i] / 1e3, 100.0 * out[i] / tot))
        rn = ['loop tail / skipped', 'point + cell loads', 'projection + warp', 'template', 'search + zmssd', 'sub-pixel']
        rt = sum(out[8 + i] for i in range(6))
        print('  k_refind block (0,0) wave 0 total %.1f kcycles' % (rt / 1e3))
        for i, n in enumerate(rn): print('  %-20s %9.1f kcyc %5.1f%%' % (n, out[8 + i] / 1e3, 100.0 * out[8 + i] / max(1, rt)))
