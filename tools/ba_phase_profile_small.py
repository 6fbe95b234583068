"""Phase profile (clock64 stamps of workgroup 0, -DVSLAM_BA_PROF build) of Bundle::Compute on BASELINE configs[2]-sized problems
(5 cameras x 300 points) through the stand-alone Bundle:   python tools/ba_phase_profile_small.py [problems]"""
import os, sys, ctypes as C; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from visualslam_android_amd import capi
capi.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'visualslam_android_amd', 'libvslam_hip_baprof.so'))
from visualslam_android_amd.ba_scene import ba_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sc = ba_scene(n_cams=5, n_pts=300, pixel_noise=0.5, outlier_frac=0.05, seed=1, n_fixed=1)
MP = int(sys.argv[2]) if len(sys.argv) > 2 else 320
MM = int(sys.argv[3]) if len(sys.argv) > 3 else 1600
g = capi.Bundle(capi.default_params(640, 480, 1, ba_max_iterations=10), N, 8, MP, MM)
for n in range(N):
    g.set_problem(n, sc["cams_init"], [int(f) for f in sc["fixed"]], sc["pts_init"], [m[0] for m in sc["meas"]], [m[1] for m in sc["meas"]], [m[2] for m in sc["meas"]], [m[3] for m in sc["meas"]])
lib = capi.load_library()
out = (C.c_ulonglong * 32)()
g.compute(); lib.vslam_debug_ba_prof(out, 1)
g.compute(); ms, st = g.timing(); lib.vslam_debug_ba_prof(out, 1)
names = ['layout', 'find_err_uncached', 'radix_sigma', 'sweep_free(+U merge)', 'sweep_fixed', 'U_generic', '-', 'schur', 'solve', 'map_update+cam_new', 'find_new_error', 'commit', 'erase_outliers']
tot = sum(out[i] for i in range(13))
print('problems', N, 'trials', st["trials"] // max(1, st["problems"]), 'meas', len(sc["meas"]), 'launch ms %.3f' % ms, 'total kcycles (workgroup 0) %.1f' % (tot / 1e3))
for i, n in enumerate(names): print('%-22s %8.1f kcyc %5.1f%%' % (n, out[i] / 1e3, 100.0 * out[i] / tot))
