import os, sys, ctypes as C; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from visualslam_android_amd import capi
capi.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'visualslam_android_amd', 'libvslam_hip_baprof.so'))
from helpers import *
W,H=640,480
f,m,frames=make_scene(W,H,n_frames=2)
S=int(sys.argv[1]) if len(sys.argv)>1 else 1
vp=capi.default_params(W,H,S,patch_size=8, ba_window=int(sys.argv[2]) if len(sys.argv)>2 else 5)
g=capi.System(vp)
for s in range(S):
    g.load_map(s,m); g.set_pose(s,f.pose(-1))
lib=capi.load_library()
out=(C.c_ulonglong*32)()
lib.vslam_debug_ba_prof(out,1)
g.track_frame(np.stack([frames[0]]*S))
lib.vslam_debug_ba_prof(out,1)
st=g.state(0)
names=['layout','find_err_uncached','radix_sigma','sweep_free(+U merge)','sweep_fixed','U_generic','-','schur','solve','map_update+cam_new','find_new_error','commit','erase_outliers']
tot=sum(out[i] for i in range(13))
print('S',S,'ba_accepted',st.ba_accepted,'trials',st.n_ba_trials,'kf',st.n_keyframes, 'total Mcycles %.2f'%(tot/1e6))
for i,n in enumerate(names): print('%-16s %8.1f kcyc %5.1f%%'%(n,out[i]/1e3,100.0*out[i]/tot))
km=g.keyframe_meas(0, st.n_keyframes-1); print('meas in new kf',len(km['pt']))
