#!/usr/bin/env python3
"""Micro-benchmark of the frame front-end (MakeKeyFrame_Lite: k_pyr_fast0 + k_fast_lvl + k_compact) on resident frames.
   python tools/fe_bench.py [streams] [iters]      -> ms per call, frames/s, algorithmic GB/s (SURVEY 8(d) B_fast)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from visualslam_android_amd import capi, feeder

if os.environ.get("VSLAM_LIB"):
    capi.load_library(os.environ["VSLAM_LIB"])      # a diagnostic build from tools/build_variant.sh

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
W, H = 640, 480
f = feeder.Feeder(W, H, seed=1)
fr = f.render(0, 8)
frames = torch.from_numpy(np.ascontiguousarray(fr[np.arange(S) % 8])).cuda()
g = capi.System(capi.default_params(W, H, S))
torch.cuda.synchronize()
for _ in range(5):
    g.make_keyframe_lite_device(frames.data_ptr(), W, W * H)
g.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    g.make_keyframe_lite_device(frames.data_ptr(), W, W * H)
g.synchronize()
dt = (time.perf_counter() - t0) / N
nc = len(g.read_corners(0, 0))
b = S * (W * H * (1 + 21 / 64.0) + 4 * nc + 4 * sum(H >> l for l in range(4)))
print("S=%d  %.4f ms/call  %.0f frames/s  %.1f GB/s algorithmic  (%d L0 corners)" % (S, dt * 1e3, S / dt, b / dt / 1e9, nc))
