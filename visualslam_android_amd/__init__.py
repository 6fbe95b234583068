"""MI355X-native PTAM tracking + local bundle adjustment (hot path of ahcorde/visualSLAM_Android).

The product is libvslam_hip.so (hand-written HIP for gfx950 behind the C ABI of include/vslam_c.h).
This package is the thin Python mirror of that ABI used by tests and bench.py; there is no CPU
fallback: without the built extension, importing `capi` raises.
"""
from . import capi  # noqa: F401
from .capi import System, Params, VslamError, default_params, load_library  # noqa: F401

__all__ = ["capi", "System", "Params", "VslamError", "default_params", "load_library"]
