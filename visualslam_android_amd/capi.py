"""ctypes binding of include/vslam_c.h (libvslam_hip.so).  No compute happens in Python."""
import ctypes as C
import os

import numpy as np

LEVELS = 4
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvslam_hip.so")

Q_CAM_INT_RADIUS = 1
Q_POSE_INT_RESIDUAL = 2
Q_NONMAX_RIGHT_NEIGHBOUR = 4


class VslamError(RuntimeError):
    pass


class Params(C.Structure):
    """Mirror of struct vslam_params (include/vslam_c.h)."""
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int), ("n_streams", C.c_int),
        ("fast_threshold", C.c_int * LEVELS), ("nonmax_barrier", C.c_int), ("patch_size", C.c_int),
        ("max_corners", C.c_int * LEVELS), ("max_points", C.c_int), ("max_keyframes", C.c_int),
        ("max_patches_per_frame", C.c_int), ("coarse_min", C.c_int), ("coarse_max", C.c_int),
        ("coarse_range", C.c_int), ("coarse_subpix_its", C.c_int), ("coarse_disabled", C.c_int),
        ("coarse_min_vel", C.c_double), ("fine_subpix_its", C.c_int), ("wls_prior", C.c_double),
        ("use_sbi", C.c_int), ("min_frames_between_kf", C.c_int), ("max_kf_dist_wiggle_mult", C.c_double),
        ("wiggle_scale", C.c_double), ("ba_max_iterations", C.c_int), ("ba_convergence_limit", C.c_double),
        ("ba_min_tukey_sigma", C.c_double), ("ba_window", C.c_int), ("ba_min_keyframes", C.c_int),
        ("cam", C.c_double * 5), ("quirks", C.c_int), ("device", C.c_int),
    ]


_lib = None

# name -> (restype, argtypes); every symbol include/vslam_c.h declares
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_ip = C.POINTER(C.c_int)
_dp = C.POINTER(C.c_double)
_sys = C.c_void_p
SYMBOLS = {
    "vslam_last_error": (C.c_char_p, []),
    "vslam_default_params": (C.c_int, [C.POINTER(Params), C.c_int, C.c_int, C.c_int]),
    "vslam_create": (C.c_int, [C.POINTER(Params), C.POINTER(_sys)]),
    "vslam_destroy": (C.c_int, [_sys]),
    "vslam_synchronize": (C.c_int, [_sys]),
    "vslam_make_keyframe_lite": (C.c_int, [_sys, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "vslam_fast_nonmax": (C.c_int, [_sys]),
    "vslam_read_level_image": (C.c_int, [_sys, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "vslam_read_corners": (C.c_int, [_sys, C.c_int, C.c_int, C.c_void_p, C.c_int, _ip]),
    "vslam_read_row_lut": (C.c_int, [_sys, C.c_int, C.c_int, C.c_void_p]),
    "vslam_read_max_corners": (C.c_int, [_sys, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, _ip]),
}


def load_library(path=None):
    """Load libvslam_hip.so and bind every declared symbol.  Raises if the extension is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise VslamError("HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)" % p)
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise VslamError("vslam error %d: %s" % (rc, load_library().vslam_last_error().decode()))


def default_params(width, height, n_streams=1, **overrides):
    p = Params()
    _check(load_library().vslam_default_params(C.byref(p), width, height, n_streams))
    for k, v in overrides.items():
        cur = getattr(p, k)
        if isinstance(cur, C.Array):
            for i, x in enumerate(v):
                cur[i] = x
        else:
            setattr(p, k, v)
    return p


class System:
    """One vslam_system: n_streams independent sequences batched on one GPU."""

    def __init__(self, params):
        self.lib = load_library()
        self.params = params
        self.h = _sys()
        _check(self.lib.vslam_create(C.byref(params), C.byref(self.h)))
        self.S = params.n_streams

    def close(self):
        if self.h:
            self.lib.vslam_destroy(self.h)
            self.h = _sys()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self.lib.vslam_synchronize(self.h))

    def level_shape(self, level):
        return self.params.height >> level, self.params.width >> level

    # ---- front-end ---------------------------------------------------------------------------
    def make_keyframe_lite(self, gray):
        """gray: uint8 array [S, H, W] (host).  KeyFrame::MakeKeyFrame_Lite for all streams."""
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        if g.ndim == 2:
            g = g[None]
        assert g.shape == (self.S, self.params.height, self.params.width), g.shape
        _check(self.lib.vslam_make_keyframe_lite(self.h, g.ctypes.data, g.shape[2], g.shape[1] * g.shape[2], 0))
        self.synchronize()  # the host buffer is only borrowed for the call

    def make_keyframe_lite_device(self, dev_ptr, row_stride, stream_stride):
        _check(self.lib.vslam_make_keyframe_lite(self.h, dev_ptr, row_stride, stream_stride, 1))

    def fast_nonmax(self):
        _check(self.lib.vslam_fast_nonmax(self.h))

    def read_level_image(self, stream, level):
        h, w = self.level_shape(level)
        out = np.empty((h, w), dtype=np.uint8)
        _check(self.lib.vslam_read_level_image(self.h, stream, level, out.ctypes.data, w))
        return out

    def read_corners(self, stream, level):
        cap = self.params.max_corners[level]
        out = np.empty(cap, dtype=np.uint32)
        n = C.c_int(0)
        _check(self.lib.vslam_read_corners(self.h, stream, level, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value].copy()

    def read_row_lut(self, stream, level):
        h, _ = self.level_shape(level)
        out = np.empty(h, dtype=np.int32)
        _check(self.lib.vslam_read_row_lut(self.h, stream, level, out.ctypes.data))
        return out

    def read_max_corners(self, stream, level):
        cap = self.params.max_corners[level]
        out = np.empty(cap, dtype=np.uint32)
        sc = np.zeros(cap, dtype=np.int32)
        n = C.c_int(0)
        _check(self.lib.vslam_read_max_corners(self.h, stream, level, out.ctypes.data, sc.ctypes.data, cap, C.byref(n)))
        return out[:n.value].copy(), sc
