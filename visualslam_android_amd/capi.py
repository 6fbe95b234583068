"""ctypes binding of include/vslam_c.h (libvslam_hip.so).  No compute happens in Python."""
import threading
import ctypes as C
import os

import numpy as np

LEVELS = 4
N_STAGES = 14
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
        ("cam", C.c_double * 5), ("quirks", C.c_int), ("device", C.c_int), ("ba_delay_frames", C.c_int),
        ("grow_map", C.c_int), ("ba_batch_frames", C.c_int), ("idle_iterations", C.c_int), ("bootstrap", C.c_int), ("ba_sum_order", C.c_int),
    ]


class TrackState(C.Structure):
    """Mirror of struct vslam_track_state."""
    _fields_ = [("pose", C.c_double * 12), ("velocity", C.c_double * 6), ("msd_velocity", C.c_double),
                ("depth_mean", C.c_double), ("depth_sigma", C.c_double), ("attempted", C.c_int * 4), ("found", C.c_int * 4),
                ("quality", C.c_int), ("lost_frames", C.c_int), ("frame", C.c_int), ("did_coarse", C.c_int),
                ("kf_added", C.c_int), ("n_keyframes", C.c_int), ("n_points", C.c_int), ("ba_accepted", C.c_int),
                ("n_zmssd", C.c_longlong), ("n_ba_trials", C.c_longlong)]


_lib = None

# name -> (restype, argtypes); every symbol include/vslam_c.h declares
_ip = C.POINTER(C.c_int)
_sys = C.c_void_p
_vp = C.c_void_p
_i, _d, _sz = C.c_int, C.c_double, C.c_size_t
SYMBOLS = {
    "vslam_last_error": (C.c_char_p, []),
    "vslam_default_params": (_i, [C.POINTER(Params), _i, _i, _i]),
    "vslam_create": (_i, [C.POINTER(Params), C.POINTER(_sys)]),
    "vslam_destroy": (_i, [_sys]),
    "vslam_synchronize": (_i, [_sys]),
    "vslam_eval_transcendental": (_i, [_i, _i, _vp, _vp, _i]),
    "vslam_make_keyframe_lite": (_i, [_sys, _vp, _sz, _sz, _i]),
    "vslam_fast_nonmax": (_i, [_sys]),
    "vslam_read_level_image": (_i, [_sys, _i, _i, _vp, _sz]),
    "vslam_read_corners": (_i, [_sys, _i, _i, _vp, _i, _ip]),
    "vslam_read_row_lut": (_i, [_sys, _i, _i, _vp]),
    "vslam_read_max_corners": (_i, [_sys, _i, _i, _vp, _vp, _i, _ip]),
    "vslam_get_keyframe_corners": (_i, [_sys, _i, _i, _i, _vp, _i, _ip]),
    "vslam_read_sbi": (_i, [_sys, _i, _vp, _vp, _vp]),
    "vslam_make_keyframe_rest": (_i, [_sys, C.c_double]),
    "vslam_thin_candidates": (_i, [_sys, _i]),
    "vslam_read_candidates": (_i, [_sys, _i, _i, _vp, _vp, _i, _ip]),
    "vslam_minipatch_sample": (_i, [_sys, _i, _i, _vp, _vp, _vp]),
    "vslam_minipatch_find": (_i, [_sys, _i, _i, _vp, _vp, _i, _i, _vp]),
    "vslam_add_keyframe": (_i, [_sys, _i]),
    "vslam_map_add_keyframe": (_i, [_sys, _i, _vp, _i, _vp, _sz, _d, _d]),
    "vslam_map_add_keyframes": (_i, [_sys, _i, _i, _vp, _vp, _vp, _sz, _sz, _vp]),
    "vslam_map_add_point": (_i, [_sys, _i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "vslam_map_add_points": (_i, [_sys, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vslam_map_add_measurement": (_i, [_sys, _i, _i, _i, _i, _vp, _i, _i]),
    "vslam_map_add_measurements": (_i, [_sys, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vslam_map_set_good": (_i, [_sys, _i]),
    "vslam_set_pose": (_i, [_sys, _i, _vp]),
    "vslam_set_velocity": (_i, [_sys, _i, _vp]),
    "vslam_set_last_keyframe_dropped": (_i, [_sys, _i, _i]),
    "vslam_track_frame": (_i, [_sys, _vp, _sz, _sz, _i]),
    "vslam_update": (_i, [_sys, _vp, _sz, _sz]),
    "vslam_patch_search": (_i, [_sys, _i]),
    "vslam_pose_update": (_i, [_sys, _i]),
    "vslam_finish_frame": (_i, [_sys]),
    "vslam_map_set_point_positions": (_i, [_sys, _i, _i, _i, _vp]),
    "vslam_map_set_keyframe_pose": (_i, [_sys, _i, _i, _vp]),
    "vslam_touch": (_i, [_sys]),
    "vslam_get_state": (_i, [_sys, _i, C.POINTER(TrackState)]),
    "vslam_get_states": (_i, [_sys, _i, _i, _vp]),
    "vslam_get_message": (_i, [_sys, _i, C.c_char_p, _sz]),
    "vslam_need_new_keyframe": (_i, [_sys, _i, _ip]),
    "vslam_distance_to_nearest_keyframe_excessive": (_i, [_sys, _i, _ip]),
    "vslam_get_point_tracks": (_i, [_sys, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "vslam_get_points": (_i, [_sys, _i, _vp, _vp, _vp, _vp, _i]),
    "vslam_get_keyframe_pose": (_i, [_sys, _i, _i, _vp]),
    "vslam_save_map": (_i, [_sys, _i, C.c_char_p]),
    "vslam_get_bundle_stats": (_i, [_sys, _i, _vp]),
    "vslam_get_idle_stats": (_i, [_sys, _i, _vp]),
    "vslam_mapmaker_idle_job": (_i, [_sys, _i]),
    "vslam_press_spacebar": (_i, [_sys, _i]),
    "vslam_init_from_stereo": (_i, [_sys, _vp, _vp, _sz, _i, _vp, _vp]),
    "vslam_set_boot_seed": (_i, [_sys, _i, C.c_uint]),
    "vslam_get_init_info": (_i, [_sys, _i, _vp]),
    "vslam_get_trails": (_i, [_sys, _i, _vp, _i, _vp]),
    "vslam_read_map_dump": (_i, [C.c_char_p, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "vslam_load_map": (_i, [_sys, _i, C.c_char_p]),
    "vslam_get_keyframe_measurements": (_i, [_sys, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "vslam_get_template": (_i, [_sys, _i, _i, _vp, _ip, _ip, _ip]),
    "vslam_get_templates": (_i, [_sys, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "vslam_stage_name": (C.c_char_p, [_i]),
    "vslam_profile_begin": (_i, [_sys, _i]),
    "vslam_profile_end": (_i, [_sys, _vp, _ip]),
    "vslam_profile_launches": (_i, [_sys, _vp]),
    "vslam_profile_ba_stats": (_i, [_sys, _vp]),
    "vslam_get_ba_launch_totals": (_i, [_sys, _vp]),
    "vslam_get_mapmaker_timing": (_i, [_sys, _vp, _vp]),
    "vslam_bundle_adjust_recent": (_i, [_sys]),
    "vslam_bundle_adjust_all": (_i, [_sys]),
    "vslam_bundle_create": (_i, [C.POINTER(Params), _i, _i, _i, _i, C.POINTER(_vp)]),
    "vslam_bundle_destroy": (_i, [_vp]),
    "vslam_bundle_add_camera": (_i, [_vp, _i, _vp, _i]),
    "vslam_bundle_add_point": (_i, [_vp, _i, _vp]),
    "vslam_bundle_add_meas": (_i, [_vp, _i, _i, _i, _vp, _d]),
    "vslam_bundle_compute": (_i, [_vp]),
    "vslam_bundle_set_problem": (_i, [_vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "vslam_bundle_get_timing": (_i, [_vp, _vp, _vp]),
    "vslam_bundle_synchronize": (_i, [_vp]),
    "vslam_bundle_get_result": (_i, [_vp, _i, _ip, _ip, C.POINTER(_d), C.POINTER(_d), C.POINTER(C.c_longlong)]),
    "vslam_bundle_get_camera": (_i, [_vp, _i, _i, _vp]),
    "vslam_bundle_get_point": (_i, [_vp, _i, _i, _vp]),
    "vslam_bundle_get_outlier_meas": (_i, [_vp, _i, _vp, _i]),
    "vslam_bundle_get_outlier_points": (_i, [_vp, _i, _vp, _i]),
}


_load_lock = threading.RLock()


def load_library(path=None):
    """Load libvslam_hip.so and bind every declared symbol.  Raises if the extension is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    with _load_lock:                           # feeders are created from a thread pool: bind exactly one CDLL object
        return _load_library_locked(path)


def _load_library_locked(path):
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
    if rc < 0:
        raise VslamError("vslam error %d: %s" % (rc, load_library().vslam_last_error().decode()))
    return rc


def read_map_dump(directory):
    """Reads what `System.save_map` / the reference's "SaveMap" wrote: (positions [n, 3], source levels [n], keyframe poses
    [k, 12] as rotation rows then translation).  The text holds 6 significant digits."""
    tok = open(os.path.join(directory, "map.dump")).read().split()
    a = np.array(tok, dtype=np.float64).reshape(-1, 4) if tok else np.zeros((0, 4))
    poses = []
    k = 0
    while os.path.exists(os.path.join(directory, "keyframes", "%d.info" % k)):
        m = np.array(open(os.path.join(directory, "keyframes", "%d.info" % k)).read().split(), dtype=np.float64).reshape(3, 4)
        poses.append(np.concatenate([m[:, :3].reshape(-1), m[:, 3]]))
        k += 1
    return a[:, :3], a[:, 3].astype(np.int32), np.array(poses).reshape(-1, 12)


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


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


TRANSCENDENTALS = ("sin", "cos", "tan", "atan", "asin", "acos", "sqrt", "rcp")


def eval_transcendental(name, x, on_host=False):
    """csrc/vslam_libm.h (plus sqrt and the reciprocal) over the array x, on the device or as compiled for the host."""
    x = _f64(x).reshape(-1)
    y = np.zeros_like(x)
    _check(load_library().vslam_eval_transcendental(TRANSCENDENTALS.index(name), len(x), x.ctypes.data, y.ctypes.data, int(on_host)))
    return y


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
    def _frames(self, gray):
        g = np.ascontiguousarray(gray, dtype=np.uint8)
        if g.ndim == 2:
            g = g[None]
        assert g.shape == (self.S, self.params.height, self.params.width), g.shape
        return g

    def make_keyframe_lite(self, gray):
        """gray: uint8 array [S, H, W] (host).  KeyFrame::MakeKeyFrame_Lite for all streams."""
        g = self._frames(gray)
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

    def minipatch_sample(self, stream, pos_xy):
        pos = np.ascontiguousarray(pos_xy, np.int32).reshape(-1, 2)
        n = len(pos)
        patches = np.zeros((n, 9, 9), np.uint8)
        ok = np.zeros(n, np.int32)
        _check(self.lib.vslam_minipatch_sample(self.h, stream, n, pos.ctypes.data, patches.ctypes.data, ok.ctypes.data))
        return patches, ok

    def minipatch_find(self, stream, patches, pos_xy, rng=10, max_ssd=100000):
        pos = np.ascontiguousarray(pos_xy, np.int32).reshape(-1, 2).copy()
        p = np.ascontiguousarray(patches, np.uint8)
        found = np.zeros(len(pos), np.int32)
        _check(self.lib.vslam_minipatch_find(self.h, stream, len(pos), p.ctypes.data, pos.ctypes.data, rng, max_ssd, found.ctypes.data))
        return found, pos

    def add_keyframe_now(self, stream=-1):
        _check(self.lib.vslam_add_keyframe(self.h, stream))
        self.synchronize()

    # ---- map ---------------------------------------------------------------------------------
    def add_keyframe(self, stream, pose12, fixed, gray, dmean, dsigma):
        g = np.ascontiguousarray(gray, np.uint8)
        return _check(self.lib.vslam_map_add_keyframe(self.h, stream, _f64(pose12).ctypes.data, int(fixed), g.ctypes.data, g.shape[1], dmean, dsigma))

    def add_point(self, stream, pos, src_kf, level, irx, iry, right, down):
        return _check(self.lib.vslam_map_add_point(self.h, stream, _f64(pos).ctypes.data, src_kf, level, irx, iry,
                                                   _f64(right).ctypes.data, _f64(down).ctypes.data))

    def load_map(self, stream, m):
        """m: dict from visualslam_android_amd.feeder.build_map"""
        kfs = m["keyframes"]
        if kfs:                                  # all keyframes in one call: one pyramid launch per level
            img = np.ascontiguousarray(np.stack([k["image"] for k in kfs]), np.uint8)
            poses = np.ascontiguousarray(np.stack([np.asarray(k["pose"], np.float64) for k in kfs]))
            fixed = np.array([1 if k["fixed"] else 0 for k in kfs], np.int32)
            depth = np.array([[k["depth_mean"], k["depth_sigma"]] for k in kfs], np.float64)
            _check(self.lib.vslam_map_add_keyframes(self.h, stream, len(kfs), poses.ctypes.data, fixed.ctypes.data, img.ctypes.data, img.shape[2],
                                                    img.shape[1] * img.shape[2], depth.ctypes.data))
        pk = m.get("packed") if hasattr(m, "get") else None
        if pk is not None and "points" not in dict.keys(m) and "meas" not in dict.keys(m):   # untouched build_map output: no per-item Python
            pos, right, down = (np.ascontiguousarray(pk[k_], np.float64) for k_ in ("pos", "right", "down"))
            kf_, lv_, ir = (np.ascontiguousarray(pk[k_], np.int32) for k_ in ("src_kf", "level", "ir"))
            npts = len(pos)
            kf, pt, lv, sp, src = (np.ascontiguousarray(pk[k_], np.int32) for k_ in ("m_kf", "m_pt", "m_level", "m_subpix", "m_source"))
            root = np.ascontiguousarray(pk["m_root"], np.float64)
        else:
            pts = m["points"]
            npts = len(pts)
            pos = np.array([q["pos"] for q in pts], np.float64).reshape(-1, 3); right = np.array([q["right"] for q in pts], np.float64).reshape(-1, 3)
            down = np.array([q["down"] for q in pts], np.float64).reshape(-1, 3); kf_ = np.array([q["src_kf"] for q in pts], np.int32)
            lv_ = np.array([q["level"] for q in pts], np.int32); ir = np.array([[q["irx"], q["iry"]] for q in pts], np.int32).reshape(-1, 2)
            ms = m["meas"]
            kf = np.array([x[0] for x in ms], np.int32); pt = np.array([x[1] for x in ms], np.int32)
            lv = np.array([x[2] for x in ms], np.int32); root = np.array([[x[3], x[4]] for x in ms], np.float64).reshape(-1, 2)
            sp = np.array([x[5] for x in ms], np.int32); src = np.array([x[6] for x in ms], np.int32)
        if npts:
            _check(self.lib.vslam_map_add_points(self.h, stream, npts, pos.ctypes.data, kf_.ctypes.data, lv_.ctypes.data,
                                                 ir.ctypes.data, right.ctypes.data, down.ctypes.data))
        n = len(kf)
        _check(self.lib.vslam_map_add_measurements(self.h, stream, n, kf.ctypes.data, pt.ctypes.data, lv.ctypes.data,
                                                   root.ctypes.data, sp.ctypes.data, src.ctypes.data))
        _check(self.lib.vslam_map_set_good(self.h, stream))

    def set_pose(self, stream, pose12):
        _check(self.lib.vslam_set_pose(self.h, stream, _f64(pose12).ctypes.data))

    def set_velocity(self, stream, v6):
        _check(self.lib.vslam_set_velocity(self.h, stream, _f64(v6).ctypes.data))

    def set_last_keyframe_dropped(self, stream, frame):
        _check(self.lib.vslam_set_last_keyframe_dropped(self.h, stream, int(frame)))

    # ---- tracking ----------------------------------------------------------------------------
    def track_frame(self, gray):
        """Tracker::TrackFrame on host frames [S, H, W]; synchronous (vslam_update)."""
        g = self._frames(gray)
        _check(self.lib.vslam_update(self.h, g.ctypes.data, g.shape[2], g.shape[1] * g.shape[2]))

    def track_frame_device(self, dev_ptr, row_stride, stream_stride):
        _check(self.lib.vslam_track_frame(self.h, dev_ptr, row_stride, stream_stride, 1))

    # TrackFrame stage by stage: make_keyframe_lite, patch_search(0), pose_update(0), patch_search(1), pose_update(1), finish_frame
    def patch_search(self, stage):
        _check(self.lib.vslam_patch_search(self.h, int(stage)))

    def pose_update(self, stage):
        _check(self.lib.vslam_pose_update(self.h, int(stage)))

    def finish_frame(self):
        _check(self.lib.vslam_finish_frame(self.h))
        self.synchronize()

    def set_point_positions(self, stream, pos, first=0):
        a = _f64(pos).reshape(-1, 3)
        _check(self.lib.vslam_map_set_point_positions(self.h, stream, int(first), len(a), a.ctypes.data))

    def set_keyframe_pose(self, stream, kf, pose12):
        _check(self.lib.vslam_map_set_keyframe_pose(self.h, stream, int(kf), _f64(pose12).ctypes.data))

    def states(self, first=0, n=None):
        """-> list of TrackState of the streams [first, first + n) with one copy (vslam_get_states)"""
        n = self.params.n_streams - first if n is None else n
        arr = (TrackState * n)()
        _check(self.lib.vslam_get_states(self.h, first, n, C.cast(arr, C.c_void_p)))
        return list(arr)

    def state(self, stream):
        s = TrackState()
        _check(self.lib.vslam_get_state(self.h, stream, C.byref(s)))
        return s

    def need_new_keyframe(self, stream):
        v = C.c_int(0)
        _check(self.lib.vslam_need_new_keyframe(self.h, stream, C.byref(v)))
        return bool(v.value)

    def distance_to_nearest_keyframe_excessive(self, stream):
        v = C.c_int(0)
        _check(self.lib.vslam_distance_to_nearest_keyframe_excessive(self.h, stream, C.byref(v)))
        return bool(v.value)

    def message(self, stream):
        buf = C.create_string_buffer(512)
        _check(self.lib.vslam_get_message(self.h, stream, buf, 512))
        return buf.value.decode()

    def point_tracks(self, stream):
        n = self.state(stream).n_points
        found, searched, level, subpix = (np.zeros(n, np.int32) for _ in range(4))
        vfound, image = np.zeros((n, 2)), np.zeros((n, 2))
        _check(self.lib.vslam_get_point_tracks(self.h, stream, found.ctypes.data, searched.ctypes.data, level.ctypes.data,
                                               subpix.ctypes.data, vfound.ctypes.data, image.ctypes.data, n))
        return {"found": found, "searched": searched, "level": level, "subpix": subpix, "vfound": vfound, "image": image}

    def points(self, stream):
        n = self.state(stream).n_points
        pos = np.zeros((n, 3)); bad, nin, nout = (np.zeros(n, np.int32) for _ in range(3))
        _check(self.lib.vslam_get_points(self.h, stream, pos.ctypes.data, bad.ctypes.data, nin.ctypes.data, nout.ctypes.data, n))
        return {"pos": pos, "bad": bad, "n_in": nin, "n_out": nout}

    def keyframe_corners(self, stream, kf, level, cap=16384):
        out = np.zeros(cap, np.uint32); n = C.c_int(0)
        _check(self.lib.vslam_get_keyframe_corners(self.h, stream, kf, level, out.ctypes.data, cap, C.byref(n)))
        return out[:min(n.value, cap)].copy()

    def read_sbi(self, stream):
        hs, ws = self.params.height // 16, self.params.width // 16
        small = np.zeros((hs, ws), np.uint8); tmpl = np.zeros((hs, ws), np.float32); rot = np.zeros(8)
        _check(self.lib.vslam_read_sbi(self.h, stream, small.ctypes.data, tmpl.ctypes.data, rot.ctypes.data))
        return small, tmpl, rot[:6].copy(), float(rot[6])

    def make_keyframe_rest(self, min_score=70.0):
        _check(self.lib.vslam_make_keyframe_rest(self.h, float(min_score)))

    def thin_candidates(self, keyframe=-1):
        _check(self.lib.vslam_thin_candidates(self.h, int(keyframe)))

    def read_candidates(self, stream, level):
        n = C.c_int(0)
        _check(self.lib.vslam_read_candidates(self.h, stream, level, None, None, 0, C.byref(n)))
        pos = np.zeros(max(n.value, 1), np.uint32); sc = np.zeros(max(n.value, 1), np.float64)
        _check(self.lib.vslam_read_candidates(self.h, stream, level, pos.ctypes.data, sc.ctypes.data, len(pos), C.byref(n)))
        return pos[:n.value], sc[:n.value]

    def bundle_stats(self, stream):
        """Sizes of the last assembled bundle-adjustment problem of the stream."""
        o = np.zeros(6, np.int32)
        _check(self.lib.vslam_get_bundle_stats(self.h, stream, o.ctypes.data))
        return dict(zip(("cams", "free_cams", "points", "meas", "trials", "accepted"), (int(x) for x in o)))

    def mapmaker_idle_job(self, job):
        _check(self.lib.vslam_mapmaker_idle_job(self.h, job))

    def load_map_dump(self, stream, directory):
        _check(self.lib.vslam_load_map(self.h, stream, str(directory).encode()))

    def init_from_stereo(self, gray_first, gray_second, matches_xyxy):
        """MapMaker::InitFromStereo with the caller's frames and matches (vslam_init_from_stereo) -> (ok, pose12)"""
        a = np.ascontiguousarray(gray_first, np.uint8); b = np.ascontiguousarray(gray_second, np.uint8)
        m = np.ascontiguousarray(matches_xyxy, np.int32).reshape(-1, 4)
        pose = np.zeros(12)
        rc = _check(self.lib.vslam_init_from_stereo(self.h, a.ctypes.data, b.ctypes.data, a.shape[1], len(m), m.ctypes.data, pose.ctypes.data))
        return bool(rc), pose

    def press_spacebar(self, stream=-1):
        _check(self.lib.vslam_press_spacebar(self.h, stream))

    def set_boot_seed(self, stream, seed):
        _check(self.lib.vslam_set_boot_seed(self.h, stream, seed))

    def init_info(self, stream):
        o = np.zeros(6, np.int32)
        _check(self.lib.vslam_get_init_info(self.h, stream, o.ctypes.data))
        return dict(zip(("stage", "trails", "init_ok", "hom_inliers", "stereo_points", "map_good"), (int(x) for x in o)))

    def trails(self, stream):
        o = np.zeros((1000, 4), np.int32)
        n = C.c_int(0)
        _check(self.lib.vslam_get_trails(self.h, stream, o.ctypes.data, 1000, C.byref(n)))
        return o[:n.value]

    def idle_stats(self, stream):
        o = np.zeros(6, np.int32)
        _check(self.lib.vslam_get_idle_stats(self.h, stream, o.ctypes.data))
        return dict(zip(("refound_new", "refound_failed", "ba_all", "ba_recent_idle", "failure_queue", "new_queue"), (int(x) for x in o)))

    def keyframe_pose(self, stream, kf):
        p = np.zeros(12)
        _check(self.lib.vslam_get_keyframe_pose(self.h, stream, kf, p.ctypes.data))
        return p

    def save_map(self, stream, directory):
        """MapMaker's "SaveMap" dump (jni/MapMaker.cc:1254-1286): <directory>/map.dump and <directory>/keyframes/<i>.info."""
        os.makedirs(os.path.join(directory, "keyframes"), exist_ok=True)
        return _check(self.lib.vslam_save_map(self.h, stream, os.fsencode(directory)))

    def keyframe_meas(self, stream, kf, cap=8192):
        pt, level, source = (np.zeros(cap, np.int32) for _ in range(3))
        root = np.zeros((cap, 2))
        n = _check(self.lib.vslam_get_keyframe_measurements(self.h, stream, kf, pt.ctypes.data, level.ctypes.data, root.ctypes.data, source.ctypes.data, cap))
        return {"pt": pt[:n], "level": level[:n], "root": root[:n], "source": source[:n]}

    def template(self, stream, pt):
        P = self.params.patch_size
        t = np.zeros(P * P, np.uint8)
        s, sq, bad = C.c_int(0), C.c_int(0), C.c_int(0)
        have = _check(self.lib.vslam_get_template(self.h, stream, pt, t.ctypes.data, C.byref(s), C.byref(sq), C.byref(bad)))
        return {"tmpl": t.reshape(P, P), "sum": s.value, "sumsq": sq.value, "bad": bad.value, "have": have}

    def templates(self, stream, n=None):
        """cached warped templates of the first n map points: (tmpl [n, P, P], sum, sumsq, bad, have)"""
        n = self.state(stream).n_points if n is None else n
        P = self.params.patch_size
        t = np.zeros((n, P, P), np.uint8)
        s, sq, bad, have = (np.zeros(n, np.int32) for _ in range(4))
        _check(self.lib.vslam_get_templates(self.h, stream, 0, n, t.ctypes.data, s.ctypes.data, sq.ctypes.data, bad.ctypes.data, have.ctypes.data))
        return {"tmpl": t, "sum": s, "sumsq": sq, "bad": bad, "have": have}

    def profile_begin(self, max_frames):
        _check(self.lib.vslam_profile_begin(self.h, max_frames))

    def profile_end(self):
        """-> ({stage name: summed ms}, n_frames) measured with HIP events on the system's stream"""
        ms = np.zeros(N_STAGES)
        n = C.c_int(0)
        _check(self.lib.vslam_profile_end(self.h, ms.ctypes.data, C.byref(n)))
        return {self.lib.vslam_stage_name(k).decode(): float(ms[k]) for k in range(N_STAGES)}, n.value

    def profile_launches(self):
        """-> {stage name: launches recorded by the last profile_begin/profile_end pair}"""
        c = np.zeros(N_STAGES, np.int32)
        _check(self.lib.vslam_profile_launches(self.h, c.ctypes.data))
        return {self.lib.vslam_stage_name(k).decode(): int(c[k]) for k in range(N_STAGES)}

    BA_STAT_KEYS = ("problems", "trials", "trials_x_meas", "trials_x_cams", "trials_x_points", "trials_x_points_x_pairs", "trials_x_6n_cubed", "launches")

    def profile_ba_stats(self):
        """-> what the k_ba_compute launches of the last profile window ran, counted on the device (vslam_profile_ba_stats)"""
        st = np.zeros(8, np.uint64)
        _check(self.lib.vslam_profile_ba_stats(self.h, st.ctypes.data))
        return {k: int(v) for k, v in zip(self.BA_STAT_KEYS, st)}

    def ba_launch_totals(self):
        """-> the device counters of every k_ba_compute launch since the system was created, summed (vslam_get_ba_launch_totals)"""
        st = np.zeros(8, np.uint64)
        _check(self.lib.vslam_get_ba_launch_totals(self.h, st.ctypes.data))
        return {k: int(v) for k, v in zip(self.BA_STAT_KEYS, st)}

    def mapmaker_timing(self):
        """-> ({"assemble": ms, "compute": ms, "writeback": ms}, counters) of the last bundle_adjust_recent / bundle_adjust_all (HIP events)"""
        ms = np.zeros(3)
        st = np.zeros(8, np.uint64)
        _check(self.lib.vslam_get_mapmaker_timing(self.h, ms.ctypes.data, st.ctypes.data))
        return {"assemble": float(ms[0]), "compute": float(ms[1]), "writeback": float(ms[2])}, {k: int(v) for k, v in zip(self.BA_STAT_KEYS, st)}

    def bundle_adjust_recent(self):
        _check(self.lib.vslam_bundle_adjust_recent(self.h))
        self.synchronize()

    def bundle_adjust_all(self):
        _check(self.lib.vslam_bundle_adjust_all(self.h))
        self.synchronize()


class Bundle:
    """Batched stand-alone Bundle (jni/Bundle.h:111-121): n_problems independent problems, one launch."""

    def __init__(self, params, n_problems=1, max_cameras=16, max_points=2048, max_meas=32768):
        self.lib = load_library()
        self.h = _vp()
        _check(self.lib.vslam_bundle_create(C.byref(params), n_problems, max_cameras, max_points, max_meas, C.byref(self.h)))
        self.n = n_problems
        self.ncam = [0] * n_problems
        self.npt = [0] * n_problems

    def close(self):
        if self.h:
            self.lib.vslam_bundle_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_camera(self, pose12, fixed, problem=0):
        self.ncam[problem] += 1
        return _check(self.lib.vslam_bundle_add_camera(self.h, problem, _f64(pose12).ctypes.data, int(fixed)))

    def add_point(self, pos, problem=0):
        self.npt[problem] += 1
        return _check(self.lib.vslam_bundle_add_point(self.h, problem, _f64(pos).ctypes.data))

    def add_meas(self, cam, pt, pos2, sigma2, problem=0):
        _check(self.lib.vslam_bundle_add_meas(self.h, problem, cam, pt, _f64(pos2).ctypes.data, float(sigma2)))

    def set_problem(self, problem, cams, fixed, pts, meas_cam, meas_pt, meas_xy, meas_sigma2):
        """the whole problem at once (vslam_bundle_set_problem): cams [n][12], fixed [n], pts [m][3], measurements in AddMeas order"""
        cams = np.ascontiguousarray(cams, np.float64); fx = np.ascontiguousarray(fixed, np.int32); pts = np.ascontiguousarray(pts, np.float64)
        mc = np.ascontiguousarray(meas_cam, np.int32); mp = np.ascontiguousarray(meas_pt, np.int32)
        xy = np.ascontiguousarray(meas_xy, np.float64); s2 = np.ascontiguousarray(meas_sigma2, np.float64)
        _check(self.lib.vslam_bundle_set_problem(self.h, problem, len(cams), cams.ctypes.data, fx.ctypes.data, len(pts), pts.ctypes.data,
                                                 len(mc), mc.ctypes.data, mp.ctypes.data, xy.ctypes.data, s2.ctypes.data))
        self.ncam[problem], self.npt[problem] = len(cams), len(pts)

    def timing(self):
        """-> (HIP-event ms of the last compute launch, the launch's device counters as System.profile_ba_stats)"""
        ms = C.c_double(0.0)
        st = np.zeros(8, np.uint64)
        _check(self.lib.vslam_bundle_get_timing(self.h, C.byref(ms), st.ctypes.data))
        return ms.value, {k: int(v) for k, v in zip(System.BA_STAT_KEYS, st)}

    def compute(self, sync=True):
        _check(self.lib.vslam_bundle_compute(self.h))
        if sync:
            _check(self.lib.vslam_bundle_synchronize(self.h))

    def synchronize(self):
        _check(self.lib.vslam_bundle_synchronize(self.h))

    def result(self, problem=0):
        acc, conv = C.c_int(0), C.c_int(0)
        s2, lam, tr = C.c_double(0), C.c_double(0), C.c_longlong(0)
        _check(self.lib.vslam_bundle_get_result(self.h, problem, C.byref(acc), C.byref(conv), C.byref(s2), C.byref(lam), C.byref(tr)))
        return {"accepted": acc.value, "converged": bool(conv.value), "sigma2": s2.value, "lambda": lam.value, "trials": tr.value}

    def cameras(self, problem=0):
        out = np.zeros((self.ncam[problem], 12))
        for i in range(self.ncam[problem]):
            _check(self.lib.vslam_bundle_get_camera(self.h, problem, i, out[i].ctypes.data))
        return out

    def points(self, problem=0):
        out = np.zeros((self.npt[problem], 3))
        for i in range(self.npt[problem]):
            _check(self.lib.vslam_bundle_get_point(self.h, problem, i, out[i].ctypes.data))
        return out

    def outlier_meas(self, problem=0, cap=65536):
        a = np.zeros((cap, 2), np.int32)
        n = _check(self.lib.vslam_bundle_get_outlier_meas(self.h, problem, a.ctypes.data, cap))
        return a[:n]

    def outlier_points(self, problem=0, cap=65536):
        a = np.zeros(cap, np.int32)
        n = _check(self.lib.vslam_bundle_get_outlier_points(self.h, problem, a.ctypes.data, cap))
        return a[:n]
