"""ctypes binding of the CPU oracle (oracle/_build/libptam_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under visualslam_android_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ORC_ORACLE_VARIANT=stdlibm selects the variant built on glibc's transcendentals instead of the product's libm-free ones (one process
# loads one variant; tests/test_oracle_tracker.py runs the variant in a child process)
LIB = os.path.join(HERE, "_build", "libptam_oracle_stdlibm.so" if os.environ.get("ORC_ORACLE_VARIANT") == "stdlibm" else "libptam_oracle.so")
REF_MEST = os.path.join(HERE, "_ref", "libref_mestimator.so")
LEVELS = 4
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_shi_tomasi.restype = C.c_double
        _lib.orc_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def halfsample(img):
    h, w = img.shape
    img = np.ascontiguousarray(img)
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().orc_halfsample(_p(img), w, h, w, _p(out), w // 2)
    return out


def fast10(img, thr, cap=None):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    cap = cap or w * h
    out = np.empty(cap, np.uint32)
    n = lib().orc_fast10(_p(img), w, h, w, int(thr), _p(out), cap)
    return out[:min(n, cap)].copy()


def row_lut(corners, h):
    corners = np.ascontiguousarray(corners, np.uint32)
    lut = np.empty(h, np.int32)
    lib().orc_row_lut(_p(corners), len(corners), h, _p(lut))
    return lut


def fast_score(img, corners, barrier):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    corners = np.ascontiguousarray(corners, np.uint32)
    sc = np.empty(len(corners), np.int32)
    lib().orc_fast_score(_p(img), w, h, w, _p(corners), len(corners), int(barrier), _p(sc))
    return sc


def nonmax(corners, scores, quirk=False):
    corners = np.ascontiguousarray(corners, np.uint32)
    scores = np.ascontiguousarray(scores, np.int32)
    out = np.empty(max(len(corners), 1), np.uint32)
    n = lib().orc_nonmax(_p(corners), _p(scores), len(corners), int(bool(quirk)), _p(out))
    return out[:n].copy()


def candidates(img, maxcorners, min_score=70.0, border=10):
    """MakeKeyFrame_Rest candidate loop on one level image -> (packed positions, Shi-Tomasi scores)."""
    img = np.ascontiguousarray(img)
    mc = np.ascontiguousarray(maxcorners, np.uint32)
    pos = np.empty(max(len(mc), 1), np.uint32); sc = np.empty(max(len(mc), 1), np.float64)
    n = lib().orc_candidates(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(mc), len(mc), float(min_score), int(border), _p(pos), _p(sc), len(pos))
    return pos[:n].copy(), sc[:n].copy()


def thin_candidates(pos, score, level, meas_root, meas_level):
    pos = np.ascontiguousarray(pos, np.uint32); score = np.ascontiguousarray(score, np.float64)
    mr = np.ascontiguousarray(meas_root, np.float64).reshape(-1, 2); ml = np.ascontiguousarray(meas_level, np.int32)
    op = np.empty(max(len(pos), 1), np.uint32); os_ = np.empty(max(len(pos), 1), np.float64)
    n = lib().orc_thin_candidates(_p(pos), _p(score), len(pos), int(level), _p(mr), _p(ml), len(ml), _p(op), _p(os_))
    return op[:n].copy(), os_[:n].copy()


def reproject_point(AfromB12, v2A, v2B):
    """MapMaker::ReprojectPoint: 3-D point in frame B from its z = 1 plane projections in frames A and B."""
    T = (C.c_double * 12)(*AfromB12); a = (C.c_double * 2)(*v2A); b = (C.c_double * 2)(*v2B); out = (C.c_double * 3)()
    lib().orc_reproject_point(T, a, b, out)
    return np.array(out[:])


def sbi_make(level3, blur=0.75):
    """SmallBlurryImage::MakeFromKF on a level-3 image -> (small u8 image, zero-mean blurred fp32 template)."""
    l3 = np.ascontiguousarray(level3, np.uint8)
    h3, w3 = l3.shape
    small = np.empty((h3 // 2, w3 // 2), np.uint8); tmpl = np.empty((h3 // 2, w3 // 2), np.float32)
    lib().orc_sbi_make(_p(l3), w3, h3, C.c_double(blur), _p(small), _p(tmpl))
    return small, tmpl


def sbi_rotation(cur_l3, last_l3, cam5, quirks=0, blur=0.75):
    """Tracker::CalcSBIRotation between two level-3 images -> (6-vector ln of the SE3 adjustment, final ESM score)."""
    a = np.ascontiguousarray(cur_l3, np.uint8); b = np.ascontiguousarray(last_l3, np.uint8)
    cam = (C.c_double * 5)(*cam5); out = (C.c_double * 6)(); score = C.c_double(0)
    lib().orc_sbi_rotation(_p(a), _p(b), a.shape[1], a.shape[0], C.c_double(blur), cam, int(quirks), out, C.byref(score))
    return np.array(out[:]), score.value


def shi_tomasi(img, nsize, px, py):
    img = np.ascontiguousarray(img)
    return lib().orc_shi_tomasi(_p(img), img.shape[1], nsize, px, py)


def make_keyframe_lite(gray, thr=(10, 15, 15, 10), cap=None):
    """Returns list of (level image, corners, lut) for the 4 levels."""
    gray = np.ascontiguousarray(gray)
    h, w = gray.shape
    cap = cap or w * h
    imgs = [np.empty((h >> l, w >> l), np.uint8) for l in range(LEVELS)]
    cors = [np.empty(cap, np.uint32) for _ in range(LEVELS)]
    luts = [np.empty(h >> l, np.int32) for l in range(LEVELS)]
    nc = (C.c_int * LEVELS)()
    thr_a = (C.c_int * LEVELS)(*thr)
    pi = (C.c_void_p * LEVELS)(*[a.ctypes.data for a in imgs])
    pc = (C.c_void_p * LEVELS)(*[a.ctypes.data for a in cors])
    pl = (C.c_void_p * LEVELS)(*[a.ctypes.data for a in luts])
    lib().orc_make_keyframe_lite(_p(gray), w, h, w, thr_a, pi, pc, cap, nc, pl)
    return [(imgs[l], cors[l][:nc[l]].copy(), luts[l]) for l in range(LEVELS)]


# ---- verbatim-compiled reference pieces (oracle/_ref) -------------------------------------------
_ref = None


def ref_mestimator():
    """jni/MEstimator.h compiled verbatim (None if neither the reference nor a prebuilt _ref exists)."""
    global _ref
    if _ref is None and os.path.exists(REF_MEST):
        _ref = C.CDLL(REF_MEST)
        for f in ("ref_find_sigma_squared", "ref_weight", "ref_sqrt_weight", "ref_objective"):
            getattr(_ref, f).restype = C.c_double
        _ref.ref_find_sigma_squared.argtypes = [C.c_int, C.c_void_p, C.c_int]
        for f in ("ref_weight", "ref_sqrt_weight", "ref_objective"):
            getattr(_ref, f).argtypes = [C.c_int, C.c_double, C.c_double]
    return _ref


# ---- whole-path oracle ------------------------------------------------------------------------------
class OrcParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("patch_size", C.c_int), ("thr", C.c_int * 4),
                ("nonmax_barrier", C.c_int), ("max_patches", C.c_int), ("coarse_min", C.c_int), ("coarse_max", C.c_int),
                ("coarse_range", C.c_int), ("coarse_subpix_its", C.c_int), ("coarse_disabled", C.c_int),
                ("coarse_min_vel", C.c_double), ("fine_subpix_its", C.c_int), ("wls_prior", C.c_double),
                ("min_frames_between_kf", C.c_int), ("max_kf_dist_wiggle_mult", C.c_double), ("wiggle_scale", C.c_double),
                ("ba_max_iterations", C.c_int), ("ba_convergence_limit", C.c_double), ("ba_min_tukey_sigma", C.c_double),
                ("ba_window", C.c_int), ("ba_min_keyframes", C.c_int), ("cam", C.c_double * 5), ("quirks", C.c_int),
                ("ba_delay_frames", C.c_int), ("use_sbi", C.c_int), ("grow_map", C.c_int), ("idle_iterations", C.c_int)]


class TrackState(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("velocity", C.c_double * 6), ("msd_velocity", C.c_double),
                ("depth_mean", C.c_double), ("depth_sigma", C.c_double), ("attempted", C.c_int * 4), ("found", C.c_int * 4),
                ("quality", C.c_int), ("lost_frames", C.c_int), ("frame", C.c_int), ("did_coarse", C.c_int),
                ("kf_added", C.c_int), ("n_keyframes", C.c_int), ("n_points", C.c_int), ("ba_accepted", C.c_int),
                ("n_zmssd", C.c_longlong), ("n_ba_trials", C.c_longlong)]


def params_from_vslam(vp):
    """Build orc_params from a visualslam_android_amd.capi.Params (same tunables)."""
    p = OrcParams()
    p.width, p.height, p.patch_size = vp.width, vp.height, vp.patch_size
    for i in range(4):
        p.thr[i] = vp.fast_threshold[i]
    p.nonmax_barrier = vp.nonmax_barrier
    p.max_patches = vp.max_patches_per_frame
    for f in ("coarse_min", "coarse_max", "coarse_range", "coarse_subpix_its", "coarse_disabled", "coarse_min_vel",
              "fine_subpix_its", "wls_prior", "min_frames_between_kf", "max_kf_dist_wiggle_mult", "wiggle_scale",
              "ba_max_iterations", "ba_convergence_limit", "ba_min_tukey_sigma", "ba_window", "ba_min_keyframes", "quirks",
              "ba_delay_frames", "use_sbi", "grow_map", "idle_iterations"):
        setattr(p, f, getattr(vp, f))
    for i in range(5):
        p.cam[i] = vp.cam[i]
    return p


def _d(a):
    return np.ascontiguousarray(a, np.float64)


class OracleSystem:
    """One sequence of the CPU oracle (Tracker + MapMaker BA driver)."""

    def __init__(self, params):
        L = lib()
        L.orc_sys_create.restype = C.c_void_p
        for f in ("orc_sys_destroy", "orc_sys_add_meas", "orc_sys_set_map_good", "orc_sys_set_pose", "orc_sys_set_velocity",
                  "orc_sys_track_frame", "orc_sys_get_state", "orc_sys_get_keyframe_pose", "orc_sys_frame_begin", "orc_sys_search_stage",
                  "orc_sys_pose_stage", "orc_sys_frame_end", "orc_sys_idle_iteration", "orc_sys_idle_job", "orc_sys_get_idle_stats",
                  "orc_sys_press_spacebar", "orc_sys_set_last_keyframe_dropped", "orc_sys_set_boot_seed", "orc_sys_get_init_info"):
            getattr(L, f).restype = None
        self.L = L
        self.p = params
        self.h = C.c_void_p(L.orc_sys_create(C.byref(params)))

    def close(self):
        if self.h:
            self.L.orc_sys_destroy(self.h)
            self.h = None

    def add_keyframe(self, pose12, fixed, gray, dmean, dsigma):
        g = np.ascontiguousarray(gray, np.uint8)
        return self.L.orc_sys_add_keyframe(self.h, _p(_d(pose12)), int(fixed), _p(g), g.shape[1], C.c_double(dmean), C.c_double(dsigma))

    def add_point(self, pos, src_kf, level, irx, iry, right, down):
        return self.L.orc_sys_add_point(self.h, _p(_d(pos)), src_kf, level, irx, iry, _p(_d(right)), _p(_d(down)))

    def add_meas(self, kf, pt, level, rx, ry, subpix, source):
        self.L.orc_sys_add_meas(self.h, kf, pt, level, _p(_d([rx, ry])), int(subpix), int(source))

    def load_map(self, m):
        for k in m["keyframes"]:
            self.add_keyframe(k["pose"], k["fixed"], k["image"], k["depth_mean"], k["depth_sigma"])
        for q in m["points"]:
            self.add_point(q["pos"], q["src_kf"], q["level"], q["irx"], q["iry"], q["right"], q["down"])
        for (kf, pt, level, rx, ry, sp, src) in m["meas"]:
            self.add_meas(kf, pt, level, rx, ry, sp, src)
        self.L.orc_sys_set_map_good(self.h)

    def set_pose(self, pose12):
        self.L.orc_sys_set_pose(self.h, _p(_d(pose12)))

    def set_velocity(self, v6):
        self.L.orc_sys_set_velocity(self.h, _p(_d(v6)))

    def track_frame(self, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        self.L.orc_sys_track_frame(self.h, _p(g), g.shape[1])

    # TrackFrame in stages (jni/Tracker.cc:76-146 cut where the C ABI's stage entry points cut it)
    def frame_begin(self, gray):
        g = np.ascontiguousarray(gray, np.uint8)
        self.L.orc_sys_frame_begin(self.h, _p(g), g.shape[1])

    def search_stage(self, stage):
        self.L.orc_sys_search_stage(self.h, int(stage))

    def pose_stage(self, stage):
        self.L.orc_sys_pose_stage(self.h, int(stage))

    def frame_end(self):
        self.L.orc_sys_frame_end(self.h)

    def idle_iteration(self):
        self.L.orc_sys_idle_iteration(self.h)

    def idle_job(self, job):
        self.L.orc_sys_idle_job(self.h, job)

    def set_last_keyframe_dropped(self, frame):
        self.L.orc_sys_set_last_keyframe_dropped(self.h, C.c_int(frame))

    def init_from_stereo(self, gray_first, gray_second, matches_xyxy):
        a = np.ascontiguousarray(gray_first, np.uint8); b = np.ascontiguousarray(gray_second, np.uint8)
        m = np.ascontiguousarray(matches_xyxy, np.int32).reshape(-1, 4)
        pose = np.zeros(12)
        rc = self.L.orc_sys_init_from_stereo(self.h, _p(a), _p(b), a.shape[1], _p(m), len(m), _p(pose))
        return rc == 1, pose

    def press_spacebar(self):
        self.L.orc_sys_press_spacebar(self.h)

    def set_boot_seed(self, seed):
        self.L.orc_sys_set_boot_seed(self.h, C.c_uint(seed))

    def init_info(self):
        o = np.zeros(6, np.int32)
        self.L.orc_sys_get_init_info(self.h, _p(o))
        return dict(zip(("stage", "trails", "init_ok", "hom_inliers", "stereo_points", "map_good"), (int(x) for x in o)))

    def trails(self):
        o = np.zeros((1000, 4), np.int32)
        self.L.orc_sys_get_trails.restype = C.c_int
        n = self.L.orc_sys_get_trails(self.h, _p(o), 1000)
        return o[:n]

    def idle_stats(self):
        o = np.zeros(6, np.int32)
        self.L.orc_sys_get_idle_stats(self.h, _p(o))
        return dict(zip(("refound_new", "refound_failed", "ba_all", "ba_recent_idle", "failure_queue", "new_queue"), (int(x) for x in o)))

    def state(self):
        s = TrackState()
        self.L.orc_sys_get_state(self.h, C.byref(s))
        return s

    def grow_log(self, cap=65536):
        out = np.zeros((cap, 3), np.int32)
        n = self.L.orc_sys_get_grow_log(self.h, _p(out), cap)
        return out[:n]

    def point_tracks(self):
        n = self.state().n_points
        found, searched, level, subpix = (np.zeros(n, np.int32) for _ in range(4))
        vfound, image = np.zeros((n, 2)), np.zeros((n, 2))
        self.L.orc_sys_get_point_tracks(self.h, _p(found), _p(searched), _p(level), _p(subpix), _p(vfound), _p(image), n)
        return {"found": found, "searched": searched, "level": level, "subpix": subpix, "vfound": vfound, "image": image}

    def points(self):
        n = self.state().n_points
        pos = np.zeros((n, 3)); bad, nin, nout = (np.zeros(n, np.int32) for _ in range(3))
        self.L.orc_sys_get_points(self.h, _p(pos), _p(bad), _p(nin), _p(nout), n)
        return {"pos": pos, "bad": bad, "n_in": nin, "n_out": nout}

    def keyframe_pose(self, kf):
        p = np.zeros(12)
        self.L.orc_sys_get_keyframe_pose(self.h, kf, _p(p))
        return p

    def keyframe_meas(self, kf, cap=8192):
        pt, level, source = (np.zeros(cap, np.int32) for _ in range(3))
        root = np.zeros((cap, 2))
        n = self.L.orc_sys_get_keyframe_meas(self.h, kf, _p(pt), _p(level), _p(root), _p(source), cap)
        return {"pt": pt[:n], "level": level[:n], "root": root[:n], "source": source[:n]}

    def template(self, pt):
        P = self.p.patch_size
        t = np.zeros(P * P, np.uint8)
        s, sq, bad = C.c_int(0), C.c_int(0), C.c_int(0)
        have = self.L.orc_sys_get_template(self.h, pt, _p(t), C.byref(s), C.byref(sq), C.byref(bad))
        return {"tmpl": t.reshape(P, P), "sum": s.value, "sumsq": sq.value, "bad": bad.value, "have": have}

    def templates(self, n=None):
        """cached warped templates of the first n map points, like capi.System.templates"""
        n = self.state().n_points if n is None else n
        P = self.p.patch_size
        t = np.zeros((n, P, P), np.uint8)
        s, sq, bad, have = (np.zeros(n, np.int32) for _ in range(4))
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        for i in range(n):
            have[i] = self.L.orc_sys_get_template(self.h, i, _p(t[i]), C.byref(a), C.byref(b), C.byref(c))
            s[i], sq[i], bad[i] = a.value, b.value, c.value
        return {"tmpl": t, "sum": s, "sumsq": sq, "bad": bad, "have": have}

    def bundle_adjust_recent(self):
        return self.L.orc_sys_bundle_adjust_recent(self.h)

    def bundle_adjust_all(self):
        return self.L.orc_sys_bundle_adjust_all(self.h)


class OracleBundle:
    """jni/Bundle.h:111-121 surface of the oracle."""

    def __init__(self, cam5, w, h, quirks=0, max_iterations=20, convergence_limit=1e-6, min_sigma=0.4):
        L = lib()
        L.orc_ba_create.restype = C.c_void_p
        for f in ("orc_ba_destroy", "orc_ba_add_meas", "orc_ba_get_camera", "orc_ba_get_point", "orc_ba_get_stats"):
            getattr(L, f).restype = None
        self.L = L
        self.h = C.c_void_p(L.orc_ba_create(_p(_d(cam5)), w, h, quirks, max_iterations, C.c_double(convergence_limit), C.c_double(min_sigma)))
        self.ncam = self.npt = 0

    def close(self):
        if self.h:
            self.L.orc_ba_destroy(self.h); self.h = None

    def add_camera(self, pose12, fixed):
        self.ncam += 1
        return self.L.orc_ba_add_camera(self.h, _p(_d(pose12)), int(fixed))

    def add_point(self, pos):
        self.npt += 1
        return self.L.orc_ba_add_point(self.h, _p(_d(pos)))

    def add_meas(self, cam, pt, pos2, sigma2):
        self.L.orc_ba_add_meas(self.h, cam, pt, _p(_d(pos2)), C.c_double(sigma2))

    def compute(self):
        return self.L.orc_ba_compute(self.h)

    def cameras(self):
        out = np.zeros((self.ncam, 12))
        for i in range(self.ncam):
            self.L.orc_ba_get_camera(self.h, i, _p(out[i]))
        return out

    def points(self):
        out = np.zeros((self.npt, 3))
        for i in range(self.npt):
            self.L.orc_ba_get_point(self.h, i, _p(out[i]))
        return out

    def converged(self):
        return bool(self.L.orc_ba_converged(self.h))

    def outlier_meas(self, cap=65536):
        a = np.zeros((cap, 2), np.int32)
        n = self.L.orc_ba_get_outlier_meas(self.h, _p(a), cap)
        return a[:n]

    def outlier_points(self, cap=65536):
        a = np.zeros(cap, np.int32)
        n = self.L.orc_ba_get_outlier_points(self.h, _p(a), cap)
        return a[:n]

    def stats(self):
        s2, lam, tr = C.c_double(0), C.c_double(0), C.c_longlong(0)
        self.L.orc_ba_get_stats(self.h, C.byref(s2), C.byref(lam), C.byref(tr))
        return s2.value, lam.value, tr.value


def se3_exp(mu):
    lib().orc_se3_exp.restype = None
    out = np.zeros(12)
    lib().orc_se3_exp(_p(_d(mu)), _p(out))
    return out


def se3_ln(pose12):
    lib().orc_se3_ln.restype = None
    out = np.zeros(6)
    lib().orc_se3_ln(_p(_d(pose12)), _p(out))
    return out


def cam_project(cam5, w, h, cx, cy, quirks=0):
    L = lib(); L.orc_cam_project.restype = None
    im, d = np.zeros(2), np.zeros(4)
    inv, lr = C.c_int(0), C.c_double(0)
    L.orc_cam_project(_p(_d(cam5)), w, h, quirks, C.c_double(cx), C.c_double(cy), _p(im), _p(d), C.byref(inv), C.byref(lr))
    return im, d.reshape(2, 2), inv.value, lr.value


def cam_unproject(cam5, w, h, ix, iy):
    L = lib(); L.orc_cam_unproject.restype = None
    out = np.zeros(2)
    L.orc_cam_unproject(_p(_d(cam5)), w, h, C.c_double(ix), C.c_double(iy), _p(out))
    return out


def find_sigma_squared(est, v):
    L = lib(); L.orc_find_sigma_squared.restype = C.c_double
    v = _d(v)
    return L.orc_find_sigma_squared(est, _p(v), len(v))


def mest(fn, est, e2, s2):
    L = lib(); f = getattr(L, "orc_" + fn); f.restype = C.c_double; f.argtypes = [C.c_int, C.c_double, C.c_double]
    return f(est, e2, s2)


def transform_image(img, P, M, in_orig, out_orig):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((P, P), np.uint8)
    n = lib().orc_transform_image(_p(img), img.shape[1], img.shape[0], img.shape[1], _p(out), P, _p(_d(M)), _p(_d(in_orig)), _p(_d(out_orig)))
    return out, n


def zmssd(tmpl, img, x, y):
    tmpl = np.ascontiguousarray(tmpl, np.uint8); img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_zmssd(_p(tmpl), tmpl.shape[0], _p(img), img.shape[1], img.shape[0], img.shape[1], int(x), int(y))


def subpix_refine(tmpl, img, x, y, max_its=10):
    tmpl = np.ascontiguousarray(tmpl, np.uint8); img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(2)
    ok = lib().orc_subpix_refine(_p(tmpl), tmpl.shape[0], _p(img), img.shape[1], img.shape[0], int(x), int(y), int(max_its), _p(out))
    return bool(ok), out


def minipatch_sample(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((9, 9), np.uint8)
    ok = lib().orc_minipatch_sample(_p(img), img.shape[1], img.shape[0], img.shape[1], int(x), int(y), _p(out))
    return out if ok else None


def minipatch_find(patch, img, corners, x, y, rng=10, max_ssd=100000):
    img = np.ascontiguousarray(img, np.uint8); patch = np.ascontiguousarray(patch, np.uint8)
    corners = np.ascontiguousarray(corners, np.uint32)
    pos = np.array([x, y], np.int32)
    ok = lib().orc_minipatch_find(_p(patch), _p(img), img.shape[1], img.shape[0], img.shape[1], _p(corners), len(corners), int(rng), int(max_ssd), _p(pos))
    return bool(ok), int(pos[0]), int(pos[1])


def homography_init(matches8, max_pixel_error=5.0, seed=1):
    """HomographyInit::Compute on (n, 8) matches [first xy, second xy, 2x2 pixel Jacobian]: (ok, 3x4 second-from-first, inliers)."""
    L = lib()
    m = np.ascontiguousarray(matches8, np.float64)
    out = np.zeros(12); ninl = C.c_int(0)
    L.orc_homography_init.restype = C.c_int
    ok = L.orc_homography_init(_p(m), C.c_int(len(m)), C.c_double(max_pixel_error), C.c_uint(seed), _p(out), C.byref(ninl))
    return bool(ok), out, ninl.value


def calc_plane_aligner(pos, seed=1):
    L = lib()
    p = np.ascontiguousarray(pos, np.float64)
    out = np.zeros(12)
    L.orc_calc_plane_aligner.restype = C.c_int
    ok = L.orc_calc_plane_aligner(_p(p), C.c_int(len(p)), C.c_uint(seed), _p(out))
    return bool(ok), out
