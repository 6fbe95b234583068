"""ctypes binding of the CPU oracle (oracle/_build/libptam_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  Nothing under visualslam_android_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libptam_oracle.so")
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
