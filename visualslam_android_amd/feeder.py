"""ctypes binding of include/vslam_feeder.h + the ground-truth map builder used by tests and bench.py.

The feeder replaces the reference's camera/UI plumbing and map bootstrap (both out of scope); it is host code.
"""
import ctypes as C

import numpy as np

from . import capi

_bound = False
FEEDER_SYMBOLS = {
    "vslam_feeder_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "vslam_feeder_destroy": (C.c_int, [C.c_void_p]),
    "vslam_feeder_pose": (C.c_int, [C.c_void_p, C.c_double, C.c_void_p]),
    "vslam_feeder_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "vslam_feeder_render_pose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    "vslam_feeder_make_point": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vslam_feeder_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]),
}
REF_CAM = (0.841906, 1.10893, 0.505171, 0.470265, -0.0133843)  # jni/ATANCamera.cc:20-24


def _lib():
    global _bound
    lib = capi.load_library()
    if not _bound:
        for name, (res, args) in FEEDER_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return lib


class Feeder:
    def __init__(self, width, height, seed=1234, cam=REF_CAM, noise=2):
        self.w, self.h = width, height
        self.lib = _lib()
        self.h_ = C.c_void_p()
        cam_a = (C.c_double * 5)(*cam)
        rc = self.lib.vslam_feeder_create(width, height, cam_a, seed, noise, C.byref(self.h_))
        if rc:
            raise capi.VslamError("feeder_create failed")

    def close(self):
        if self.h_:
            self.lib.vslam_feeder_destroy(self.h_)
            self.h_ = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pose(self, t):
        p = np.empty(12)
        self.lib.vslam_feeder_pose(self.h_, float(t), p.ctypes.data)
        return p

    def render(self, first, count, threads=8):
        out = np.empty((count, self.h, self.w), np.uint8)
        self.lib.vslam_feeder_render(self.h_, first, count, out.ctypes.data, self.w, self.w * self.h, threads)
        return out

    def render_pose(self, pose12, key=0):
        out = np.empty((self.h, self.w), np.uint8)
        p = np.ascontiguousarray(pose12, np.float64)
        self.lib.vslam_feeder_render_pose(self.h_, p.ctypes.data, key, out.ctypes.data, self.w)
        return out

    def make_point(self, pose12, level, cx, cy):
        p = np.ascontiguousarray(pose12, np.float64)
        pos, r, d = np.empty(3), np.empty(3), np.empty(3)
        rc = self.lib.vslam_feeder_make_point(self.h_, p.ctypes.data, level, int(cx), int(cy), pos.ctypes.data, r.ctypes.data, d.ctypes.data)
        return (pos, r, d) if rc == 0 else None

    def project(self, pose12, pos, border=0):
        p = np.ascontiguousarray(pose12, np.float64)
        x = np.ascontiguousarray(pos, np.float64)
        im = np.empty(2)
        depth = C.c_double(0)
        ok = self.lib.vslam_feeder_project(self.h_, p.ctypes.data, x.ctypes.data, border, im.ctypes.data, C.byref(depth))
        return ok == 1, im, depth.value


def se3_perturb(pose12, rng, trans_sigma, rot_sigma):
    """Left-multiply pose by a small random SE3 (host helper for noisy initial maps)."""
    R = pose12[:9].reshape(3, 3)
    t = pose12[9:]
    w = rng.normal(0, rot_sigma, 3)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    dR = np.eye(3) + (np.sin(th) / th if th > 1e-12 else 1.0) * K + ((1 - np.cos(th)) / th ** 2 if th > 1e-12 else 0.5) * K @ K
    dt = rng.normal(0, trans_sigma, 3)
    return np.concatenate([(dR @ R).ravel(), dR @ t + dt])


def build_map(feeder, corner_fn, n_keyframes=8, kf_spacing=20, per_level=(260, 90, 32, 12), patch_border=10,
              point_noise=0.0, pose_noise=(0.0, 0.0), seed=7):
    """Ground-truth initial map from `n_keyframes` source keyframes at frames -spacing*n .. -spacing.

    corner_fn(gray) -> list of 4 arrays of packed (x | y<<16) maximal FAST corners per level (from the HIP
    front-end in bench.py / GPU tests, from the oracle in CPU tests).  Returns a dict:
      keyframes: list of {pose, fixed, image, depth_mean, depth_sigma}
      points:    list of {pos, src_kf, level, irx, iry, right, down}
      meas:      list of (kf, point, level, root_x, root_y, subpix, source)
    Points are thinned on a grid per level so no two share a cell; measurements of a point in the other
    keyframes are its exact projections (the bootstrap that would produce them is out of scope).
    """
    rng = np.random.default_rng(seed)
    times = [-kf_spacing * (n_keyframes - k) for k in range(n_keyframes)]
    kfs, points, meas = [], [], []
    true_poses = []
    for k, t in enumerate(times):
        pose = feeder.pose(t)
        img = feeder.render_pose(pose, key=1000 + k)
        true_poses.append(pose)
        kfs.append({"pose": pose.copy(), "fixed": k == 0, "image": img, "depth_mean": 1.0, "depth_sigma": 0.1})
    for k in range(n_keyframes):
        corners = corner_fn(kfs[k]["image"])
        for level in range(4):
            c = np.asarray(corners[level], np.uint32)
            if len(c) == 0:
                continue
            xs = (c & 0xFFFF).astype(np.int64)
            ys = (c >> 16).astype(np.int64)
            lw, lh = feeder.w >> level, feeder.h >> level
            ok = (xs >= patch_border) & (ys >= patch_border) & (xs < lw - patch_border) & (ys < lh - patch_border)
            xs, ys = xs[ok], ys[ok]
            # one corner per grid cell, cells visited in raster order -> deterministic thinning
            cell = max(4, int(np.sqrt(lw * lh / max(1, per_level[level] * 2))))
            seen = set()
            n_added = 0
            for x, y in zip(xs, ys):
                key = (x // cell, y // cell)
                if key in seen:
                    continue
                seen.add(key)
                mp = feeder.make_point(true_poses[k], level, x, y)
                if mp is None:
                    continue
                pos, right, down = mp
                pid = len(points)
                points.append({"pos": pos, "src_kf": k, "level": level, "irx": int(x), "iry": int(y), "right": right, "down": down})
                s = 1 << level
                meas.append((k, pid, level, (x + 0.5) * s - 0.5, (y + 0.5) * s - 0.5, 1, 2))   # SRC_ROOT
                for k2 in range(n_keyframes):
                    if k2 == k:
                        continue
                    inside, im, _ = feeder.project(true_poses[k2], pos, border=12 * s)
                    if inside:
                        meas.append((k2, pid, level, float(im[0]), float(im[1]), 1, 0))     # SRC_TRACKER
                n_added += 1
                if n_added >= per_level[level]:
                    break
    # scene depth per keyframe (MapMaker::RefreshSceneDepth, jni/MapMaker.cc:1236-1252)
    for k in range(n_keyframes):
        zs = []
        for (kk, pid, *_r) in meas:
            if kk == k:
                zs.append(feeder.project(true_poses[k], points[pid]["pos"])[2])
        zs = np.array(zs)
        kfs[k]["depth_mean"] = float(zs.mean())
        kfs[k]["depth_sigma"] = float(np.sqrt(max(0.0, (zs ** 2).mean() - zs.mean() ** 2)))
    if point_noise > 0:
        for p in points:
            p["pos"] = p["pos"] + rng.normal(0, point_noise, 3)
    if pose_noise[0] > 0 or pose_noise[1] > 0:
        for k in range(1, n_keyframes):
            kfs[k]["pose"] = se3_perturb(kfs[k]["pose"], rng, pose_noise[0], pose_noise[1])
    return {"keyframes": kfs, "points": points, "meas": meas, "times": times}
