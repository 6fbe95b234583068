"""ctypes binding of include/vslam_feeder.h + the ground-truth map builder used by tests and bench.py.

The feeder replaces the reference's camera/UI plumbing and map bootstrap (both out of scope); it is host code.
"""
import threading
import ctypes as C

import numpy as np

from . import capi

_bound = None
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


_bind_lock = threading.Lock()


def _lib():
    global _bound
    lib = capi.load_library()
    with _bind_lock:                           # Feeder objects are built from thread pools (bench.py)
        if _bound is not lib:
            for name, (res, args) in FEEDER_SYMBOLS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _bound = lib
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


def _unproject_np(cam, w, h, ix, iy):
    """ATANCamera::UnProject (jni/ATANCamera.cc:149-164) vectorised (host harness code)."""
    fx, fy, cx, cy, ww = w * cam[0], h * cam[1], w * cam[2] - 0.5, h * cam[3] - 0.5, cam[4]
    dx, dy = (ix - cx) / fx, (iy - cy) / fy
    dr = np.hypot(dx, dy)
    r = dr if ww == 0.0 else np.tan(dr * ww) / (2.0 * np.tan(ww / 2.0))
    f = np.where(dr > 0.01, r / np.maximum(dr, 1e-300), 1.0)
    return dx * f, dy * f


def _project_np(cam, w, h, x, y):
    """ATANCamera::Project (jni/ATANCamera.cc:133-145) vectorised."""
    fx, fy, cx, cy, ww = w * cam[0], h * cam[1], w * cam[2] - 0.5, h * cam[3] - 0.5, cam[4]
    r = np.hypot(x, y)
    fac = np.where((r < 0.001) | (ww == 0.0), 1.0, np.arctan(r * 2.0 * np.tan(ww / 2.0)) / ww / np.maximum(r, 1e-300))
    return cx + fx * x * fac, cy + fy * y * fac


def keyframe_images(feeder, n_keyframes=8, kf_spacing=20):
    """The images of build_map's source keyframes (frames -spacing*n .. -spacing): lets a caller extract the corners of many maps'
    keyframes in one batched front-end call and hand them to build_map (corners=)."""
    times = [-kf_spacing * (n_keyframes - k) for k in range(n_keyframes)]
    return [feeder.render_pose(feeder.pose(t), key=1000 + k) for k, t in enumerate(times)]


def build_map(feeder, corner_fn, n_keyframes=8, kf_spacing=20, per_level=(260, 90, 32, 12), patch_border=10,
              point_noise=0.0, pose_noise=(0.0, 0.0), seed=7, cam=REF_CAM, images=None, corners=None):
    """Ground-truth initial map from `n_keyframes` source keyframes at frames -spacing*n .. -spacing.

    corner_fn(gray) -> list of 4 arrays of packed (x | y<<16) maximal FAST corners per level (from the HIP
    front-end in bench.py / GPU tests, from the oracle in CPU tests).  Returns a dict:
      keyframes: list of {pose, fixed, image, depth_mean, depth_sigma}
      points:    list of {pos, src_kf, level, irx, iry, right, down}
      meas:      list of (kf, point, level, root_x, root_y, subpix, source)
    Points are thinned on a grid per level (first corner of each cell in raster order); a corner is back-projected onto
    the plane z = 0 and given the patch vectors of MapMaker::AddPointEpipolar (jni/MapMaker.cc:652-684) +
    MapPoint::RefreshPixelVectors (jni/MapPoint.cc:4-29); its measurements in the other keyframes are its exact
    projections (the bootstrap that would produce them is out of scope).
    """
    rng = np.random.default_rng(seed)
    W, H = feeder.w, feeder.h
    times = [-kf_spacing * (n_keyframes - k) for k in range(n_keyframes)]
    kfs = []
    poses = [feeder.pose(t) for t in times]
    Rs = [p[:9].reshape(3, 3) for p in poses]
    ts = [p[9:] for p in poses]
    for k in range(n_keyframes):
        img = images[k] if images is not None else feeder.render_pose(poses[k], key=1000 + k)
        kfs.append({"pose": poses[k].copy(), "fixed": k == 0, "image": img, "depth_mean": 1.0, "depth_sigma": 0.1})
    depth_acc = [[] for _ in range(n_keyframes)]
    n_points = 0
    P = {k_: [] for k_ in ("pos", "right", "down", "src_kf", "level", "ir")}
    M = {k_: [] for k_ in ("kf", "pt", "level", "root", "subpix", "source")}
    for k in range(n_keyframes):
        corners_k = corners[k] if corners is not None else corner_fn(kfs[k]["image"])
        R, t = Rs[k], ts[k]
        C = -R.T @ t
        for level in range(4):
            c = np.asarray(corners_k[level], np.uint32)
            if len(c) == 0:
                continue
            xs = (c & 0xFFFF).astype(np.int64)
            ys = (c >> 16).astype(np.int64)
            lw, lh = W >> level, H >> level
            ok = (xs >= patch_border) & (ys >= patch_border) & (xs < lw - patch_border) & (ys < lh - patch_border)
            xs, ys = xs[ok], ys[ok]
            cell = max(4, int(np.sqrt(lw * lh / max(1, per_level[level] * 2))))
            key = (ys // cell) * 100000 + (xs // cell)
            _u, first = np.unique(key, return_index=True)       # first corner of every cell, raster order kept below
            first = np.sort(first)[:per_level[level]]
            xs, ys = xs[first], ys[first]
            if len(xs) == 0:
                continue
            s = 1 << level
            rx, ry = (xs + 0.5) * s - 0.5, (ys + 0.5) * s - 0.5  # LevelZeroPos
            cx, cy = _unproject_np(cam, W, H, rx, ry)
            d = np.stack([cx, cy, np.ones_like(cx)], 1) @ R      # R^T applied to rays (row vectors)
            good = d[:, 2] > 1e-9
            lam = -C[2] / np.where(good, d[:, 2], 1.0)
            pos = C[None, :] + lam[:, None] * d
            pos[:, 2] = 0.0
            # patch vectors
            def unit(ix, iy):
                ux, uy = _unproject_np(cam, W, H, ix, iy)
                v = np.stack([ux, uy, np.ones_like(ux)], 1)
                return v / np.linalg.norm(v, axis=1, keepdims=True)
            cen, rgt, dwn = unit(rx, ry), unit(rx + s, ry), unit(rx, ry + s)
            pc = pos @ R.T + t[None, :]
            camh = np.abs(pc[:, 2])                               # |v3PlanePoint_C . (0,0,-1)|
            cop = cen * (camh / np.abs(cen[:, 2]))[:, None]
            rop = rgt * (camh / np.abs(rgt[:, 2]))[:, None]
            dop = dwn * (camh / np.abs(dwn[:, 2]))[:, None]
            right_w = (rop - cop) @ R                             # R^T * v for row vectors
            down_w = (dop - cop) @ R
            idx = np.flatnonzero(good)
            base = n_points
            pid = base + np.arange(len(idx))
            n_points += len(idx)
            P["pos"].append(pos[idx]); P["right"].append(right_w[idx]); P["down"].append(down_w[idx])
            P["src_kf"].append(np.full(len(idx), k, np.int32)); P["level"].append(np.full(len(idx), level, np.int32))
            P["ir"].append(np.stack([xs[idx], ys[idx]], 1).astype(np.int32))

            def add_meas(kf_id, pts_, u_, v_, source):
                n_ = len(pts_)
                M["kf"].append(np.full(n_, kf_id, np.int32)); M["pt"].append(np.asarray(pts_, np.int32))
                M["level"].append(np.full(n_, level, np.int32)); M["root"].append(np.stack([u_, v_], 1).astype(np.float64))
                M["subpix"].append(np.ones(n_, np.int32)); M["source"].append(np.full(n_, source, np.int32))

            add_meas(k, pid, rx[idx], ry[idx], 2)                  # SRC_ROOT
            depth_acc[k].append(pc[idx, 2])
            for k2 in range(n_keyframes):
                if k2 == k:
                    continue
                pc2 = pos[idx] @ Rs[k2].T + ts[k2][None, :]
                z = pc2[:, 2]
                zz = np.where(z > 0.001, z, 1.0)
                u, v = _project_np(cam, W, H, pc2[:, 0] / zz, pc2[:, 1] / zz)
                b = 12 * s
                ins = (z > 0.001) & (u >= b) & (v >= b) & (u < W - b) & (v < H - b)
                add_meas(k2, pid[ins], u[ins], v[ins], 0)          # SRC_TRACKER
                depth_acc[k2].append(z[ins])
    packed = {key: (np.concatenate(val) if val else np.zeros((0,) + shp, dt)) for (key, val, shp, dt) in (
        ("pos", P["pos"], (3,), np.float64), ("right", P["right"], (3,), np.float64), ("down", P["down"], (3,), np.float64),
        ("src_kf", P["src_kf"], (), np.int32), ("level", P["level"], (), np.int32), ("ir", P["ir"], (2,), np.int32),
        ("m_kf", M["kf"], (), np.int32), ("m_pt", M["pt"], (), np.int32), ("m_level", M["level"], (), np.int32),
        ("m_root", M["root"], (2,), np.float64), ("m_subpix", M["subpix"], (), np.int32), ("m_source", M["source"], (), np.int32))}
    # scene depth per keyframe (MapMaker::RefreshSceneDepth, jni/MapMaker.cc:1236-1252)
    for k in range(n_keyframes):
        zs = np.concatenate(depth_acc[k]) if depth_acc[k] else np.ones(1)
        kfs[k]["depth_mean"] = float(zs.mean())
        kfs[k]["depth_sigma"] = float(np.sqrt(max(0.0, (zs ** 2).mean() - zs.mean() ** 2)))
    if point_noise > 0:
        packed["pos"] = packed["pos"] + rng.normal(0, point_noise, packed["pos"].shape)
    if pose_noise[0] > 0 or pose_noise[1] > 0:
        for k in range(1, n_keyframes):
            kfs[k]["pose"] = se3_perturb(kfs[k]["pose"], rng, pose_noise[0], pose_noise[1])
    return MapData(keyframes=kfs, packed=packed, times=times)


class MapData(dict):
    """The map as build_map leaves it: "keyframes" (list of dicts), "times", and "packed" numpy arrays of the points and
    measurements.  "points" (list of dicts) and "meas" (list of tuples) -- the form the tests read -- are derived from
    the packed arrays on first access."""

    def __missing__(self, key):
        pk = dict.__getitem__(self, "packed")
        if key == "points":
            val = [{"pos": pk["pos"][i].copy(), "src_kf": int(pk["src_kf"][i]), "level": int(pk["level"][i]), "irx": int(pk["ir"][i, 0]),
                    "iry": int(pk["ir"][i, 1]), "right": pk["right"][i].copy(), "down": pk["down"][i].copy()} for i in range(len(pk["pos"]))]
        elif key == "meas":
            val = list(zip(pk["m_kf"].tolist(), pk["m_pt"].tolist(), pk["m_level"].tolist(), pk["m_root"][:, 0].tolist(),
                           pk["m_root"][:, 1].tolist(), pk["m_subpix"].tolist(), pk["m_source"].tolist()))
        else:
            raise KeyError(key)
        self[key] = val
        return val
