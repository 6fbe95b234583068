"""Seeded synthetic bundle-adjustment scenes (BASELINE.json configs[2] / SURVEY.md 8(d) config 3): host-side test and bench input, numpy only."""
import numpy as np

CAM = (0.841906, 1.10893, 0.505171, 0.470265, -0.0133843)


def _rot(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def project(pose12, X, w=640, h=480):
    R, t = pose12[:9].reshape(3, 3), pose12[9:]
    c = R @ X + t
    x, y = c[0] / c[2], c[1] / c[2]
    r = np.hypot(x, y)
    ww = CAM[4]
    fac = 1.0 if r < 0.001 else np.arctan(r * 2 * np.tan(ww / 2)) / ww / r
    return np.array([w * CAM[2] - 0.5 + w * CAM[0] * x * fac, h * CAM[3] - 0.5 + h * CAM[1] * y * fac]), c[2]


def ba_scene(n_cams=5, n_pts=300, pixel_noise=0.5, outlier_frac=0.05, visibility=1.0, seed=0,
             init_noise=(0.01, 0.5 * np.pi / 180, 0.02), n_fixed=1):
    """n_cams cameras (camera 0 fixed) looking at n_pts points in a 2 x 2 x 0.5 box; init = truth + noise."""
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-0.6, 0.6, n_pts), rng.uniform(-0.45, 0.45, n_pts), rng.uniform(-0.25, 0.25, n_pts)], 1)
    cams = []
    for j in range(n_cams):
        C = np.array([0.25 * np.cos(2 * np.pi * j / n_cams), 0.25 * np.sin(2 * np.pi * j / n_cams), -1.6 + 0.05 * j])
        R = _rot(rng.normal(0, 0.04, 3))
        cams.append(np.concatenate([R.ravel(), -R @ C]))
    cams = np.array(cams)
    meas, outliers = [], []
    for j in range(n_cams):
        for i in range(n_pts):
            if rng.uniform() > visibility:
                continue
            xy, z = project(cams[j], pts[i])
            if z <= 0.1 or not (5 < xy[0] < 635 and 5 < xy[1] < 475):
                continue
            level = int(rng.integers(0, 4))
            xy = xy + rng.normal(0, pixel_noise, 2) if pixel_noise > 0 else xy
            if rng.uniform() < outlier_frac:
                xy = xy + rng.choice([-20.0, 20.0], 2)
                outliers.append((j, i))
            meas.append((j, i, xy, float((1 << level) ** 2)))
    cams_init = cams.copy()
    for j in range(n_fixed, n_cams):
        R, t = cams[j][:9].reshape(3, 3), cams[j][9:]
        dR = _rot(rng.normal(0, init_noise[1], 3))
        cams_init[j] = np.concatenate([(dR @ R).ravel(), dR @ t + rng.normal(0, init_noise[0], 3)])
    pts_init = pts + rng.normal(0, init_noise[2], pts.shape)
    return {"cams_true": cams, "pts_true": pts, "cams_init": cams_init, "pts_init": pts_init,
            "fixed": [j < n_fixed for j in range(n_cams)], "meas": meas, "outliers": outliers}
