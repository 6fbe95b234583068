"""CPU tests: oracle Bundle (jni/Bundle.cc restatement) -- known-answer test K5 of SURVEY.md 8(c)."""
import numpy as np

from ba_scene import ba_scene, CAM


def run_oracle(oracle, sc, max_it=20, conv=1e-6):
    b = oracle.OracleBundle(CAM, 640, 480, max_iterations=max_it, convergence_limit=conv)
    for i, (pose, fixed) in enumerate(zip(sc["cams_init"], sc["fixed"])):
        b.add_camera(pose, fixed)
    for p in sc["pts_init"]:
        b.add_point(p)
    for (c, p, xy, s2) in sc["meas"]:
        b.add_meas(c, p, xy, s2)
    return b


def test_noiseless_scene_converges_to_ground_truth(oracle):
    # two fixed cameras remove the scale gauge freedom (with one, monocular BA is only defined up to scale)
    sc = ba_scene(n_cams=5, n_pts=120, pixel_noise=0.0, outlier_frac=0.0, seed=3, n_fixed=2)
    b = run_oracle(oracle, sc, max_it=60, conv=1e-22)    # tighter than the reference's 1e-6 so LM runs to the optimum
    for _ in range(4):
        if b.compute() <= 0:
            break
    # gauge: camera 0 fixed; cameras and points return to the truth
    assert np.abs(b.cameras() - sc["cams_true"]).max() < 1e-7
    assert np.abs(b.points() - sc["pts_true"]).max() < 1e-6
    b2 = run_oracle(oracle, sc, max_it=60)                # reference convergence limit (sum of squared updates < 1e-6)
    while not b2.converged() and b2.compute() >= 0:
        pass
    assert np.abs(b2.cameras() - sc["cams_true"]).max() < 2e-3
    assert len(b.outlier_meas()) < 0.05 * len(sc["meas"])   # the perturbed start may cost a few measurements (Tukey cut-off)


def test_gross_outliers_are_flagged(oracle):
    sc = ba_scene(n_cams=5, n_pts=150, pixel_noise=0.3, outlier_frac=0.06, seed=4)
    b = run_oracle(oracle, sc, max_it=40)
    for _ in range(4):
        if b.compute() < 0 or b.converged():
            break
    flagged = {(int(p), int(c)) for p, c in b.outlier_meas()}
    truth = {(p, c) for (c, p) in sc["outliers"]}
    # +-20 px outliers are erased (a level-3 measurement is weighted by 1/8, so a few coarse ones survive the cut-off)
    assert len(truth & flagged) >= 0.85 * len(truth)
    assert len(flagged) <= len(truth) + 0.10 * len(sc["meas"])  # and few inliers are
    assert np.abs(b.cameras() - sc["cams_true"]).max() < 5e-3


def test_counters_and_edge_cases(oracle):
    sc = ba_scene(n_cams=3, n_pts=30, pixel_noise=0.2, outlier_frac=0.0, seed=5)
    b = run_oracle(oracle, sc, max_it=3)
    acc = b.compute()
    s2, lam, trials = b.stats()
    assert 0 <= acc <= 3 and trials <= 3 and s2 >= 0.16        # iteration cap counts trials; sigma^2 clamp (jni/Bundle.cc:224-227)
    empty = oracle.OracleBundle(CAM, 640, 480)
    empty.add_camera(sc["cams_init"][0], True)
    empty.add_point([0, 0, 0])
    assert empty.compute() < 0                                  # no measurements: error, not a crash
