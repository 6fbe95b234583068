#!/usr/bin/env python3
"""Golden vectors for the M-estimators from the reference's jni/MEstimator.h compiled verbatim
(oracle/_ref/libref_mestimator.so, built by oracle/Makefile).  Writes tests/golden/mestimator_ref.json:
inputs (seeded) and the reference's outputs only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import binding as B  # noqa: E402


def main():
    ref = B.ref_mestimator()
    if ref is None:
        raise SystemExit("oracle/_ref/libref_mestimator.so missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(2024)
    cases = []
    for n in (1, 2, 3, 4, 7, 50, 333, 1000):
        v = (rng.standard_normal(n) ** 2 * rng.uniform(0.01, 30)).tolist()
        if n == 7:
            v[3] = v[5]   # ties
        arr = np.ascontiguousarray(v, np.float64)   # keep alive across the calls
        sig = [ref.ref_find_sigma_squared(e, arr.ctypes.data, n) for e in range(4)]
        cases.append({"v": v, "sigma_squared": sig})
    pts = []
    for e2, s2 in [(0.0, 1.0), (0.5, 1.0), (1.0, 1.0), (1.0000001, 1.0), (3.7, 2.2), (16.0, 16.0), (100.0, 0.16), (1e-9, 0.16)]:
        pts.append({"e2": e2, "s2": s2,
                    "weight": [ref.ref_weight(e, e2, s2) for e in range(4)],
                    "sqrt_weight": [ref.ref_sqrt_weight(e, e2, s2) for e in range(4)],
                    "objective": [ref.ref_objective(e, e2, s2) for e in range(4)]})
    out = {"source": "jni/MEstimator.h compiled verbatim by oracle/Makefile (estimators: 0 Tukey, 1 Cauchy, 2 Huber, 3 LeastSquares)",
           "sigma_cases": cases, "point_cases": pts}
    json.dump(out, open(os.path.join(HERE, "..", "tests", "golden", "mestimator_ref.json"), "w"), indent=1)
    print("wrote", len(cases), "sigma cases,", len(pts), "point cases")


if __name__ == "__main__":
    main()
