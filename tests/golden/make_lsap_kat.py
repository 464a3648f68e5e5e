"""Known-answer vectors for oracle/lsap.c and the HIP LSAP kernel, generated from SciPy (the third-party dependency the
reference calls at models/matcher.py:85 and models/mdetr.py:100,539; scipy 1.15.3 in the build container, 1.7.3 pinned by the
reference -- same documented algorithm).  They travel to the GPU box, where SciPy need not exist.
Run:  python tests/golden/make_lsap_kat.py"""
import json
import os

import numpy as np
import scipy
from scipy.optimize import linear_sum_assignment

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260928)
    cases = []
    shapes = [(4, 2), (2, 4), (5, 3), (100, 4), (100, 10), (20, 30), (7, 7), (97, 97), (1, 1), (3, 0)]
    for i, (nr, nc) in enumerate(shapes):
        for mode in range(4):
            if mode == 0:
                c = rng.random((nr, nc)).astype(np.float32)
            elif mode == 1:
                c = rng.integers(0, 3, (nr, nc)).astype(np.float32)          # many exact ties
            elif mode == 2:
                c = np.zeros((nr, nc), dtype=np.float32)
            else:
                c = rng.random((nr, nc)).astype(np.float32)
                if nc > 1:
                    c[:, 1] = c[:, 0]                                        # duplicated target column
                if nr > 2:
                    c[2] = c[0]                                              # duplicated query row
            r, k = linear_sum_assignment(c.astype(np.float64))
            cases.append({"rows": nr, "cols": nc, "cost": [float(v) for v in c.reshape(-1)], "row_ind": r.tolist(), "col_ind": k.tolist()})
    with open(os.path.join(HERE, "lsap_kat.json"), "w") as f:
        json.dump({"scipy": scipy.__version__, "cases": cases}, f)
    print("wrote lsap_kat.json:", len(cases), "cases, scipy", scipy.__version__)


if __name__ == "__main__":
    main()
