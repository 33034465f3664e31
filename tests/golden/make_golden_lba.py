#!/usr/bin/env python3
"""Regenerate tests/golden/lba_selection.npz: frame lists of the LBA selection for a few seeded maps, computed by the numpy
restatement oracle/lba_select.py (FindLocalBundle / CovisibilityNeibors / gauge rule of
/root/reference/src/optimization/ba_solver.cc:393-584).  NOT produced by the reference itself (unbuildable here): the
fixture pins the restatement against regressions and gives the adapter test a list that does not come from the adapter."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [  # n_cams, n_points, k_obs, seed, mode, frame, init1, init2
    (9, 260, 4, 131, "sequential", 2, 0, 1),
    (14, 400, 4, 132, "sequential", 7, 0, 1),       # no init frame in the local set: gauge = last two frames of the bundle
    (12, 500, 6, 133, "unordered", 5, 0, 1),
    (20, 900, 5, 134, "unordered", 11, 0, 1),
    (16, 700, 8, 135, "sequential", 9, 0, 1),
    (16, 700, 8, 135, "sequential", 9, 9, 3),       # the new frame itself is an init frame
    (3, 60, 3, 136, "sequential", 1, 0, 2),         # fewer frames than the bundle size: everything is taken
]


def main():
    from oracle import lba_select as ls
    from tests import helpers as H
    out = {"cases": np.array([(c[0], c[1], c[2], c[3], 0 if c[4] == "sequential" else 1, c[5], c[6], c[7]) for c in CASES], np.int64)}
    for i, (nc, npts, k, seed, mode, fr, i1, i2) in enumerate(CASES):
        arr = H.make(nc, npts, k, seed=seed, mode=mode)
        local, fixed, n1, n2 = ls.lba_frames_and_gauge(fr, arr["obs_cam"], arr["obs_pt"], arr["cam_q"], arr["cam_t"], arr["points"], i1, i2)
        out[f"local{i}"] = np.array(local); out[f"fixed{i}"] = np.array(fixed); out[f"n1_{i}"] = np.array(n1); out[f"n2_{i}"] = np.array(n2)
    np.savez(os.path.join(ROOT, "tests", "golden", "lba_selection.npz"), **out)
    print("written", len(CASES), "cases")


if __name__ == "__main__":
    main()
