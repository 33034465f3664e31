#!/usr/bin/env python3
"""Golden fixture of the bal9 mode (9-wide camera blocks: extension camera model 5 with variable {f, k1, k2} per camera),
generated with the numpy oracle (oracle/ba_oracle.py).  Run from the repo root:  python tests/golden/make_golden_bal9.py

Like the other fixtures it pins OUR oracle ("parity unpinned" against real Ceres); the reference never frees intrinsics
(/root/reference/src/optimization/ba_solver.cc:602-606), so it has no vectors for this mode either."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    arr = H.make_bal9(8, 90, 4, seed=31)
    cc = arr["cam_const"].copy(); cc[3] &= 3; cc[6] |= 2          # one camera with constant intrinsics, one with constant translation
    arr["cam_const"] = cc
    arr["point_const"] = (np.arange(90) % 11 == 0).astype(np.uint8)
    opt = bo.Options(max_iterations=15)
    pr = H.to_oracle(arr)
    cost0, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    s = bo.solve(pr, opt)
    np.savez_compressed(
        os.path.join(OUT, "wide_bal9.npz"), **{"in_" + k: v for k, v in arr.items()},
        opt=np.array([opt.max_iterations, opt.function_tolerance, opt.parameter_tolerance, opt.initial_radius]),
        init_cost=cost0, init_r=rt, init_Jc=Fc, init_Jp=Ep,
        out_cam_q=pr.cam_q, out_cam_t=pr.cam_t, out_points=pr.points, out_intr=pr.intr_params,
        final_cost=s.final_cost, n_successful=s.n_successful, n_unsuccessful=s.n_unsuccessful)
    print("wide_bal9", s.termination, s.n_successful, s.n_unsuccessful, Fc.shape)


if __name__ == "__main__":
    main()
