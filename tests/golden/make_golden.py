#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the numpy oracle (oracle/ba_oracle.py).

Run from the repo root:  python tests/golden/make_golden.py
The reference ships no golden vectors for this path and its arithmetic (Ceres) cannot run here
(SURVEY.md 8c), so these vectors pin OUR oracle — "parity unpinned" against real Ceres.  They keep the
oracle, the C restatement and the HIP path from drifting apart silently.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, arr, opt):
    pr = H.to_oracle(arr)
    cost0, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    s = bo.solve(pr, opt)
    trace_cost = np.array([t["cost"] for t in s.trace if t.get("ok") is not None], float)
    trace_ok = np.array([1 if t["ok"] else 0 for t in s.trace if t.get("ok") is not None], np.int8)
    rm = bo.rmse_pair(pr)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), **{"in_" + k: v for k, v in arr.items()},
        opt=np.array([opt.max_iterations, opt.function_tolerance, opt.parameter_tolerance, opt.initial_radius]),
        init_cost=cost0, init_r=rt, init_Jc=Fc, init_Jp=Ep,
        out_cam_q=pr.cam_q, out_cam_t=pr.cam_t, out_points=pr.points,
        final_cost=s.final_cost, n_successful=s.n_successful, n_unsuccessful=s.n_unsuccessful,
        trace_cost=trace_cost, trace_ok=trace_ok, rmse_ref_style=rm[0], rmse_plain=rm[1])
    print(name, s.termination, s.n_successful, s.n_unsuccessful, "rmse", rm)


def main():
    only = sys.argv[1:]     # optional: names of the cases to (re)generate
    global case
    if only:
        _case = case
        case = lambda name, arr, opt: _case(name, arr, opt) if name in only else None   # noqa: E731
    # GBA accurate (ba_solver.cc:626-629), KITTI SIMPLE_RADIAL intrinsics
    case("gba_kitti", H.make(6, 60, 4, seed=100), bo.Options())
    # all five camera models of camera_model.hpp, one per camera (cycled)
    case("gba_models", H.with_models(H.make(10, 120, 4, seed=101), seed=5), bo.Options())
    # a few points behind their cameras: clamp branch (12,12) with zero Jacobian (cost_factor_ceres.h:29-31)
    arr = H.make(6, 60, 4, seed=102)
    arr["points"][::9] += np.array([0.0, 0.0, -80.0])
    case("gba_behind", arr, bo.Options(max_iterations=10))
    # KGBA options (ba_solver.cc:667-670)
    case("kgba", H.make(8, 80, 3, seed=103),
         bo.Options(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5, initial_radius=1e6))
    # LBA: 5 iterations, points not seen by the newest frame held constant (ba_solver.cc:380-382, 587-589)
    arr = H.make(7, 90, 4, seed=104)
    seen = np.zeros(arr["points"].shape[0], bool); seen[arr["obs_pt"][arr["obs_cam"] == 6]] = True
    arr["point_const"][:] = (~seen).astype(np.uint8)
    case("lba", arr, bo.Options(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5))
    # structure-only GBA(map, true, true) (ba_solver.cc:616-621)
    arr = H.make(6, 60, 4, seed=105); arr["cam_const"][:] = 3
    case("structure_only", arr, bo.Options())
    # pose-only refinement of RegisterImage (pnp.cc:38-71): one camera, every point constant, Ceres defaults, 10 iterations
    case("pose_refine", H.make_pose_problem(150, seed=106, model=2),
         bo.Options(max_iterations=10, function_tolerance=1e-6, parameter_tolerance=1e-8))


if __name__ == "__main__":
    main()
