#!/usr/bin/env python3
"""Developer aid (GPU box): more seeds of tests/test_gpu_fuzz.py's random problems against the numpy oracle, exact solver only,
with the panel schedule's macro tiles and panel widths forced on and off (they are plan-time switches).
usage: python tools/fuzz_extended.py [first_seed] [n] [big|bigchol|tiny|pcg]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo          # noqa: E402
from tests import helpers as H              # noqa: E402
from tests.test_gpu_fuzz import _problem    # noqa: E402
from xrsfm_amd import capi                  # noqa: E402


def _big_problem(seed):
    """Larger problems than tests/test_gpu_fuzz.py draws: 70-260 cameras (nested-dissection orderings with split levels, deep
    trees with several tile columns), tracks of 2-24 observations, missed detections, constant blocks."""
    rng = np.random.default_rng(5000 + seed)
    n_cams = int(rng.integers(70, 260))
    mode = "unordered" if rng.random() < 0.35 else "sequential"
    k_obs = int(rng.integers(2, 9)) if rng.random() < 0.6 else int(rng.integers(9, 25))
    n_pts = int(rng.integers(300, 2500))
    dropout = float(rng.choice([0.0, 0.2, 0.4])) if (mode == "sequential" and k_obs > 2) else 0.0
    arr = H.make(n_cams, n_pts, k_obs, seed=7000 + seed, mode=mode, dropout=dropout, min_tri_angle_deg=0.5)
    if rng.random() < 0.4:
        arr = H.with_models(arr, seed=seed)
    if rng.random() < 0.3:
        arr["point_const"] = (rng.random(arr["points"].shape[0]) < 0.3).astype(np.uint8)
    if rng.random() < 0.3:
        cc = arr["cam_const"].copy(); cc[rng.integers(0, n_cams, 3)] |= 3; arr["cam_const"] = cc
    return arr, int(rng.integers(0, 2))


def _tiny_problem(seed):
    """LBA-sized: 2-12 cameras (one or two tiles of the reduced system), 10-300 points, some constant."""
    rng = np.random.default_rng(9000 + seed)
    n_cams = int(rng.integers(2, 13))
    k_obs = int(rng.integers(2, n_cams + 1))
    n_pts = int(rng.integers(10, 300))
    arr = H.make(n_cams, n_pts, k_obs, seed=11000 + seed, mode="unordered" if rng.random() < 0.5 else "sequential", min_tri_angle_deg=0.5)
    if rng.random() < 0.5:
        arr["point_const"] = (rng.random(arr["points"].shape[0]) < 0.4).astype(np.uint8)
    if rng.random() < 0.5 and n_cams > 3:
        cc = arr["cam_const"].copy(); cc[rng.integers(0, n_cams, 2)] |= 3; arr["cam_const"] = cc
    return arr, int(rng.integers(0, 2))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    big = len(sys.argv) > 3 and sys.argv[3] in ("big", "bigchol")
    tiny = len(sys.argv) > 3 and sys.argv[3] == "tiny"
    bad = 0
    ran = 0
    for seed in range(first, first + n):
        try:
            arr, solver = _big_problem(seed) if big else (_tiny_problem(seed) if tiny else _problem(seed))
        except (ValueError, RuntimeError):      # the generators' parameter draws are not valid for every seed
            continue
        if len(sys.argv) > 3 and sys.argv[3] == "bigchol":
            solver = 1
        if not big and not tiny:
            solver = 0 if (len(sys.argv) > 3 and sys.argv[3] == "pcg") else 1
        for var in ("XRSFM_BA_PANEL_MACRO", "XRSFM_BA_PANEL_COLS", "XRSFM_BA_LOOKAHEAD", "XRSFM_BA_PACKED"):
            os.environ.pop(var, None)
        mode = seed % 4
        if mode == 1:
            os.environ["XRSFM_BA_PANEL_MACRO"] = "1"
        elif mode == 2:
            os.environ["XRSFM_BA_PANEL_MACRO"] = "1"; os.environ["XRSFM_BA_PANEL_COLS"] = "4"
        elif mode == 3:
            os.environ["XRSFM_BA_LOOKAHEAD"] = "0"; os.environ["XRSFM_BA_PACKED"] = "1"      # round-2 panel schedule, packed tile storage
        ran += 1
        pr = H.to_oracle(arr)
        s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
        prod = H.to_product(arr)
        try:
            s = capi.solve(prod, capi.default_options(linear_solver=solver, max_iterations=6))
        except Exception as e:                  # duplicate observations etc. do not occur in these problems
            print("seed", seed, "mode", mode, "EXCEPTION", e); bad += 1; continue
        n_res = 2 * arr["obs_cam"].shape[0]
        ok = ((s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
              and abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
              and np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5)
        if not ok:
            bad += 1
            print("seed", seed, "mode", mode, "solver", solver, "cams", arr["cam_q"].shape[0], "MISMATCH", (s.n_successful, s.n_unsuccessful), (s_ref.n_successful, s_ref.n_unsuccessful),
                  s.final_cost, s_ref.final_cost, np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max())
    print(f"{ran} problems of {n} seeds, {bad} mismatches")


if __name__ == "__main__":
    main()
