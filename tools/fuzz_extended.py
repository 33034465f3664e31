#!/usr/bin/env python3
"""Developer aid (GPU box): more seeds of tests/test_gpu_fuzz.py's random problems against the numpy oracle, exact solver only,
with the panel schedule's macro tiles and panel widths forced on and off (they are plan-time switches).
usage: python tools/fuzz_extended.py [first_seed] [n]"""
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


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = 0
    for seed in range(first, first + n):
        try:
            arr, _ = _problem(seed)
        except ValueError:                      # the generator's parameter draw is not valid for every seed
            continue
        for var in ("XRSFM_BA_PANEL_MACRO", "XRSFM_BA_PANEL_COLS", "XRSFM_BA_PANEL_LL"):
            os.environ.pop(var, None)
        mode = seed % 4
        if mode == 1:
            os.environ["XRSFM_BA_PANEL_MACRO"] = "1"
        elif mode == 2:
            os.environ["XRSFM_BA_PANEL_MACRO"] = "1"; os.environ["XRSFM_BA_PANEL_COLS"] = "4"
        elif mode == 3:
            os.environ["XRSFM_BA_PANEL_LL"] = "0"
        pr = H.to_oracle(arr)
        s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
        prod = H.to_product(arr)
        try:
            s = capi.solve(prod, capi.default_options(linear_solver=1, max_iterations=6))
        except Exception as e:                  # duplicate observations etc. do not occur in these problems
            print("seed", seed, "mode", mode, "EXCEPTION", e); bad += 1; continue
        n_res = 2 * arr["obs_cam"].shape[0]
        ok = ((s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
              and abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
              and np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5)
        if not ok:
            bad += 1
            print("seed", seed, "mode", mode, "MISMATCH", (s.n_successful, s.n_unsuccessful), (s_ref.n_successful, s_ref.n_unsuccessful),
                  s.final_cost, s_ref.final_cost, np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max())
    print(f"{n} problems, {bad} mismatches")


if __name__ == "__main__":
    main()
