#!/usr/bin/env python3
"""Developer aid (GPU box): given fuzz seeds of tests/test_gpu_fuzz.py's generator, solve with the library XRSFM_BA_LIB points to (default:
the shipped one) and print the differences to the numpy oracle.  usage: python tools/fuzz_seeds.py solver seed [seed ...]"""
import math, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo
from tests import helpers as H
from tests.test_gpu_fuzz import _problem
from xrsfm_amd import capi
solver = int(sys.argv[1])
for seed in map(int, sys.argv[2:]):
    arr, _ = _problem(seed)
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(linear_solver=solver, max_iterations=6))
    n_res = 2 * arr["obs_cam"].shape[0]
    print(f"seed {seed} solver {solver}: LM {(s.n_successful, s.n_unsuccessful)} vs {(s_ref.n_successful, s_ref.n_unsuccessful)}, rmse diff "
          f"{abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)):.2e}, cam diff {max(np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max()):.2e}, "
          f"cost {s.final_cost:.6e} pcg {s.pcg_iterations}")
