import math, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import ba_oracle as bo
from tests import helpers as H
from tests.test_gpu_fuzz import _problem
from xrsfm_amd import capi
for seed in (63, 96, 270, 249):
    arr, _ = _problem(seed)
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
    for solver in (1, 0):
        prod = H.to_product(arr)
        s = capi.solve(prod, capi.default_options(linear_solver=solver, max_iterations=6))
        print("seed", seed, "solver", solver, (s.n_successful, s.n_unsuccessful), (s_ref.n_successful, s_ref.n_unsuccessful), s.final_cost, s_ref.final_cost, float(np.abs(prod.cam_t - pr.cam_t).max()))
