"""Developer check: a sequential problem several times the size of config K (default 4000 cameras / 5M points / 20M
observations) through the default path: sizes, index arithmetic and memory at scale."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xrsfm_amd import capi, synth
n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
t0 = time.perf_counter()
d = synth.make_problem(n_cams=n_cams, n_points=n_points, k_obs=4, seed=13)
arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
t1 = time.perf_counter()
ctx = capi.Context(capi.ProblemArrays(**arr))
t2 = time.perf_counter()
s = ctx.run(capi.default_options())
t3 = time.perf_counter()
ctx.reset(); s2 = ctx.run(capi.default_options())
t4 = time.perf_counter()
n_res = 2 * arr["obs_cam"].shape[0]
print(f"{n_cams} cams / {n_points} points / {n_res // 2} obs: generate {t1-t0:.1f} s, create {1e3*(t2-t1):.0f} ms, first run {1e3*(t3-t2):.0f} ms, "
      f"second run {1e3*(t4-t3):.0f} ms, solver {s.linear_solver_used}, steps {s.n_successful}+{s.n_unsuccessful}, "
      f"rmse {math.sqrt(s.initial_cost / n_res):.3f} -> {math.sqrt(s.final_cost / n_res):.3f} px")
assert s.final_cost < 0.1 * s.initial_cost and s2.final_cost == s.final_cost
ctx.close()
