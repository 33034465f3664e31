"""Developer check: a camera set far beyond the dense-Cholesky limit with random visibility goes down the PCG path of AUTO
(the plan refuses it with ETOOBIG before the symbolic factorisation) and runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xrsfm_amd import capi, synth
n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = synth.make_problem(n_cams=n_cams, n_points=5 * n_cams, k_obs=4, seed=12, mode="unordered")
arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
t0 = time.perf_counter()
ctx = capi.Context(capi.ProblemArrays(**arr))
t1 = time.perf_counter()
s = ctx.run(capi.default_options(max_iterations=3, pcg_max_iterations=200))
t2 = time.perf_counter()
print(f"{n_cams} cams: create {1e3*(t1-t0):.0f} ms, run {1e3*(t2-t1):.0f} ms, solver {s.linear_solver_used}, "
      f"steps {s.n_successful}+{s.n_unsuccessful}, pcg {s.pcg_iterations}, cost {s.initial_cost:.4e} -> {s.final_cost:.4e}")
assert s.linear_solver_used == capi.SOLVER_PCG and s.final_cost < s.initial_cost
ctx.close()
