"""Developer aid: one exact solve of a quarter-size clustered collection (2500 cameras: 250 tile columns, the look-ahead panel
schedule of config T at a size a trace can hold).  usage: rocprofv3 --kernel-trace ... -- python tools/tq_trace.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xrsfm_amd import capi, synth  # noqa: E402

d = synth.make_collection(n_cams=int(os.environ.get("TQ_CAMS", "2500")), n_points=int(os.environ.get("TQ_POINTS", "600000")), seed=12)
prob = capi.ProblemArrays(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})
print(capi.debug_chol_plan(capi.ProblemArrays(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})))
ctx = capi.Context(prob)
opt = capi.default_options(max_iterations=int(os.environ.get("TQ_ITERS", "4")), linear_solver=capi.SOLVER_CHOLESKY)
t0 = time.perf_counter()
s = ctx.run(opt)
print("solve", round((time.perf_counter() - t0) * 1e3, 1), "ms", s.n_successful, s.n_unsuccessful, s.linear_solver_used)
ctx.close()
