"""Developer tool: where the one-shot time of an LBA-sized call goes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrsfm_amd import capi
from tests import helpers as H
arr = H.make(7, 1500, 4, seed=5)
opt = capi.default_options(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5)
capi.solve(H.to_product(arr), opt)
acc = np.zeros(4)
N = 20
for _ in range(N):
    p = H.to_product(arr)
    t0 = time.perf_counter(); ctx = capi.Context(p); t1 = time.perf_counter(); ctx.run(opt); t2 = time.perf_counter()
    ctx.download(); t3 = time.perf_counter(); ctx.close(); t4 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3]
print("create %.3f ms, run (incl. Cholesky set-up) %.3f ms, download %.3f ms, destroy %.3f ms" % tuple(acc / N * 1e3))
