import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from xrsfm_amd import capi, synth
from tests import helpers as H
for (nc, npts, k) in [(7, 1500, 4), (7, 6000, 5), (30, 20000, 4), (100, 50000, 4)]:
    arr = H.make(nc, npts, k, seed=5)
    opt = capi.default_options(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5)
    capi.solve(H.to_product(arr), opt)
    ts=[]
    for _ in range(5):
        p = H.to_product(arr); t0=time.perf_counter(); s = capi.solve(p, opt); ts.append(time.perf_counter()-t0)
    ctx = capi.Context(H.to_product(arr)); ctx.run(opt)
    tr=[]
    for _ in range(5):
        ctx.reset(); t0=time.perf_counter(); s2 = ctx.run(opt); tr.append(time.perf_counter()-t0)
    print(f"{nc} cams {npts} pts {arr['obs_cam'].shape[0]} obs: one-shot solve {min(ts)*1e3:.2f} ms (run only {min(tr)*1e3:.2f} ms), {s.n_successful+s.n_unsuccessful} iterations")
