"""Developer tool: host-side cost of one BA call = pack + upload (xrsfm_ba_create), first run (Cholesky set-up), run."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xrsfm_amd import capi, synth
for cfg in sys.argv[1:] or ["S", "L"]:
    d = synth.make_problem(**synth.CONFIGS[cfg])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    p = capi.ProblemArrays(**arr)
    t0 = time.perf_counter(); ctx = capi.Context(p); t1 = time.perf_counter()
    s = ctx.run(); t2 = time.perf_counter()
    ctx.reset(); s = ctx.run(); t3 = time.perf_counter()
    q, t, P = ctx.download(); t4 = time.perf_counter()
    print(f"{cfg}: create {1e3*(t1-t0):.1f} ms, first run {1e3*(t2-t1):.1f} ms, second run {1e3*(t3-t2):.1f} ms, download {1e3*(t4-t3):.1f} ms")
    ctx.close()
