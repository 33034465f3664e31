#!/usr/bin/env python3
"""Developer tool: N one-shot solves (create + run + download + destroy on fresh host arrays, as bench.py's host_inclusive) of one
config back to back, optionally with a pause between them — does a call pay for the release of the previous one?
usage (GPU box): XRSFM_BA_PACK_TIMING=1 python tools/oneshot_repeat.py K 4 [pause_ms]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xrsfm_amd import capi, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "K"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pause = float(sys.argv[3]) / 1e3 if len(sys.argv) > 3 else 0.0
d = synth.make_collection(**synth.CONFIGS[cfg]) if cfg == "T" else synth.make_problem(**synth.CONFIGS[cfg])
arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
opt = capi.default_options()
for rep in range(n):
    prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
    print(f"== call {rep}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    ctx = capi.Context(prob)
    t1 = time.perf_counter()
    s = ctx.run(opt)
    t2 = time.perf_counter()
    ctx.download(out=(prob.cam_q, prob.cam_t, prob.points))
    t3 = time.perf_counter()
    ctx.close()
    t4 = time.perf_counter()
    print(f"call {rep}: create {1e3 * (t1 - t0):.2f} run {1e3 * (t2 - t1):.2f} download {1e3 * (t3 - t2):.2f} destroy {1e3 * (t4 - t3):.2f} ms")
    if pause:
        time.sleep(pause)
