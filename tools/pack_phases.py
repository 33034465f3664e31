"""Developer tool: phase times of xrsfm_ba_create (XRSFM_BA_PACK_TIMING=1) for a bench configuration or an LBA-sized problem.
usage: python tools/pack_phases.py L | lba:<cams>:<points>:<k_obs>"""
import os, sys, time
os.environ["XRSFM_BA_PACK_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xrsfm_amd import capi, synth
for cfg in sys.argv[1:] or ["L"]:
    if cfg.startswith("lba:"):
        nc, npt, k = (int(x) for x in cfg.split(":")[1:])
        d = synth.make_problem(n_cams=nc, n_points=npt, k_obs=k, seed=11)
    else:
        d = synth.make_problem(**synth.CONFIGS[cfg])
    p = capi.ProblemArrays(**{k: d[k] for k in capi.ProblemArrays.FIELDS})
    for rep in range(3):
        print(f"== {cfg} create #{rep}", file=sys.stderr, flush=True)
        t0 = time.perf_counter(); ctx = capi.Context(p); t1 = time.perf_counter()
        s = ctx.run(); t2 = time.perf_counter()
        print(f"== {cfg}: create {1e3*(t1-t0):.2f} ms, first run {1e3*(t2-t1):.2f} ms", file=sys.stderr, flush=True)
        ctx.close()
