"""Developer tool: xrsfm_ba_create + first run (Cholesky set-up) with host packing vs device packing over problem sizes —
where the device path (ba_pack_dev.h) starts to pay.  usage (GPU box): python tools/pack_crossover.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from xrsfm_amd import capi, synth
    n_c, n_p = int(sys.argv[1]), int(sys.argv[2])
    d = synth.make_problem(n_cams=n_c, n_points=n_p, k_obs=4, seed=3)
    p = capi.ProblemArrays(**{k: d[k] for k in capi.ProblemArrays.FIELDS})
    best = None
    for rep in range(6):
        t0 = time.perf_counter(); ctx = capi.Context(p); t1 = time.perf_counter()
        ctx.run(capi.default_options(max_iterations=1)); t2 = time.perf_counter()
        ctx.close()
        if rep >= 2:
            cur = (1e3 * (t1 - t0), 1e3 * (t2 - t1))
            best = cur if best is None or sum(cur) < sum(best) else best
    print(f"{best[0]:.3f} {best[1]:.3f}")
else:
    print("cams points obs | host: create first-run(1 it) | device: create first-run(1 it)   [ms]")
    for n_c, n_p in ((8, 1500), (20, 5000), (40, 12000), (60, 25000), (100, 50000), (200, 100000), (400, 250000)):
        row = []
        for mode in ("0", "1"):
            env = dict(os.environ, XRSFM_BA_DEVICE_PACK=mode)
            r = subprocess.run([sys.executable, __file__, str(n_c), str(n_p)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            row.append(r.stdout.decode().strip().splitlines()[-1] if r.stdout else "fail")
        print(f"{n_c:5d} {n_p:7d} {4 * n_p:8d} | {row[0]:>16s} | {row[1]:>16s}", flush=True)
