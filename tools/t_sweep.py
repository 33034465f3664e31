#!/usr/bin/env python3
"""Developer tool: config T (BASELINE config 5's shape) generated ONCE, then solved under several plan settings.

usage: python tools/t_sweep.py [--scale 1.0] SETTING [SETTING ...]      SETTING = comma-separated ENV=VALUE pairs, or "default"
e.g.   python tools/t_sweep.py default XRSFM_BA_ND=0
Per setting: create (incl. plan) ms, solve ms (second run of the context: no set-up), LM iterations, final cost, and the
HIP-event totals per kernel class of a third, profiled run.  The plan's environment switches are read when the context's
Cholesky structures are built, so one process can compare them."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from tests import helpers as H  # noqa: E402
from xrsfm_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of config T's cameras and points")
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    cfg = dict(synth.CONFIGS["T"])
    cfg["n_cams"] = int(cfg["n_cams"] * a.scale); cfg["n_points"] = int(cfg["n_points"] * a.scale)
    t0 = time.time()
    d = synth.make_collection(**cfg)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    print(f"generated {cfg['n_cams']} cams / {cfg['n_points']} points / {arr['obs_cam'].shape[0]} obs in {time.time() - t0:.1f} s", flush=True)
    ref_cost = None
    for st in a.settings:
        env = {} if st == "default" else dict(kv.split("=", 1) for kv in st.split(","))
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            opt = capi.default_options(max_iterations=a.iterations, function_tolerance=1e-4, parameter_tolerance=1e-5)
            t1 = time.perf_counter()
            ctx = capi.Context(H.to_product(arr))
            t2 = time.perf_counter()
            s = ctx.run(opt)                        # first run: Cholesky set-up (plan) inside
            t3 = time.perf_counter()
            ctx.reset()
            s2 = ctx.run(opt)
            t4 = time.perf_counter()
            ctx.reset()
            popt = capi.default_options(max_iterations=a.iterations, function_tolerance=1e-4, parameter_tolerance=1e-5, profile=1)
            ctx.run(popt)
            prof = ctx.profile()
            ctx.close()
            if ref_cost is None:
                ref_cost = s.final_cost
            kern = ", ".join(f"{k} {v[0]:.1f} ms/{v[1]}" for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:8])
            print(f"{st}: create {1e3 * (t2 - t1):.0f} ms, first run {1e3 * (t3 - t2):.0f} ms, solve {1e3 * (t4 - t3):.1f} ms "
                  f"({s2.n_successful}+{s2.n_unsuccessful} LM, solver {s2.linear_solver_used}), cost {s2.final_cost:.9e} "
                  f"(rel. to first setting {abs(s2.final_cost - ref_cost) / ref_cost:.1e}), same as first run: {s2.final_cost == s.final_cost}", flush=True)
            print(f"    {kern}", flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if __name__ == "__main__":
    main()
