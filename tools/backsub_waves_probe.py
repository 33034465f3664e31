#!/usr/bin/env python3
"""Round-2 finding (xi), root-caused on the GPU box: k_backsub built with amdgpu_waves_per_eu(5,5) (96 VGPRs, 22 spilled)
"produced a wrong model decrease (LM trajectory of 23 instead of 13 steps)".  This probe runs ONE back-substitution from the
same camera solution under three builds of the same sources — shipped (4 waves), -DXBA_BACKSUB_WAVES=5, -DXBA_POISON — and
compares the per-item partials (model decrease, squared point step) and the candidate state bit for bit, then the full LM
trajectories.  usage: python tools/backsub_waves_probe.py [S|L|R] -> JSON on stdout"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch  # noqa: F401
from xrsfm_amd import capi, synth
cfg = sys.argv[2]
d = synth.make_problem(**synth.CONFIGS[cfg])
arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
ctx = capi.Context(prob)
ctx.debug_linearize(5.99, True)
y, _ = ctx.debug_cholesky_solve(1e4)
out = ctx.debug_backsub()
out["y"] = y
ctx.reset()
s = ctx.run(capi.default_options())
q, t, P = ctx.download()
ctx.close()
out["summary"] = np.array([s.n_successful, s.n_unsuccessful, s.initial_cost, s.final_cost], float)
out["q"] = q; out["t"] = t
np.savez(sys.argv[1], **out)
"""


def run(lib, cfg, path):
    env = dict(os.environ)
    if lib:
        env["XRSFM_BA_LIB"] = lib
    else:
        env.pop("XRSFM_BA_LIB", None)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, path, cfg], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-1500:] + r.stderr[-1500:])
    return dict(np.load(path))


REPRO_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch  # noqa: F401
from xrsfm_amd import capi, synth
d = synth.make_problem(**synth.CONFIGS[sys.argv[1]])
prob = capi.ProblemArrays(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})
s = capi.solve(prob)
print("RESULT", s.n_successful, s.n_unsuccessful, repr(s.final_cost))
"""


def repro(cfg):
    """Round-2 sources (commit c63e5f7) with k_backsub built for 5 waves per SIMD — (A) as they were, (B) with only
    cam_update_one / quat_plus of the current tree (no CamRec copy in scratch memory): LM trajectories of a full solve.
    The libraries are built by hand from a checkout of that commit (finding xi)."""
    out = {}
    for name in ("r02_w5", "r02_w5_noscratch"):
        lib = os.path.join(ROOT, "xrsfm_amd", "lib", f"libxrsfm_ba_{name}.so")
        if not os.path.exists(lib):
            out[name] = "library not built"; continue
        env = dict(os.environ); env["XRSFM_BA_LIB"] = lib
        r = subprocess.run([sys.executable, "-c", REPRO_CHILD % ROOT, cfg], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
        out[name] = line[0] if line else ("rc %d: " % r.returncode) + r.stderr[-400:]
    return out


def main():
    from xrsfm_amd import _build
    cfg = sys.argv[1] if len(sys.argv) > 1 else "S"
    if len(sys.argv) > 2 and sys.argv[2] == "repro":
        print(json.dumps(repro(cfg), indent=1)); return
    libs = {"shipped": None, "backsub_w5": _build.build_lib(variant="backsub_w5")}      # (poison is built without FP contraction: compare it with "strict", tests/test_gpu_hardening.py)
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, lib in libs.items():
            res[name] = run(lib, cfg, os.path.join(td, name + ".npz"))
    base = res["shipped"]
    report = {"config": cfg}
    for name in ("backsub_w5",):          # (the poison build is compiled without FP contraction: tests/test_gpu_hardening.py compares it with its own twin)
        r = res[name]
        rep = {}
        for k in ("y", "part_model", "part_step2", "cand_points", "point_step", "cand_cam_q", "cand_cam_t", "q", "t"):
            a, b = base[k], r[k]
            neq = int((a != b).sum()) + int(np.isnan(b).sum())
            rep[k] = {"differing": neq, "of": int(a.size), "max_abs_diff": float(np.nanmax(np.abs(a - b))) if a.size else 0.0}
        rep["lm_steps"] = [int(r["summary"][0]), int(r["summary"][1])]
        rep["final_cost"] = float(r["summary"][3])
        rep["model_decrease_sum"] = float(r["part_model"].sum())
        report[name] = rep
    report["shipped"] = {"lm_steps": [int(base["summary"][0]), int(base["summary"][1])], "final_cost": float(base["summary"][3]),
                         "model_decrease_sum": float(base["part_model"].sum())}
    print(json.dumps(report, indent=1))


if __name__ == "__main__":
    main()
