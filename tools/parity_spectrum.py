#!/usr/bin/env python3
"""Where do two exact FP64 solvers disagree?  Solve a bench configuration with the HIP library and with the C restatement
(oracle/ba_cpu.c), take the reduced camera matrix S of the final linearisation (unscaled tangent coordinates, no damping)
and expand the camera-parameter difference in its eigenvectors.  Developer aid behind tests/test_gpu_parity.py::
test_headline_config_camera_parity and DESIGN.md section 5.   usage (GPU box): python tools/parity_spectrum.py [config] [threads]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "L"
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import torch
    from oracle import ba_cpu
    from xrsfm_amd import capi, parity, synth
    d = synth.make_problem(**synth.CONFIGS[cfg])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    prod = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
    s = capi.solve(prod, capi.default_options())
    cpu = {k: np.array(v, copy=True) for k, v in arr.items()}
    sc = ba_cpu.solve(cpu, max_iterations=50, threads=threads)
    print("iterations", s.n_successful, s.n_unsuccessful, "cpu", sc["n_successful"], sc["n_unsuccessful"], "final cost", s.final_cost, sc["final_cost"])
    print("max |dq|", np.abs(prod.cam_q - cpu["cam_q"]).max(), "max |dt|", np.abs(prod.cam_t - cpu["cam_t"]).max(),
          "max |dP|", np.abs(prod.points - cpu["points"]).max())
    t0 = time.time()
    rep = parity.camera_difference_spectrum(prod, cpu["cam_q"], cpu["cam_t"], device="cuda")
    print("spectrum time", time.time() - t0)
    lam, c = rep["eigenvalues"], rep["coefficients"]
    print("n", lam.shape[0], "lambda min/1%/10%/50%/max", lam[0], lam[int(0.01 * len(lam))], lam[int(0.1 * len(lam))], lam[len(lam) // 2], lam[-1])
    print("|d|_inf", rep["d_inf"], "|d|_2", rep["d_2"], "energy d^T S d", rep["energy"], "final cost", s.final_cost)
    for k in (0, 1, 2, 3, 4, 6, 7, 8, 12, 16, 24, 32, 64, 128, 256):
        print(f"  without the {k:3d} weakest modes: |rest|_inf = {rep['rest_inf'](k):.3e}   lambda_k = {lam[min(k, len(lam) - 1)]:.3e}")
    print("largest coefficients (mode index, lambda, c):")
    for i in np.argsort(-np.abs(c))[:12]:
        print(f"   {i:5d} {lam[i]:.3e} {c[i]:+.3e}")
    out = {"config": cfg, "threads": threads, "d_inf": rep["d_inf"], "energy": rep["energy"], "final_cost": s.final_cost,
           "rest_inf": {str(k): rep["rest_inf"](k) for k in (0, 1, 2, 4, 7, 8, 16, 32, 64)}, "lambda_first_16": lam[:16].tolist()}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"parity_spectrum_{cfg}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
