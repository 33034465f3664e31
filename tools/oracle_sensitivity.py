#!/usr/bin/env python3
"""Which recalled Ceres details matter?  

The oracle (oracle/ba_oracle.py) restates Ceres' trust-region loop from memory and is UNPINNED:
Ceres is not vendored under /root/reference and not installed here.  This tool flips ONE recalled detail at a time
(ba_oracle.ALT_DETAILS) and reports, per problem, the change in LM step counts and in the final reference-style RMSE
(sqrt(final_cost / num_residuals), ba_solver.cc:36-39) against the restatement as recalled — so that a maintainer with a
Ceres install knows which details a real-Ceres golden would actually discriminate, and which cannot be told apart on these
problems at all.  CPU only.  usage: python tools/oracle_sensitivity.py [out.md]"""
import glob
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo      # noqa: E402
from tests import helpers as H          # noqa: E402


def problems():
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        name = os.path.basename(p)[:-4]
        if name in ("track_filter", "tag_refine", "pose_graph", "lba_selection", "pose_refine") or name.startswith("ceres_"):
            continue
        z = np.load(p)
        arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
        mi, ft, pt, rad = z["opt"]
        out.append((name, arr, dict(max_iterations=int(mi), function_tolerance=float(ft), parameter_tolerance=float(pt), initial_radius=float(rad))))
    # config S of BASELINE.json at a tenth of its points (the numpy oracle's dense Schur complement does not scale to 200k observations)
    out.append(("S/10 (100 cams, 5000 pts)", H.make(100, 5000, 4, seed=2), {}))
    # an ill-conditioned one: Appendix D read literally at small size (0.25-unit baselines), where accept/reject decisions are close calls
    from xrsfm_amd import synth
    d = synth.make_problem(200, 4000, 4, seed=4, literal_appendix_d=True)
    out.append(("L0-like (200 cams, radius-40 ring)", {k: d[k] for k in H.FIELDS}, {}))
    # ... and one whose trajectory has rejected steps (the shape catalogue's last problem: 254 cameras, 40-camera tracks)
    out.append(("with rejected steps (254 cams)", H.shape_problems()[20][0], dict(max_iterations=12)))
    return out


def main():
    rows = []
    probs = problems()
    base = {}
    for name, arr, kw in probs:
        pr = H.to_oracle(arr)
        s = bo.solve(pr, bo.Options(**kw))
        n_res = 2 * arr["obs_cam"].shape[0]
        base[name] = (s.n_successful, s.n_unsuccessful, math.sqrt(s.final_cost / n_res), pr.cam_q.copy(), pr.cam_t.copy())
    lines = ["| recalled detail flipped | " + " | ".join(n for n, _, _ in probs) + " |", "|---|" + "---|" * len(probs)]
    lines.append("| *(restatement as recalled: LM steps ok+rejected, RMSE px)* | " +
                 " | ".join(f"{base[n][0]}+{base[n][1]}, {base[n][2]:.6f}" for n, _, _ in probs) + " |")
    matter = {}
    for alt, what in bo.ALT_DETAILS.items():
        cells = []
        for name, arr, kw in probs:
            pr = H.to_oracle(arr)
            try:
                s = bo.solve(pr, bo.Options(alt=alt, **kw))
            except Exception as e:          # a flipped detail may make the linear system singular
                cells.append(f"fails ({type(e).__name__})"); matter[alt] = True; continue
            n_res = 2 * arr["obs_cam"].shape[0]
            b = base[name]
            d_steps = (s.n_successful - b[0], s.n_unsuccessful - b[1])
            d_rmse = math.sqrt(s.final_cost / n_res) - b[2]
            d_cam = max(np.abs(pr.cam_q - b[3]).max(), np.abs(pr.cam_t - b[4]).max())
            if d_steps == (0, 0) and abs(d_rmse) < 1e-12 and d_cam < 1e-12:
                cells.append("=")
            else:
                cells.append(f"{d_steps[0]:+d}/{d_steps[1]:+d}, {d_rmse:+.1e} px, cams {d_cam:.0e}")
                if d_steps != (0, 0) or abs(d_rmse) > 1e-6 or d_cam > 1e-5:
                    matter[alt] = True
        lines.append(f"| `{alt}` — {what} | " + " | ".join(cells) + " |")
    out = "\n".join(lines)
    out += ("\n\n`iteration_zero_counted` changes no parameter and no cost: it moves only the printed `Iterations :` line of "
            "`PrintSolverSummary` (`ba_solver.cc:22-25`) and the numerator of the bench metric by +1 per solve (config L: 13 -> 14 "
            "LM iterations, +7.7 % on `value`); bench.py counts LM steps only (iteration 0 not counted), the smaller number.")
    out += "\n\nDetails that move a result beyond the parity bar (steps, 1e-6 px, 1e-5 cameras) on at least one problem: " + \
           (", ".join(f"`{a}`" for a in bo.ALT_DETAILS if matter.get(a)) or "none") + \
           ".\nDetails that change nothing measurable on these problems (a real-Ceres golden of these problems could not pin them): " + \
           (", ".join(f"`{a}`" for a in bo.ALT_DETAILS if not matter.get(a)) or "none") + ".\n"
    print(out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(out)


if __name__ == "__main__":
    main()
