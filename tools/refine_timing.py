"""Developer tool: latency of one pose refinement (RegisterImage, pnp.cc:38-71): persistent kernel vs general engine."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from tests import helpers as H
from xrsfm_amd import capi
for n in (200, 1000, 5000):
    arr = H.make_pose_problem(n, seed=5, model=2)
    args = (2, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0])
    for label, env in (("kernel", None), ("engine", "1")):
        if env: os.environ["XRSFM_BA_REFINE_ENGINE"] = env
        else: os.environ.pop("XRSFM_BA_REFINE_ENGINE", None)
        for _ in range(5): q, t, s = capi.refine_pose(*args)
        t0 = time.perf_counter()
        for _ in range(200): q, t, s = capi.refine_pose(*args)
        dt = (time.perf_counter() - t0) / 200
        print(f"n={n:5d} {label}: {1e3*dt:.3f} ms per call, {s.n_successful}+{s.n_unsuccessful} steps, cost {s.initial_cost:.4e} -> {s.final_cost:.4e}")
os.environ.pop("XRSFM_BA_REFINE_ENGINE", None)
