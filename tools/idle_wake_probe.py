#!/usr/bin/env python3
"""Developer aid (GPU box): how long the first small call takes after the GPU has been idle for a while, (a) a torch op, (b) the library's
own pose refinement (two kernels, pinned staging, no copy engine) — with the host sleeping or busy meanwhile.  Background: the mapper
replay's pose refinements take 20-30 ms whenever they are the first GPU work after a KGBA + whole-map filter (10-15 ms of host-only
time); the kernel itself runs 0.1 ms."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrsfm_amd import capi, synth          # noqa: E402

x = torch.zeros(1024, device="cuda")
torch.cuda.synchronize()
d = synth.make_problem(8, 400, 4, seed=3)
P = d["points"][:300].copy()
q = d["cam_q"][0].copy(); t = d["cam_t"][0].copy()
M = np.zeros((3, 3)); qq = q / np.linalg.norm(q)
xq, yq, zq, wq = qq
R = np.array([[1 - 2 * (yq * yq + zq * zq), 2 * (xq * yq - zq * wq), 2 * (xq * zq + yq * wq)],
              [2 * (xq * yq + zq * wq), 1 - 2 * (xq * xq + zq * zq), 2 * (yq * zq - xq * wq)],
              [2 * (xq * zq - yq * wq), 2 * (yq * zq + xq * wq), 1 - 2 * (xq * xq + yq * yq)]])
Pc = P @ R.T + t
f = float(d["intr_params"][0][0])
uv = np.stack([f * Pc[:, 0] / Pc[:, 2], f * Pc[:, 1] / Pc[:, 2]], axis=1) + 0.3


def refine():
    capi.refine_pose(int(d["intr_model"][0]), d["intr_params"][0], P, uv, q.copy(), t.copy())


def busy(ms):
    t0 = time.perf_counter()
    a = np.random.rand(200, 200)
    while (time.perf_counter() - t0) * 1e3 < ms:
        a = a @ a * 1e-3


refine(); refine()
for mode, wait in (("sleep", lambda ms: time.sleep(ms * 1e-3)), ("busy host", busy)):
    for idle_ms in (0, 5, 10, 20, 50):
        ts_t, ts_r = [], []
        for _ in range(5):
            (x + 1).sum().item(); refine()
            wait(idle_ms)
            t0 = time.perf_counter(); refine(); ts_r.append((time.perf_counter() - t0) * 1e3)
            (x + 1).sum().item()
            wait(idle_ms)
            t0 = time.perf_counter(); (x + 1).sum().item(); ts_t.append((time.perf_counter() - t0) * 1e3)
        print(f"{mode:9s} idle {idle_ms:3d} ms -> refine_pose median {sorted(ts_r)[2]:.3f} max {max(ts_r):.3f} ms | torch op median {sorted(ts_t)[2]:.3f} max {max(ts_t):.3f} ms")
