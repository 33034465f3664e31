#!/usr/bin/env python3
"""Where bench/ceres_harness exists (a box with Ceres < 2.2 + Eigen): run it on the seeded problems of tests/golden/make_golden.py
and write tests/golden/ceres_<name>.npz (inputs seed + Ceres' final state, step counts and costs).  tests/test_oracle_golden.py
picks such files up and checks the oracle, the C restatement and — with a GPU — the HIP path against them.  No such file is
committed yet: Ceres is absent from every box this repository has seen, so parity is UNPINNED against the real reference."""
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXE = os.path.join(ROOT, "bench", "_build", "ceres_harness")

CASES = {"gba_kitti": dict(n_cams=10, n_points=400, k_obs=4, seed=103), "gba_models": dict(n_cams=10, n_points=400, k_obs=4, seed=103, models=True),
         "S": dict(n_cams=100, n_points=50_000, k_obs=4, seed=2)}


def dump(arr, path):
    with open(path, "wb") as f:
        f.write(struct.pack("4i", arr["cam_q"].shape[0], arr["points"].shape[0], arr["obs_cam"].shape[0], arr["intr_model"].shape[0]))
        for k, dt in (("cam_q", "f8"), ("cam_t", "f8"), ("cam_intr", "i4"), ("intr_model", "i4"), ("intr_params", "f8"),
                      ("points", "f8"), ("obs_cam", "i4"), ("obs_pt", "i4"), ("obs_uv", "f8"), ("cam_const", "u1"), ("point_const", "u1")):
            f.write(np.ascontiguousarray(arr[k], dtype=dt).tobytes())


def main():
    if not os.path.exists(EXE):
        raise SystemExit(f"{EXE} is missing: cmake -S bench -B bench/_build && cmake --build bench/_build (needs Ceres < 2.2)")
    from tests import helpers as H
    for name, kw in CASES.items():
        kw = dict(kw); models = kw.pop("models", False)
        arr = H.make(kw.pop("n_cams"), kw.pop("n_points"), kw.pop("k_obs"), **kw)
        if models:
            arr = H.with_models(arr, seed=4)
        inp, out = f"/tmp/ceres_{name}.in", f"/tmp/ceres_{name}.out"
        dump(arr, inp)
        subprocess.run([EXE, inp, out], check=True)
        raw = open(out, "rb").read()
        term, ns, nu = struct.unpack("3i", raw[:12]); c0, c1, secs = struct.unpack("3d", raw[12:36])
        nc, npt = arr["cam_q"].shape[0], arr["points"].shape[0]
        q = np.frombuffer(raw, "f8", 4 * nc, 36).reshape(nc, 4); t = np.frombuffer(raw, "f8", 3 * nc, 36 + 32 * nc).reshape(nc, 3)
        P = np.frombuffer(raw, "f8", 3 * npt, 36 + 56 * nc).reshape(npt, 3)
        np.savez(os.path.join(ROOT, "tests", "golden", f"ceres_{name}.npz"), termination=term, n_successful=ns, n_unsuccessful=nu,
                 initial_cost=c0, final_cost=c1, seconds=secs, out_cam_q=q, out_cam_t=t, out_points=P, **{f"in_{k}": v for k, v in arr.items()})
        print(name, "steps", ns, nu, "cost", c0, "->", c1, f"{secs:.3f} s")


if __name__ == "__main__":
    main()
