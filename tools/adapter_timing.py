"""Developer tool: what xrsfm::BASolver::GBA costs around the solve at BASELINE.json config 4's size — the source-compatible adapter
(compat/optimization/ba_solver.cc) on the test shim of base/map.h: Map -> SoA (FlatProblem), xrsfm_ba_solve (create + set-up +
solve + download), SoA -> Map.  The process calls GBA twice; the second (warm) call is the one to read.
usage (GPU box): python tools/adapter_timing.py [config]"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from xrsfm_amd import capi, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "L"
d = synth.make_problem(**synth.CONFIGS[cfg])
exe = os.path.join(ROOT, "tests", "shim", "_build", "adapter_main")
if not os.path.exists(exe):
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "shim")], check=True, capture_output=True)
with tempfile.TemporaryDirectory() as tmp:
    inp, out = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([d["cam_q"].shape[0], d["points"].shape[0], d["obs_cam"].shape[0], d["intr_model"].shape[0]], np.int32).tobytes())
        for k, dt in (("cam_q", "f8"), ("cam_t", "f8"), ("cam_intr", "i4"), ("intr_model", "i4"), ("intr_params", "f8"),
                      ("points", "f8"), ("obs_cam", "i4"), ("obs_pt", "i4"), ("obs_uv", "f8")):
            f.write(np.ascontiguousarray(d[k], dtype=dt).tobytes())
    p = subprocess.run([exe, inp, out, "gba_timed"], capture_output=True, text=True, timeout=600, env=dict(os.environ, XRSFM_BA_TRACE_CALLS="1"))
    print("\n".join(l for l in p.stderr.splitlines() if l.startswith("[")))
    print("exit", p.returncode)
