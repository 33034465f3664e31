#!/usr/bin/env python3
"""Developer aid (GPU box): the calls of 5 ms and more of the mapper-shaped replay (tests/shim/mapper_main, MAPPER_TRACE_SLOW=1),
with their class (0 GBA, 1 LBA, 2 KGBA, 3 pose refinement, 4 per-frame filter, 5 whole-map filter) and frame."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrsfm_amd import mapper_replay
arr = mapper_replay.sequence_problem()
exe = mapper_replay.build()
with tempfile.TemporaryDirectory() as td:
    inp, out = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
    mapper_replay.dump(arr, inp)
    p = subprocess.run([exe, inp, out, "2"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=1800,
                       env=dict(os.environ, MAPPER_TRACE_SLOW="1", XRSFM_BA_TRACE_CALLS=os.environ.get("XRSFM_BA_TRACE_CALLS", "")))
    all_lines = p.stderr.splitlines()
    lines = [ln for ln in all_lines if ln.startswith("[mapper_main]")]
    print("\n".join(lines[len(lines) // 2:]))          # the second replay
    # with XRSFM_BA_TRACE_CALLS=1: the pose refinements of 20 LM steps or more, or 5 ms or more inside xrsfm_ba_refine_pose
    import re
    for ln in all_lines:
        m = re.search(r"RefineFramePose frame (\d+): .* LM (\d+)\+(\d+), .* px, ([0-9.]+) ms", ln)
        if m and (int(m.group(2)) + int(m.group(3)) >= 20 or float(m.group(4)) >= 5.0):
            print(ln)
        if ln.startswith("[xrsfm_ba_refine_pose] slow call"):
            print(ln)
