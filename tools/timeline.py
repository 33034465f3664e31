#!/usr/bin/env python3
"""Developer aid: cycle stamps of sampled waves inside the streaming kernels (library built with -DXBA_TIMELINE into
/tmp; the shipped library has no stamps).  usage (GPU box): python tools/timeline.py [config]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "L"
    # XBA_TL_FLAGS: extra compiler flags of the instrumented build (e.g. -DXBA_POTRF_PV=0), XBA_TL_TAG: a name for that build
    tag = os.environ.get("XBA_TL_TAG", "tl")
    lib = os.path.join(ROOT, "tools", "_tl", f"libxrsfm_ba_{tag}.so")
    srcs = [os.path.join(ROOT, "xrsfm_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "xrsfm_amd", "csrc")) if f.endswith((".h", ".hip"))]
    if not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in srcs):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DXBA_TIMELINE", "-Wno-unused-value",
                        "-Wno-deprecated-declarations", *os.environ.get("XBA_TL_FLAGS", "").split(), "-o", lib,
                        os.path.join(ROOT, "xrsfm_amd", "csrc", "xrsfm_ba.hip"), "-ldl"], check=True)
    from xrsfm_amd import capi, synth
    L = capi.load(lib)
    d = synth.make_problem(**synth.CONFIGS[cfg])
    prob = capi.ProblemArrays(**{k: d[k] for k in capi.ProblemArrays.FIELDS})
    ctx = capi.Context(prob)
    ctx.run(capi.default_options(max_iterations=3))
    out = np.zeros((3, 64, 16), dtype=np.uint64)
    L.xrsfm_ba_debug_stamps.argtypes = [C.c_void_p]
    assert L.xrsfm_ba_debug_stamps(out.ctypes.data) == 0
    for kern, name in enumerate(["k_schur_pairs", "k_lv_factor", "k_backsub"]):
        st = out[kern].astype(np.int64)
        ok = st[:, 0] > 0
        if not ok.any():
            continue
        st = st[ok]
        n = 16
        rel = (st[:, :n] - st[:, :1])
        rel[st[:, :n] == 0] = -1
        print(name, "samples", len(st), "cycles from wave start (median / p10 / p90) at each stamp (100 MHz counter? see deltas):")
        for i in range(n):
            col = rel[:, i]
            print(f"  stamp {i}: {np.median(col):9.0f} {np.percentile(col, 10):9.0f} {np.percentile(col, 90):9.0f}")
    ctx.close()


if __name__ == "__main__":
    main()
