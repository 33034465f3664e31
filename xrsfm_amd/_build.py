"""Build helper: compile the HIP library (gfx950) in-tree."""
from __future__ import annotations

import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "xrsfm_amd", "csrc")
LIBDIR = os.path.join(ROOT, "xrsfm_amd", "lib")
LIB = os.path.join(LIBDIR, "libxrsfm_ba.so")


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def lib_sources():
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "xrsfm_ba.h"))
    return srcs


# Hardening / probe builds of the same sources (tests/test_gpu_hardening.py, tools/backsub_waves_probe.py): name -> extra flags
VARIANTS = {
    # dead per-lane temporaries hold NaN instead of 0: results must not change by a bit — compared with "strict", the same
    # sources without -DXBA_POISON; both without FP contraction, so that the two builds evaluate identical expression trees
    # (with contraction the compiler is free to fuse differently when the initialisers differ: observed, 1e-10 relative)
    "poison": ["-DXBA_POISON", "-ffp-contract=off"],
    "strict": ["-ffp-contract=off"],
    "backsub_w5": ["-DXBA_BACKSUB_WAVES=5"],    # k_backsub register-allocated for 5 waves per SIMD (round-2 finding xi)
}


def variant_path(name: str) -> str:
    return os.path.join(LIBDIR, f"libxrsfm_ba_{name}.so")


def build_lib(force: bool = False, verbose: bool = False, variant: str | None = None) -> str:
    """hipcc --offload-arch=gfx950 -> xrsfm_amd/lib/libxrsfm_ba.so (cross-compiles without a GPU)."""
    srcs = lib_sources()
    if variant is not None:
        out = variant_path(variant)
        if not force and _newer(out, srcs):
            return out
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        os.makedirs(LIBDIR, exist_ok=True)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-deprecated-declarations",
               *VARIANTS[variant], "-o", out, os.path.join(CSRC, "xrsfm_ba.hip"), "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=ROOT)
        return out
    if not force and _newer(LIB, srcs):
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libxrsfm_ba.so")
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-deprecated-declarations",
           "-o", LIB, os.path.join(CSRC, "xrsfm_ba.hip"), "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=ROOT)
    return LIB
