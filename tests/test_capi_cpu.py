"""C-ABI library without a GPU: it builds for gfx950, loads, exports every symbol of include/xrsfm_ba.h, and the
compute entry points fail loudly (ENODEV) instead of falling back to a CPU path."""
import os
import re

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(lib):
    from xrsfm_amd import capi
    hdr = open(os.path.join(ROOT, "include", "xrsfm_ba.h")).read()
    declared = set(re.findall(r"\b(xrsfm_(?:ba|pg|tag)_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header(lib):
    """ctypes mirrors vs the C structs: sizes via a compile probe with gcc."""
    import ctypes, subprocess, tempfile
    from xrsfm_amd import capi
    src = '#include <stdio.h>\n#include "xrsfm_ba.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(xrsfm_ba_problem), sizeof(xrsfm_ba_options), sizeof(xrsfm_ba_summary), sizeof(xrsfm_pg_problem), sizeof(xrsfm_pg_options), sizeof(xrsfm_pg_summary), sizeof(xrsfm_tag_problem));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")], check=True)
        out = subprocess.run([os.path.join(d, "p")], capture_output=True, text=True, check=True).stdout.split()
    assert [int(x) for x in out] == [ctypes.sizeof(c) for c in (capi.CProblem, capi.COptions, capi.CSummary, capi.CPgProblem, capi.CPgOptions,
                                                               capi.CPgSummary, capi.CTagProblem)]


def test_default_options_are_the_reference_gba_settings(lib):
    from xrsfm_amd import capi
    o = capi.default_options()
    assert (o.max_iterations, o.function_tolerance, o.parameter_tolerance) == (50, 1e-5, 1e-6)   # ba_solver.cc:626-629
    assert (o.initial_radius, o.huber_a, o.gradient_tolerance) == (1e4, 5.99, 1e-10)


def test_no_cpu_fallback(lib):
    """Without a HIP device the product path refuses to run."""
    import torch
    from xrsfm_amd import capi
    if torch.cuda.is_available() and capi.device_count() > 0:
        pytest.skip("a GPU is present")
    assert capi.device_count() == 0
    with pytest.raises(RuntimeError, match="ENODEV"):
        capi.solve(H.to_product(H.make(6, 40, 3, seed=1)))


def test_product_code_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xrsfm_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cc", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("import oracle", "from oracle", "oracle/", "ba_oracle", "ba_cpu", "libba_cpu"):
                    assert pat not in txt, f"{f} references the test oracle ({pat})"


def test_streaming_kernels_do_not_spill(tmp_path):
    """The gfx950 code object of the streaming kernels — every instantiation of k_schur_pairs, k_linearize, k_backsub,
    k_schur_matvec, k_schur_prep, k_cost — has no spilled VGPRs and no private (scratch) segment.  Round 3 found a build of
    k_schur_pairs with 26 spilled VGPRs in its common path that produced wrong blocks of S from ~1300 tiles on,
    non-deterministically; the fix was to instantiate the kernel per operand height, and this test keeps
    it that way (hipcc cross-compiles the device code here, no GPU needed)."""
    import re
    import shutil
    import subprocess
    from xrsfm_amd import _build
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    asm = tmp_path / "xba.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                    "-Wno-deprecated-declarations", os.path.join(_build.CSRC, "xrsfm_ba.hip"), "-o", str(asm)], check=True, capture_output=True)
    text = asm.read_text()
    seen = 0
    for blk in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        # (round 6: k_refine_pose too — its 304-byte scratch segment cost the mapper a 20-28 ms scratch re-allocation on the first pose
        #  refinement after every large KGBA; k9_linearize of the bal9 mode keeps 24 bytes: not on the reference's path)
        if not any(k in name for k in ("k_schur_pairs", "k_linearize", "k_backsub", "k_schur_matvec", "k_schur_prep", "k_cost", "k_refine_pose")) or "k9_linearize" in name:
            continue
        seen += 1
        spills = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        assert spills == 0 and scratch == 0, (name, spills, scratch)
    assert seen >= 17          # 12 instantiations of k_schur_pairs + the others
