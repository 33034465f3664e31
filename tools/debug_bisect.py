#!/usr/bin/env python3
"""Developer aid (GPU box): one 20000-point solve under each historical build of the library (xrsfm_amd/lib/libxrsfm_ba_hist_*.so)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch
from xrsfm_amd import capi
from tests import helpers as H
arr = H.make(100, 20000, 4, seed=2)
s = capi.solve(H.to_product(arr), capi.default_options())
ctx = capi.Context(H.to_product(arr))
out = ctx.debug_linearize(5.99, True)
y, S = ctx.debug_cholesky_solve(1e4, want_S=True)
bad = ~np.isfinite(S)
print("solve", s.n_successful, s.n_unsuccessful, s.termination_reason, "| S non-finite entries", int(bad.sum()), "rows with any", np.unique(np.nonzero(bad)[0] // 6)[:12], "y finite", bool(np.isfinite(y).all()),
      "| lin finite", all(bool(np.isfinite(np.asarray(v)).all()) for v in out.values() if hasattr(v, "shape")))
"""
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "xrsfm_amd", "lib", "libxrsfm_ba_hist_*.so")))
for lib in libs:
    e = dict(os.environ)
    if lib: e["XRSFM_BA_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=e, capture_output=True, text=True, cwd=ROOT)
    print(os.path.basename(lib) if lib else "HEAD", "->", r.stdout.strip()[-600:], r.stderr[-500:] if r.returncode else "")
