#!/usr/bin/env python3
"""Developer aid (GPU box): at which problem size / with which schedule switch does a solve go wrong?"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch
from xrsfm_amd import capi
from tests import helpers as H
for (nc, npts) in [(100, 2000), (100, 5000), (100, 20000), (100, 50000)]:
    arr = H.make(nc, npts, 4, seed=2)
    p = H.to_product(arr)
    s = capi.solve(p, capi.default_options())
    ctx = capi.Context(H.to_product(arr))
    ctx.debug_linearize(5.99, True)
    y1, _ = ctx.debug_cholesky_solve(1e4)
    y2, S = ctx.debug_cholesky_solve(1e4, want_S=True)
    st = capi.debug_pack(H.to_product(arr)); g = capi.debug_pack_gram(H.to_product(arr))
    print(nc, npts, "solve:", s.n_successful, s.n_unsuccessful, s.termination_reason, "| fused vs materialised y: max diff", float(np.abs(y1 - y2).max()), "max |y|", float(np.abs(y2).max()),
          "finite", bool(np.isfinite(y1).all()), "| tiles", st["tiles"], "regular", st["regular_tiles"], "small/big/other", g["items_small"], g["items_big"], g["items_other"])
    ctx.close()
"""
for env in ({}, {"XRSFM_BA_PREP_FUSED": "0"}, {"XRSFM_BA_FUSED": "0"}, {"XRSFM_BA_PACKED": "1"}, {"XRSFM_BA_LOOKAHEAD": "0"}):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=e, capture_output=True, text=True, cwd=ROOT)
    print("==", env); print(r.stdout[-3000:]); print(r.stderr[-800:] if r.returncode else "")
