"""Shared test helpers: synthetic problems in both the product and the oracle representation."""
import numpy as np

from oracle import ba_oracle as bo
from xrsfm_amd import capi, synth

FIELDS = capi.ProblemArrays.FIELDS


def make(n_cams, n_points, k_obs=4, seed=100, **kw):
    d = synth.make_problem(n_cams, n_points, k_obs, seed=seed, **kw)
    return {k: d[k] for k in FIELDS}


def to_oracle(arr) -> bo.Problem:
    return bo.Problem(**{k: np.array(arr[k], copy=True) for k in FIELDS})


def to_product(arr) -> capi.ProblemArrays:
    return capi.ProblemArrays(**{k: np.array(arr[k], copy=True) for k in FIELDS})


def with_models(arr, seed=0):
    """Give every camera its own intrinsics, cycling through the 5 reference camera models."""
    rng = np.random.default_rng(seed)
    n = arr["cam_q"].shape[0]
    f, cx, cy = 718.856, 607.1928, 185.27157
    model = (np.arange(n) % 5).astype(np.int32)
    prm = np.zeros((n, 8))
    for i, m in enumerate(model):
        k = rng.normal(0, 0.02)
        if m == 0: prm[i, :3] = (f / 2, cx, cy)            # 2f quirk: halve f so the scene stays in view
        elif m == 1: prm[i, :4] = (f / 2, f / 2 * 1.01, cx, cy)
        elif m == 2: prm[i, :4] = (f, cx, cy, k)
        elif m == 3: prm[i, :5] = (f, f * 1.01, cx, cy, k)
        else: prm[i, :8] = (f, f * 0.99, cx, cy, k, rng.normal(0, 0.005), rng.normal(0, 1e-3), rng.normal(0, 1e-3))
    out = dict(arr)
    out["cam_intr"] = np.arange(n, dtype=np.int32)
    out["intr_model"] = model
    out["intr_params"] = prm
    return out


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
