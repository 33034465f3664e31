"""The off-suite fuzz mismatches of round 5 (tools/fuzz_extended.py: seeds 6449, 6649, 6654, 6669 on the exact path, 4494 through
the PCG; profiles/r05_experiments.md section 11) classified BY TEST instead of by assertion.

All of them are the `cams 72` class of tests/test_gpu_fuzz.py: one 66-observation track with random image coordinates, costs of
1e5-1e12, a reduced camera matrix whose solve amplifies rounding by many orders.  The claim "known ill-conditioned class" is a
statement about the ORACLE: on exactly these problems the numpy restatement must disagree with ITSELF after the points and
observations are relabelled (another summation order, the same mathematics) and with the C restatement by at least as much as
the HIP path disagrees with it — and on well-conditioned seeds of the same generator it must agree with itself to 1e-10.
The CPU test holds the oracle half; the GPU test adds: the HIP result lies within that self-disagreement, same LM decisions."""
import math

import numpy as np
import pytest

from oracle import ba_cpu
from oracle import ba_oracle as bo
from tests import helpers as H
from tests.test_gpu_fuzz import _problem

ILL = [6449, 6649, 6654, 6669, 9124, 9284]        # exact path (the last two: round 6's fresh-seed fuzz, tools/runs/r06_fuzz1.sh)
ILL_PCG = [4494, 8204, 8294, 8304]                # through the PCG (round 6's fuzz also met 8224 and 8264 — costs of 1e12-1e13, on which a
                                                  # truncated PCG may even take another LM decision than an exact solve: not asserted here)
WELL = [4, 9]                         # the same 72-camera class, well conditioned


def relabelled(arr, seed=1):
    """The same problem with the points renumbered and the observation list shuffled."""
    rng = np.random.default_rng(seed)
    a = {k: np.array(v, copy=True) for k, v in arr.items()}
    n_pt = a["points"].shape[0]
    perm = rng.permutation(n_pt)
    inv = np.empty(n_pt, np.int64); inv[perm] = np.arange(n_pt)
    a["points"] = a["points"][perm]; a["point_const"] = a["point_const"][perm]
    a["obs_pt"] = inv[a["obs_pt"]].astype(np.int32)
    o = rng.permutation(a["obs_cam"].shape[0])
    for k in ("obs_cam", "obs_pt", "obs_uv"):
        a[k] = a[k][o]
    return a


def oracle_self_spread(arr):
    """(camera spread, relative cost spread, decisions) of the oracle over {as given, relabelled, C restatement}."""
    runs = []
    for a in (arr, relabelled(arr)):
        pr = H.to_oracle(a)
        s = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
        runs.append((pr.cam_q.copy(), pr.cam_t.copy(), s.final_cost, (s.n_successful, s.n_unsuccessful)))
    if ba_cpu.available():
        c = {k: np.array(v, copy=True) for k, v in arr.items()}
        sc = ba_cpu.solve(c, max_iterations=6, threads=1)
        runs.append((c["cam_q"], c["cam_t"], sc["final_cost"], (sc["n_successful"], sc["n_unsuccessful"])))
    q0, t0, c0, d0 = runs[0]
    cam = max(max(np.abs(q - q0).max(), np.abs(t - t0).max()) for q, t, _, _ in runs[1:])
    cost = max(abs(c - c0) / c0 for _, _, c, _ in runs[1:])
    return cam, cost, d0, runs[0]


@pytest.mark.parametrize("seed", ILL + ILL_PCG)
def test_mismatch_seeds_are_oracle_self_mismatches(seed):
    arr, _ = _problem(seed)
    assert arr["cam_q"].shape[0] == 72
    cam, cost, _, _ = oracle_self_spread(arr)
    # the oracle cannot reproduce its own cameras to the parity bar's neighbourhood, or its own cost to 1e-9
    assert cam > 1e-7 or cost > 1e-9, (seed, cam, cost)


@pytest.mark.parametrize("seed", WELL)
def test_well_conditioned_seeds_of_the_same_class_agree_with_themselves(seed):
    arr, _ = _problem(seed)
    assert arr["cam_q"].shape[0] == 72
    cam, cost, _, _ = oracle_self_spread(arr)
    assert cam < 1e-10 and cost < 1e-12, (seed, cam, cost)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,solver", [(s, 1) for s in ILL] + [(s, 0) for s in ILL_PCG])
def test_hip_result_lies_within_the_oracles_self_disagreement(lib, seed, solver):
    from xrsfm_amd import capi
    arr, _ = _problem(seed)
    cam, cost, dec, (q0, t0, c0, _) = oracle_self_spread(arr)
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(linear_solver=solver, max_iterations=6))
    assert (s.n_successful, s.n_unsuccessful) == dec, (seed, dec)
    d_cam = max(np.abs(prod.cam_q - q0).max(), np.abs(prod.cam_t - t0).max())
    d_cost = abs(s.final_cost - c0) / c0
    # within 10x of what the oracle's own restatements differ by (+ the parity bar: 1e-5 on cameras, 1e-6 px on the RMSE)
    n_res = 2 * arr["obs_cam"].shape[0]
    assert d_cam <= 10.0 * cam + 1e-5, (seed, d_cam, cam)
    assert d_cost <= 10.0 * cost + 2e-6 / math.sqrt(c0 / n_res), (seed, d_cost, cost)
