"""Seeded random problems against the oracle: camera counts, track lengths, missed detections, camera models, constant blocks,
both linear solvers.  Meant to reach the corners of the tile packing (single-track tiles, full tiles, Gram tiles with 2..10
cameras staged in one or two rounds, per-pair tiles, long tracks) that the hand-written cases may miss."""
import math

import numpy as np
import pytest

from oracle import ba_oracle as bo
from tests import helpers as H

CASES = list(range(40))


def _problem(seed):
    rng = np.random.default_rng(1000 + seed)
    n_cams = 72 if seed % 5 == 4 else int(rng.integers(5, 70))
    k_obs = int(rng.integers(2, 7)) if rng.random() < 0.65 else int(rng.integers(7, min(n_cams, 14) + 1))
    k_obs = min(k_obs, n_cams)
    n_pts = int(rng.integers(40, 900))
    mode = "unordered" if rng.random() < 0.3 else "sequential"
    dropout = float(rng.choice([0.0, 0.2, 0.4])) if (mode == "sequential" and k_obs > 2) else 0.0
    arr = H.make(n_cams, n_pts, k_obs, seed=2000 + seed, mode=mode, dropout=dropout, min_tri_angle_deg=0.5)
    if rng.random() < 0.4:
        arr = H.with_models(arr, seed=seed)
    if rng.random() < 0.3:                       # some constant points (LBA) / constant cameras
        arr["point_const"] = (rng.random(arr["points"].shape[0]) < 0.3).astype(np.uint8)
    if rng.random() < 0.3:
        cc = arr["cam_const"].copy(); cc[rng.integers(0, n_cams, 2)] |= 3; arr["cam_const"] = cc
    if seed % 5 == 4:                            # one long track (> 64 observations) when there are enough cameras
        if n_cams > 66:
            j = arr["points"].shape[0]
            arr["points"] = np.concatenate([arr["points"], arr["points"][:1] + 0.01])
            arr["point_const"] = np.concatenate([arr["point_const"], [0]]).astype(np.uint8)
            cams = np.arange(66, dtype=np.int32)
            arr["obs_cam"] = np.concatenate([arr["obs_cam"], cams])
            arr["obs_pt"] = np.concatenate([arr["obs_pt"], np.full(66, j, np.int32)])
            arr["obs_uv"] = np.concatenate([arr["obs_uv"], rng.uniform([100, 50], [1100, 300], (66, 2))])
    solver = int(rng.integers(0, 2))
    return arr, solver


@pytest.mark.gpu
@pytest.mark.parametrize("seed", CASES)
def test_random_problem_matches_oracle(lib, seed):
    from xrsfm_amd import capi
    arr, solver = _problem(seed)
    kw = dict(max_iterations=6)
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(linear_solver=solver, **kw))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-9 * s_ref.initial_cost
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful), (seed, solver)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [63, 96, 234, 264, 270])
def test_random_problem_exact_solver(lib, seed):
    """Seeds that tools/fuzz_extended.py found failing on the exact (Cholesky) path in round 2: regular tiles of 14-camera tracks
    (the per-camera sum of the diagonal terms over a tile's tracks read its last scatter position from a retired lane)."""
    from xrsfm_amd import capi
    arr, _ = _problem(seed)
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(linear_solver="exact", max_iterations=6))
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(linear_solver=1, max_iterations=6))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
