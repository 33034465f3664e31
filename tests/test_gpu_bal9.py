"""bal9 mode on the GPU (SURVEY.md section 8(d) "optional bal9 mode", BASELINE.json north_star "2x9 / 2x3 blocks"):
9-wide camera blocks {rotation, translation, f, k1, k2} of the extension camera model 5 against the oracle
(oracle/ba_oracle.py, pinned to torch.autograd for these Jacobians in tests/test_oracle_jacobian.py; the LM loop is the same
unpinned restatement of Ceres as for the 6-wide path).  The reference itself never frees intrinsics (ba_solver.cc:602-606)."""
import math

import numpy as np
import pytest

from oracle import ba_oracle as bo
from tests import helpers as H


def _scaled_lin(pr):
    cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    ci, pi = pr.obs_cam, pr.obs_pt
    sc_c = 1 / (1 + np.sqrt(bo._scatter_add(pr.cam_q.shape[0], ci, np.sum(Fc * Fc, axis=1))))
    sc_p = 1 / (1 + np.sqrt(bo._scatter_add(pr.points.shape[0], pi, np.sum(Ep * Ep, axis=1))))
    Fs = Fc * sc_c[ci][:, None, :]; Es = Ep * sc_p[pi][:, None, :]
    return cost, rt, Fs, Es, bo._Linearization(pr, rt, Fs, Es)


def _cases():
    a = H.make_bal9(12, 600, 4, seed=5)
    b = H.make_bal9(40, 2000, 6, seed=6, dropout=0.3, min_tri_angle_deg=0.5)
    b["point_const"] = (np.arange(2000) % 7 == 0).astype(np.uint8)
    cc = b["cam_const"].copy(); cc[5] &= 3; cc[9] &= 3; cc[11] |= 1; b["cam_const"] = cc        # two cameras with constant intrinsics, one constant rotation
    c = H.make_bal9(30, 900, 12, seed=7, mode="unordered", min_tri_angle_deg=0.5)                      # dense reduced matrix, tracks of 12
    d = H.make_bal9(72, 60, 68, seed=8, mode="unordered", min_tri_angle_deg=0.5)                      # long items (> 64 observations)
    return {"seq": a, "ragged_consts": b, "unordered": c, "long": d}


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["seq", "ragged_consts", "unordered", "long"])
def test_bal9_linearisation_and_step_match_oracle(lib, name):
    import scipy.linalg as sla
    from xrsfm_amd import capi
    arr = _cases()[name]
    pr = H.to_oracle(arr)
    cost, rt, Fs, Es, lin = _scaled_lin(pr)
    radius = 3e3
    ctx = capi.Context(H.to_product(arr))
    out = ctx.debug_wide(5.99, radius)
    ctx.close()
    assert abs(out["cost"] - cost) <= 1e-12 * cost
    assert H.rel_err(out["r"], rt) < 1e-12 and H.rel_err(out["Jc"], Fs) < 1e-11 and H.rel_err(out["Jp"], Es) < 1e-11
    assert H.rel_err(out["Hcc_diag"], np.einsum("nii->ni", lin.Hcc)) < 1e-11 and H.rel_err(out["gc"], lin.gc) < 1e-11
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    yc, yp, _ = bo._solve_exact(pr, lin, Dc2, Dp2)
    # normwise backward error of the step against the oracle's reduced system is checked through the solution itself here:
    # the systems are well conditioned enough at this size (constant blocks have unit pivots on both sides)
    assert H.rel_err(out["y"], yc) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("k_cams", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("ragged", [False, True])
def test_bal9_gram_tiles_of_every_camera_count(lib, k_cams, ragged):
    """Round 4 (k9_pairs_gram): the S assembly of 9-wide blocks through Gram tiles.  Every camera count a tile can have —
    C = 2..7 cameras = 18..63 operand rows = all four operand heights NI, dense tracks (every track sees every camera) and
    ragged ones (missed detections: zero-filled operand, per-camera sums over a subset of the lanes) — plus C = 8, which must
    take the per-pair path: the step of the FIRST linearisation against the oracle's exact solve, and three LM iterations
    against the oracle (same decisions, cost, cameras, intrinsics)."""
    from xrsfm_amd import capi
    kw = dict(mode="unordered", min_tri_angle_deg=0.5)
    n_cams = k_cams + (3 if ragged else 0)
    if ragged:
        kw["dropout"] = 0.3
    arr = H.make_bal9(n_cams, 260, min(k_cams + (2 if ragged else 0), n_cams), seed=300 + 10 * k_cams + int(ragged), **kw)
    g = capi.debug_pack_gram(H.to_product(arr))
    if not ragged:
        assert (g["gram_tiles"] > 0) == (k_cams <= 7) and (k_cams > 7 or g["max_cams"] == k_cams), g
    pr = H.to_oracle(arr)
    cost, rt, Fs, Es, lin = _scaled_lin(pr)
    radius = 2e3
    ctx = capi.Context(H.to_product(arr))
    out = ctx.debug_wide(5.99, radius)
    ctx.close()
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    yc, yp, _ = bo._solve_exact(pr, lin, Dc2, Dp2)
    assert H.rel_err(out["y"], yc) < 1e-7
    pr2 = H.to_oracle(arr)
    s_ref = bo.solve(pr2, bo.Options(max_iterations=3))
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=3))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert max(np.abs(prod.cam_q - pr2.cam_q).max(), np.abs(prod.cam_t - pr2.cam_t).max()) < 1e-5
    assert np.abs(prod.intr_params[:, 0] / pr2.intr_params[:, 0] - 1).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["seq", "ragged_consts", "unordered", "long"])
def test_bal9_full_solve_parity(lib, name):
    from xrsfm_amd import capi
    arr = _cases()[name]
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(max_iterations=12))
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=12))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY
    assert s.num_effective_params == s_ref.num_effective_params and s.num_residuals == s_ref.num_residuals
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-9 * s_ref.initial_cost
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
    var = (arr["cam_const"] & 4) != 0
    assert np.abs(prod.intr_params[var, 0] / pr.intr_params[var, 0] - 1).max() < 1e-6          # f
    assert np.abs(prod.intr_params[var, 1:3] - pr.intr_params[var, 1:3]).max() < 1e-5          # k1, k2
    assert np.array_equal(prod.intr_params[~var], arr["intr_params"][~var]) and np.array_equal(prod.intr_params[:, 3:], arr["intr_params"][:, 3:])
    assert np.abs(prod.intr_params[var, :3] - arr["intr_params"][var, :3]).max() > 1e-4       # the intrinsics took part


@pytest.mark.gpu
def test_bal9_context_reruns_and_refuses_what_is_not_implemented(lib):
    from xrsfm_amd import capi
    arr = _cases()["seq"]
    ctx = capi.Context(H.to_product(arr))
    s1 = ctx.run(capi.default_options())
    q1, t1, P1 = ctx.download(); i1 = ctx.download_intrinsics()
    ctx.reset()
    s2 = ctx.run(capi.default_options())
    q2, t2, P2 = ctx.download(); i2 = ctx.download_intrinsics()
    assert s1.final_cost == s2.final_cost and np.array_equal(q1, q2) and np.array_equal(P1, P2) and np.array_equal(i1, i2)
    ctx.reset()
    with pytest.raises(RuntimeError):                 # implicit-Schur PCG is 6-wide only
        ctx.run(capi.default_options(linear_solver=capi.SOLVER_PCG))
    ctx.close()
    # bit 2 with one of the reference's camera models, or with a shared intrinsics entry: EINVAL
    bad = dict(arr); bad["intr_model"] = np.full(12, 2, np.int32)
    with pytest.raises(RuntimeError):
        capi.solve(H.to_product(bad))
    bad = dict(arr); bad["cam_intr"] = np.zeros(12, np.int32)
    with pytest.raises(RuntimeError):
        capi.solve(H.to_product(bad))
    # without the bit the same cameras are an ordinary 6-wide problem of model 5: intrinsics untouched, oracle parity
    arr6 = dict(arr); arr6["cam_const"] = (arr["cam_const"] & 3).astype(np.uint8)
    pr = H.to_oracle(arr6); s_ref = bo.solve(pr, bo.Options())
    prod = H.to_product(arr6); s = capi.solve(prod)
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost and np.array_equal(prod.intr_params, arr["intr_params"])


@pytest.mark.gpu
def test_bal9_golden(lib):
    """tests/golden/wide_bal9.npz (make_golden_bal9.py): initial 2x9 / 2x3 blocks, LM step counts, final state incl. intrinsics."""
    import os
    from xrsfm_amd import capi
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wide_bal9.npz"))
    arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    mi, ft, pt, rad = z["opt"]
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=int(mi), function_tolerance=float(ft), parameter_tolerance=float(pt), initial_radius=float(rad)))
    assert (s.n_successful, s.n_unsuccessful) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(s.final_cost - float(z["final_cost"])) <= 1e-8 * float(z["final_cost"])
    assert np.abs(prod.cam_q - z["out_cam_q"]).max() < 1e-5 and np.abs(prod.cam_t - z["out_cam_t"]).max() < 1e-5
    assert np.abs(prod.intr_params[:, 0] / z["out_intr"][:, 0] - 1).max() < 1e-6 and np.abs(prod.intr_params[:, 1:3] - z["out_intr"][:, 1:3]).max() < 1e-5


def _bal9_fuzz_problem(seed):
    """Random bal9 problem: 4-48 cameras, sequential or unordered visibility, ragged tracks, a random subset of cameras with
    constant intrinsics / rotation / translation, some constant points."""
    rng = np.random.default_rng(50000 + seed)
    n_cams = int(rng.integers(4, 49))
    mode = "unordered" if rng.random() < 0.4 else "sequential"
    k = int(rng.integers(2, min(n_cams, 9) + 1))
    n_pts = int(rng.integers(30, 400))
    kw = dict(mode=mode, min_tri_angle_deg=0.5)
    if rng.random() < 0.5:
        kw["dropout"] = float(rng.uniform(0.1, 0.4))
    arr = H.make_bal9(n_cams, n_pts, k, seed=seed, **kw)
    cc = arr["cam_const"].copy()
    drop = rng.random(n_cams) < 0.2
    cc[drop] &= 3
    cc[rng.random(n_cams) < 0.1] |= 1
    cc[rng.random(n_cams) < 0.1] |= 2
    arr["cam_const"] = cc
    arr["point_const"] = (rng.random(arr["points"].shape[0]) < 0.1).astype(np.uint8)
    return arr


@pytest.mark.gpu
def test_bal9_fuzz(lib):
    """60 random bal9 problems against the numpy oracle at the fuzz test's strict bar (same LM decisions, RMSE 1e-6 px, cameras
    1e-5, intrinsics 1e-6 relative / 1e-5); there is no second CPU restatement for width 9, so nothing may be 'explained':
    at most one problem of the slice may miss the bar, and then only by LM trajectory (step counts), never by a wrong cost."""
    from xrsfm_amd import capi
    misses = []
    n_run = 0
    for seed in range(60):
        try:
            arr = _bal9_fuzz_problem(seed)
        except (ValueError, RuntimeError):
            continue
        pr = H.to_oracle(arr)
        s_ref = bo.solve(pr, bo.Options(max_iterations=6))
        prod = H.to_product(arr)
        s = capi.solve(prod, capi.default_options(max_iterations=6))
        n_run += 1
        n_res = 2 * arr["obs_cam"].shape[0]
        assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-9 * s_ref.initial_cost, seed
        var = (arr["cam_const"] & 4) != 0
        ok = ((s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
              and abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
              and max(np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max()) < 1e-5
              and (not var.any() or (np.abs(prod.intr_params[var, 0] / pr.intr_params[var, 0] - 1).max() < 1e-6
                                     and np.abs(prod.intr_params[var, 1:3] - pr.intr_params[var, 1:3]).max() < 1e-5)))
        if not ok:
            misses.append((seed, s.n_successful, s.n_unsuccessful, s_ref.n_successful, s_ref.n_unsuccessful, s.final_cost, s_ref.final_cost))
    assert n_run >= 45 and len(misses) <= 1, misses


@pytest.mark.gpu
@pytest.mark.parametrize("n_cams,n_pts,k_obs,dropout", [(200, 20000, 4, 0.0), (150, 12000, 8, 0.35)])
def test_bal9_mid_size_matches_c_restatement(lib, n_cams, n_pts, k_obs, dropout):
    """Hundreds of tiles of 9-wide blocks — beyond what the numpy oracle solves in seconds — against the C restatement
    (oracle/ba_cpu.c with CW = 9): same LM decisions, RMSE 1e-6 px, cameras 1e-5 (translations of a 150-200-camera chain drift
    along the gauge: 1e-4), intrinsics 1e-6 relative / 1e-5; twice, bit-identical."""
    from oracle import ba_cpu
    from xrsfm_amd import capi
    if not ba_cpu.available():
        ba_cpu.build()
    arr = H.make_bal9(n_cams, n_pts, k_obs, seed=2, dropout=dropout, min_tri_angle_deg=1.0)
    cp = {k: np.array(v, copy=True) for k, v in arr.items()}
    sc = ba_cpu.solve(cp, threads=8)
    outs = []
    for _ in range(2):
        prod = H.to_product(arr)
        s = capi.solve(prod)
        outs.append((s.final_cost, prod.cam_q.copy(), prod.cam_t.copy(), prod.intr_params.copy()))
    assert outs[0][0] == outs[1][0] and all(np.array_equal(a, b) for a, b in zip(outs[0][1:], outs[1][1:]))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (sc["n_successful"], sc["n_unsuccessful"])
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(sc["final_cost"] / n_res)) < 1e-6
    assert np.abs(prod.cam_q - cp["cam_q"]).max() < 1e-6 and np.abs(prod.cam_t - cp["cam_t"]).max() < 1e-4
    assert np.abs(prod.intr_params[:, 0] / cp["intr_params"][:, 0] - 1).max() < 1e-6
    assert np.abs(prod.intr_params[:, 1:3] - cp["intr_params"][:, 1:3]).max() < 1e-5
