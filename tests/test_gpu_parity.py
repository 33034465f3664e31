"""GPU parity tests (call through the C-ABI; oracle = oracle/ba_oracle.py, parity unpinned vs real Ceres).

Tolerances: kernel-level blocks 1e-11 relative (FP64, different summation order);
full solves: reference-style RMSE within 1e-6 px, camera parameters within 1e-5
(BASELINE.json north_star).
"""
import math

import numpy as np
import pytest

from oracle import ba_oracle as bo
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _lin_oracle(pr, use_scaling):
    cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    ci, pi = pr.obs_cam, pr.obs_pt
    if use_scaling:
        sc_c = 1 / (1 + np.sqrt(bo._scatter_add(pr.cam_q.shape[0], ci, np.sum(Fc * Fc, axis=1))))
        sc_p = 1 / (1 + np.sqrt(bo._scatter_add(pr.points.shape[0], pi, np.sum(Ep * Ep, axis=1))))
        Fc = Fc * sc_c[ci][:, None, :]; Ep = Ep * sc_p[pi][:, None, :]
    return cost, rt, Fc, Ep, bo._Linearization(pr, rt, Fc, Ep)


@pytest.mark.parametrize("use_scaling", [False, True])
@pytest.mark.parametrize("variant", ["kitti", "models", "behind", "consts"])
def test_linearize_blocks(lib, variant, use_scaling):
    from xrsfm_amd import capi
    arr = H.make(8, 300, 4, seed=101)
    if variant == "models":
        arr = H.with_models(arr, seed=3)
    if variant == "behind":      # some points behind / too close to their cameras -> clamp branch (12,12), J = 0
        arr["points"][::7] += np.array([0.0, 0.0, -60.0])
    if variant == "consts":
        arr["cam_const"][:] = 0; arr["cam_const"][2] = 3; arr["cam_const"][5] = 1
        arr["point_const"][::3] = 1
    pr = H.to_oracle(arr)
    ctx = capi.Context(H.to_product(arr))
    out = ctx.debug_linearize(5.99, use_scaling)
    cost, rt, Fc, Ep, lin = _lin_oracle(pr, use_scaling)
    assert abs(out["cost"] - cost) <= 1e-12 * cost
    assert H.rel_err(out["r"], rt) < 1e-12
    assert H.rel_err(out["Jc"], Fc) < 1e-11
    assert H.rel_err(out["Jp"], Ep) < 1e-11
    Hpp = lin.Hpp
    got = out["Hpp"]
    ref = np.stack([Hpp[:, 0, 0], Hpp[:, 0, 1], Hpp[:, 0, 2], Hpp[:, 1, 1], Hpp[:, 1, 2], Hpp[:, 2, 2]], axis=1)
    assert H.rel_err(got, ref) < 1e-11
    assert H.rel_err(out["gp"], lin.gp) < 1e-11
    assert H.rel_err(out["Hcc_diag"], np.einsum("nii->ni", lin.Hcc)) < 1e-11
    assert H.rel_err(out["gc"], lin.gc) < 1e-11
    if variant == "behind":
        assert (np.abs(out["r"] - 12.0 * math.sqrt(5.99 / math.sqrt(288.0))).max(axis=1) < 1e-12).sum() > 10
    ctx.close()


@pytest.mark.parametrize("k_obs,n_pts", [(4, 300), (3, 257), (70, 12)])
def test_schur_product(lib, k_obs, n_pts):
    """y = S x and rhs b against the dense Schur complement of the oracle (long tracks: k_obs=70 > 64 lanes)."""
    from xrsfm_amd import capi
    n_cams = 8 if k_obs < 10 else 80
    arr = H.make(n_cams, n_pts, k_obs, seed=102, min_tri_angle_deg=0.5, mode="unordered" if k_obs > 10 else "sequential")
    pr = H.to_oracle(arr)
    ctx = capi.Context(H.to_product(arr))
    ctx.debug_linearize(5.99, True)
    _, _, _, _, lin = _lin_oracle(pr, True)
    radius = 1e4
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
    ci, pi = pr.obs_cam, pr.obs_pt
    rng = np.random.default_rng(5)
    x = rng.normal(size=(n_cams, 6))
    v = np.einsum("nki,ni->nk", lin.Fs, x[ci])
    tj = bo._scatter_add(n_pts, pi, np.einsum("nki,nk->ni", lin.Es, v))
    u = np.einsum("nij,nj->ni", Hinv, tj)
    zz = v - np.einsum("nki,ni->nk", lin.Es, u[pi])
    y_ref = Dc2 * x + bo._scatter_add(n_cams, ci, np.einsum("nki,nk->ni", lin.Fs, zz))
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
    b_ref = lin.gc - bo._scatter_add(n_cams, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    y, b = ctx.debug_schur_product(radius, x)
    assert H.rel_err(y, y_ref) < 1e-10
    assert H.rel_err(b, b_ref) < 1e-10
    ctx.close()


def _solve_both(arr, opt_kw=None, oracle_solver="exact"):
    from xrsfm_amd import capi
    opt_kw = opt_kw or {}
    pr = H.to_oracle(arr)
    o = bo.Options(linear_solver=oracle_solver, **{k: v for k, v in opt_kw.items() if hasattr(bo.Options, k) and k != "linear_solver"})
    s_ref = bo.solve(pr, o)
    prod = H.to_product(arr)
    copt = capi.default_options(**{k: v for k, v in opt_kw.items()})
    s = capi.solve(prod, copt)
    return pr, s_ref, prod, s


@pytest.mark.parametrize("variant", ["kitti", "models", "structure_only", "lba", "kgba"])
def test_full_solve_parity(lib, variant):
    arr = H.make(10, 400, 4, seed=103)
    kw = {}
    if variant == "models":
        arr = H.with_models(arr, seed=4)
    if variant == "structure_only":          # GBA(map, true, true): every pose constant (ba_solver.cc:616-621)
        arr["cam_const"][:] = 3
    if variant == "lba":                     # LBA options + points not seen by the new frame constant (:380-382,:587-589)
        kw = dict(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5)
        seen = np.zeros(arr["points"].shape[0], bool); seen[arr["obs_pt"][arr["obs_cam"] == 9]] = True
        arr["point_const"][:] = (~seen).astype(np.uint8)
    if variant == "kgba":                    # KGBA options (:667-670)
        kw = dict(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5, initial_radius=1e6)
    pr, s_ref, prod, s = _solve_both(arr, kw)
    n_res = 2 * arr["obs_cam"].shape[0]
    rmse_ref = math.sqrt(s_ref.final_cost / n_res); rmse = math.sqrt(s.final_cost / n_res)
    assert s.n_successful == s_ref.n_successful and s.n_unsuccessful == s_ref.n_unsuccessful
    assert abs(math.sqrt(s.initial_cost / n_res) - math.sqrt(s_ref.initial_cost / n_res)) < 1e-9
    assert abs(rmse - rmse_ref) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5
    assert np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
    assert s.num_residuals == s_ref.num_residuals and s.num_effective_params == s_ref.num_effective_params
    # the plain RMSE of the returned state, evaluated by the oracle
    r_plain_prod = bo.rmse_pair(H.to_oracle(prod.as_dict()))[1]
    r_plain_ref = bo.rmse_pair(pr)[1]
    assert abs(r_plain_prod - r_plain_ref) < 1e-6


def test_context_reset_and_rerun(lib):
    from xrsfm_amd import capi
    arr = H.make(8, 200, 4, seed=104)
    ctx = capi.Context(H.to_product(arr))
    s1 = ctx.run()
    q1, t1, P1 = ctx.download()
    ctx.reset()
    s2 = ctx.run()
    q2, t2, P2 = ctx.download()
    assert s1.final_cost == s2.final_cost and s1.n_successful == s2.n_successful
    assert np.array_equal(q1, q2) and np.array_equal(t1, t2) and np.array_equal(P1, P2)   # bit-reproducible
    ctx.close()


def test_edge_cases(lib):
    from xrsfm_amd import capi
    arr = H.make(6, 50, 3, seed=105)
    # a point without observations and a camera without observations must be left untouched
    arr["points"] = np.concatenate([arr["points"], [[1.0, 2.0, 3.0]]]); arr["point_const"] = np.append(arr["point_const"], 0).astype(np.uint8)
    arr["cam_q"] = np.concatenate([arr["cam_q"], [[0, 0, 0, 1.0]]]); arr["cam_t"] = np.concatenate([arr["cam_t"], [[4.0, 5.0, 6.0]]])
    arr["cam_const"] = np.append(arr["cam_const"], 0).astype(np.uint8); arr["cam_intr"] = np.append(arr["cam_intr"], 0).astype(np.int32)
    pr, s_ref, prod, s = _solve_both(arr)
    assert np.array_equal(prod.points[-1], [1.0, 2.0, 3.0]) and np.array_equal(prod.cam_t[-1], [4.0, 5.0, 6.0])
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost
    # empty problem
    empty = capi.ProblemArrays(cam_q=np.zeros((0, 4)), cam_t=np.zeros((0, 3)), cam_intr=np.zeros(0, np.int32),
                               intr_model=np.zeros(0, np.int32), intr_params=np.zeros((0, 8)), points=np.zeros((0, 3)),
                               obs_cam=np.zeros(0, np.int32), obs_pt=np.zeros(0, np.int32), obs_uv=np.zeros((0, 2)))
    s0 = capi.solve(empty)
    assert s0.final_cost == 0.0 and s0.termination == 0
    # bad index -> EINVAL, not a crash
    bad = H.to_product(arr); bad.obs_cam[0] = 99
    with pytest.raises(RuntimeError):
        capi.solve(bad)


@pytest.mark.parametrize("n_cams,n_pts,k_obs,mode", [(8, 300, 4, "sequential"), (30, 500, 5, "sequential"),
                                                     (80, 12, 70, "unordered"), (40, 800, 6, "unordered"),
                                                     # band / ring patterns -> nested-dissection order + level schedule
                                                     (60, 700, 2, "sequential"), (130, 1500, 4, "sequential"),
                                                     (257, 2500, 3, "sequential"),
                                                     # ragged tracks (missed detections): non-dense Gram tiles
                                                     (40, 1200, 8, "ragged"), (90, 2500, 12, "ragged"),
                                                     # regular tiles of 14 / 19 / 21-camera tracks (not Gram tiles: more than 10 cameras):
                                                     # the per-camera sum over the tracks takes 4-5 rounds of 64 values, and
                                                     # the last round's source lanes (shuffle) lie beyond its active lanes
                                                     (32, 300, 14, "sequential"), (19, 200, 19, "unordered"), (21, 150, 21, "unordered")])
def test_cholesky_reduced_system(lib, n_cams, n_pts, k_obs, mode):
    """Explicit reduced camera matrix S and the tile Cholesky solve against the oracle's dense Schur complement."""
    import scipy.linalg as sla
    from xrsfm_amd import capi
    if mode == "ragged":
        arr = H.make(n_cams, n_pts, k_obs, seed=106, min_tri_angle_deg=0.5, dropout=0.35)
    else:
        arr = H.make(n_cams, n_pts, k_obs, seed=106, min_tri_angle_deg=0.5, mode=mode)
    pr = H.to_oracle(arr)
    ctx = capi.Context(H.to_product(arr))
    ctx.debug_linearize(5.99, True)
    _, _, _, _, lin = _lin_oracle(pr, True)
    radius = 3e3
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    # dense S via the implicit product applied to the identity (oracle arithmetic)
    Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
    ci, pi = pr.obs_cam, pr.obs_pt
    n = 6 * n_cams
    S_ref = np.zeros((n, n))
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
    for c in range(n_cams):
        S_ref[6 * c:6 * c + 6, 6 * c:6 * c + 6] = lin.Hcc[c] + np.diag(Dc2[c])
    order = np.argsort(pi, kind="stable")
    ptr = np.searchsorted(pi[order], np.arange(n_pts + 1))
    for j in range(n_pts):
        ids = order[ptr[j]:ptr[j + 1]]
        for a in ids:
            for b2 in ids:
                S_ref[6 * ci[a]:6 * ci[a] + 6, 6 * ci[b2]:6 * ci[b2] + 6] -= WH[a] @ lin.W[b2].T
    b_ref = lin.gc - bo._scatter_add(n_cams, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    y_ref = sla.cho_solve(sla.cho_factor(S_ref, lower=True), b_ref.reshape(-1)).reshape(n_cams, 6)
    y, S = ctx.debug_cholesky_solve(radius, want_S=True)
    assert H.rel_err(S, S_ref) < 1e-11
    assert H.rel_err(y, y_ref) < 1e-8       # conditioned by S; the full-solve tests bound the end effect
    ctx.close()


@pytest.mark.gpu
def test_debug_backsub_needs_a_solved_step(lib):
    """xrsfm_ba_debug_backsub reads the camera part of a step and the radius it was assembled with: without a preceding
    xrsfm_ba_debug_cholesky_solve of the same linearisation it must refuse (XRSFM_BA_ESTATE, -5) instead of launching the
    kernel on uninitialised point factors / a zero radius."""
    from xrsfm_amd import capi
    arr = H.make(6, 200, 3, seed=41)
    ctx = capi.Context(H.to_product(arr))
    with pytest.raises(RuntimeError, match="-5"):
        ctx.debug_backsub()                       # not even linearised
    ctx.debug_linearize(5.99, True)
    with pytest.raises(RuntimeError, match="-5"):
        ctx.debug_backsub()                       # linearised, no step solved
    ctx.debug_cholesky_solve(1e4)
    out = ctx.debug_backsub()
    assert np.isfinite(out["cand_points"]).all() and np.isfinite(out["part_model"]).all()
    ctx.debug_linearize(5.99, True)               # a new linearisation invalidates the step
    with pytest.raises(RuntimeError, match="-5"):
        ctx.debug_backsub()
    ctx.close()


@pytest.mark.gpu
def test_regular_tiles_of_every_track_length_match_oracle(lib):
    """Every track sees every camera: regular tiles with L = n_cams cameras per track, 2 <= L <= 32 (Gram tiles up to 10 cameras,
    the per-pair path beyond; the per-camera sums over a tile's tracks take 1-7 rounds of 64 values).  Three LM iterations on
    the exact path against the oracle: same decisions, same cost, same cameras."""
    from xrsfm_amd import capi
    for L in list(range(2, 25)) + [28, 32]:
        arr = H.make(L, 120, L, seed=700 + L, mode="unordered", min_tri_angle_deg=0.5)
        st = capi.debug_pack(H.to_product(arr))
        assert st["regular_tiles"] >= st["tiles"] - 1, L
        pr, s_ref, prod, s = _solve_both(dict(arr), dict(max_iterations=3, linear_solver=1))
        n_res = 2 * arr["obs_cam"].shape[0]
        assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful), L
        assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6, L
        assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5, L


@pytest.mark.parametrize("solver", [0, 1])
def test_solver_variants_agree(lib, solver):
    """PCG (tol 1e-12) and Cholesky follow the same LM trajectory as the oracle's exact solve."""
    from xrsfm_amd import capi
    arr = H.make(20, 1500, 4, seed=107)
    pr, s_ref, prod, s = _solve_both(arr, dict(linear_solver=solver))
    assert s.linear_solver_used == solver
    n_res = 2 * arr["obs_cam"].shape[0]
    assert s.n_successful == s_ref.n_successful and s.n_unsuccessful == s_ref.n_unsuccessful
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert max(np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max()) < 1e-5


def test_profile_entries(lib):
    from xrsfm_amd import capi
    arr = H.make(12, 600, 4, seed=108)
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options(profile=1))
    prof = ctx.profile()
    # two linearisations at iteration 0; then every attempted step is followed either by a linearisation at the candidate
    # (its cost is the candidate cost) or, after a rejected step, by a cost-only pass plus a linearisation if it is accepted
    n_lin, n_cost = prof["k_linearize"][1], prof["k_cost"][1]
    assert 2 + s.n_successful <= n_lin <= 2 + s.lm_steps_attempted
    assert n_cost <= s.n_unsuccessful and (n_lin - 2) + n_cost >= s.lm_steps_attempted
    assert prof["k_potrf"][1] > 0 and s.dom_kernel_ms > 0
    ctx.close()


def _golden():
    import glob, os
    other = {"track_filter.npz", "tag_refine.npz", "pose_graph.npz", "wide_bal9.npz"}       # fixtures of the "next" rows: their own tests
    return sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                  if os.path.basename(p) not in other and os.path.basename(p) != "lba_selection.npz" and not os.path.basename(p).startswith("ceres_"))


@pytest.mark.parametrize("path", _golden(), ids=[p.split("/")[-1][:-4] for p in _golden()])
@pytest.mark.parametrize("solver", [0, 1])
def test_hip_path_reproduces_golden(lib, path, solver):
    """Committed golden fixtures (tests/golden/make_golden.py): initial blocks, LM step counts, final RMSE and cameras."""
    from xrsfm_amd import capi
    z = np.load(path)
    arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    mi, ft, pt, rad = z["opt"]
    ctx = capi.Context(H.to_product(arr))
    out = ctx.debug_linearize(5.99, False)
    assert abs(out["cost"] - float(z["init_cost"])) <= 1e-12 * float(z["init_cost"])
    assert H.rel_err(out["r"], z["init_r"]) < 1e-12 and H.rel_err(out["Jc"], z["init_Jc"]) < 1e-11 and H.rel_err(out["Jp"], z["init_Jp"]) < 1e-11
    ctx.close()
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=int(mi), function_tolerance=float(ft), parameter_tolerance=float(pt),
                                              initial_radius=float(rad), linear_solver=solver))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(math.sqrt(s.final_cost / n_res) - float(z["rmse_ref_style"])) < 1e-6
    assert np.abs(prod.cam_q - z["out_cam_q"]).max() < 1e-5 and np.abs(prod.cam_t - z["out_cam_t"]).max() < 1e-5


def test_full_size_properties(lib):
    """BASELINE.json config 2 (100 cams / 50k points / 200k obs): size-independent properties + C restatement."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, synth
    d = synth.make_problem(**synth.CONFIGS["S"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options())
    q, t, P = ctx.download()
    assert s.termination == 0 and s.final_cost < s.initial_cost
    # idempotence: re-solving from the optimum takes (almost) no steps and does not move the cameras
    prod2 = H.to_product(dict(arr, cam_q=q, cam_t=t, points=P))
    s2 = capi.solve(prod2)
    assert s2.n_successful <= 1 and np.abs(prod2.cam_q - q).max() < 1e-4
    # the returned state really has the reported cost (evaluated by the oracle)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    if ba_cpu.available():
        prob = {k: np.array(v, copy=True) for k, v in arr.items()}
        sc = ba_cpu.solve(prob, threads=8)
        n_res = 2 * arr["obs_cam"].shape[0]
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
        assert max(np.abs(prob["cam_q"] - q).max(), np.abs(prob["cam_t"] - t).max()) < 1e-5
    ctx.close()


@pytest.mark.parametrize("n_cams,n_pts,k_obs", [(40, 900, 3), (130, 2000, 4)])
def test_sharded_reduced_matrix_sums_to_full(lib, n_cams, n_pts, k_obs):
    """Multi-GPU Cholesky path, emulated on one GPU: two shards (bench.shard_problem) with the union block pattern
    that xrsfm_ba_run obtains by all-reduce; the shards' reduced camera matrices and right-hand sides must add up to
    the single-rank ones (that sum is what the RCCL all-reduce of the block values produces)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import shard_problem, count_offdiag_blocks
    from xrsfm_amd import capi
    arr = H.make(n_cams, n_pts, k_obs, seed=140)
    full = capi.Context(H.to_product(arr))
    full.debug_linearize(5.99, False)
    _, S_full = full.debug_cholesky_solve(2e3, want_S=True)
    # union pattern from the full problem
    order = np.lexsort((arr["obs_cam"], arr["obs_pt"]))
    cam = arr["obs_cam"][order]; pt = arr["obs_pt"][order]
    pairs = set()
    for d in range(1, k_obs):
        same = pt[d:] == pt[:-d]
        pairs |= set(zip(cam[d:][same].tolist(), cam[:-d][same].tolist()))
    pattern = np.array(sorted(pairs), np.int32)
    assert pattern.shape[0] == count_offdiag_blocks(arr)
    S_sum = np.zeros_like(S_full)
    for r in range(2):
        ctx = capi.Context(H.to_product(shard_problem(arr, r, 2)))
        ctx.debug_set_block_pattern(pattern)
        ctx.debug_linearize(5.99, False)
        _, S = ctx.debug_cholesky_solve(2e3, want_S=True)
        S_sum += S
        ctx.close()
    assert H.rel_err(S_sum, S_full) < 1e-12
    full.close()


def test_rccl_plumbing_single_rank(lib, monkeypatch):
    """A real 1-rank RCCL communicator (XRSFM_BA_FORCE_COMM=1): unique id, ncclCommInitRank with the id passed by value,
    every all-reduce of the solve issued on the library's stream.  Results must equal the communicator-free run bit for bit."""
    from xrsfm_amd import capi
    arr = H.make(20, 1200, 4, seed=141)
    a = capi.Context(H.to_product(arr))
    sa = a.run(); qa, ta, Pa = a.download(); a.close()
    monkeypatch.setenv("XRSFM_BA_FORCE_COMM", "1")
    uid = capi.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    for solver in (1, 0):
        b = capi.Context(H.to_product(arr))
        b.comm_init(1, 0, uid if solver == 1 else capi.comm_unique_id())
        sb = b.run(capi.default_options(linear_solver=solver)); qb, tb, Pb = b.download(); b.close()
        assert sb.n_successful == sa.n_successful
        if solver == 1:
            assert sb.final_cost == sa.final_cost and np.array_equal(qa, qb) and np.array_equal(ta, tb) and np.array_equal(Pa, Pb)
        else:
            assert abs(sb.final_cost - sa.final_cost) <= 1e-9 * sa.final_cost


@pytest.mark.parametrize("variant", ["kitti", "models"])
def test_filter_tracks_matches_reference_rules(lib, variant):
    """FilterPoints3d on the GPU against the oracle restatement (bit-exact masks and counters)."""
    from xrsfm_amd import capi
    arr = H.make(12, 900, 4, seed=150, outlier_frac=0.08, min_tri_angle_deg=0.2)
    if variant == "models":
        arr = H.with_models(arr, seed=8)
    arr["points"][::50] += np.array([0.0, 0.0, -70.0])        # behind the cameras: depth test
    arr["points"][7::60] *= 40.0                              # far away: small triangulation angle / depth > 1e3
    max_re, min_angle = 4.0, math.radians(1.5)                # rec_1dsfm.cc:90-91
    ref = bo.filter_tracks(H.to_oracle(arr), max_re, min_angle)
    got = capi.filter_tracks(H.to_product(arr), max_re, min_angle)
    assert np.array_equal(got["obs_delete"], ref["obs_delete"]) and got["obs_delete"].sum() > 20
    assert np.array_equal(got["track_outlier"], ref["track_outlier"])
    assert set(np.unique(ref["track_outlier"])) == {0, 1, 2}
    assert np.array_equal(got["num_filtered"], ref["num_filtered"])
    keep = ref["track_outlier"] != 1
    assert np.abs(got["track_error"][keep] - ref["track_error"][keep]).max() < 1e-9
    assert np.abs(got["track_angle"][keep] - ref["track_angle"][keep]).max() < 1e-12


def test_filter_tracks_reproduces_golden(lib):
    """tests/golden/track_filter.npz (make_golden_extra.py): masks and counters bit-exact, error / angle to 1e-9 / 1e-12."""
    import os
    from xrsfm_amd import capi
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "track_filter.npz"))
    arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    got = capi.filter_tracks(H.to_product(arr), float(z["max_re"]), float(z["min_angle"]))
    for k in ("obs_delete", "track_outlier", "num_filtered"):
        assert np.array_equal(got[k], z["out_" + k]), k
    keep = z["out_track_outlier"] != 1
    assert np.abs(got["track_error"][keep] - z["out_track_error"][keep]).max() < 1e-9
    assert np.abs(got["track_angle"][keep] - z["out_track_angle"][keep]).max() < 1e-12


def test_headline_config_properties(lib):
    """BASELINE.json config 4 (1k cams / 500k points / 2M obs), the bench workload: termination, cost decrease, the
    reported cost is the cost of the returned state, bit-reproducibility, and RMSE parity with the C restatement."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, synth
    d = synth.make_problem(**synth.CONFIGS["L"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options())
    q, t, P = ctx.download()
    n_res = 2 * arr["obs_cam"].shape[0]
    assert s.termination == 0 and s.final_cost < 0.05 * s.initial_cost and s.linear_solver_used == 1
    ctx.reset()
    s2 = ctx.run(capi.default_options())
    q2, t2, P2 = ctx.download()
    assert s2.final_cost == s.final_cost and np.array_equal(q, q2) and np.array_equal(P, P2)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    if ba_cpu.available():
        prob = {k: np.array(v, copy=True) for k, v in arr.items()}
        sc = ba_cpu.solve(prob, threads=8)
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
    ctx.close()


def test_pose_only_refinement(lib):
    """the pose refinement of RegisterImage (/root/reference/src/geometry/pnp.cc:38-71) is the same
    engine with one camera, every point constant and Ceres' default options (10 iterations, ftol 1e-6, ptol 1e-8)."""
    from xrsfm_amd import capi
    full = H.make(8, 400, 4, seed=170)
    keep = full["obs_cam"] == 5
    pts, inv = np.unique(full["obs_pt"][keep], return_inverse=True)
    rng = np.random.default_rng(3)
    arr = dict(cam_q=bo.quat_plus(full["cam_q"][5:6], rng.normal(0, 0.02, (1, 3))), cam_t=full["cam_t"][5:6] + rng.normal(0, 0.1, (1, 3)),
               cam_const=np.zeros(1, np.uint8), cam_intr=np.zeros(1, np.int32), intr_model=full["intr_model"], intr_params=full["intr_params"],
               points=full["points"][pts], point_const=np.ones(len(pts), np.uint8),
               obs_cam=np.zeros(int(keep.sum()), np.int32), obs_pt=inv.astype(np.int32), obs_uv=full["obs_uv"][keep])
    kw = dict(max_iterations=10, function_tolerance=1e-6, parameter_tolerance=1e-8)
    pr, s_ref, prod, s = _solve_both(arr, kw)
    assert s.num_effective_params == 6 and s_ref.num_effective_params == 6
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-9 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-8
    assert np.array_equal(prod.points, arr["points"])           # constant points are returned untouched


@pytest.mark.parametrize("solver", [0, 1])
def test_ragged_tracks(lib, solver):
    """Ragged input: track lengths 1..6 after dropping random observations (single-observation points enter with a rank-2
    point block that only the LM damping regularises), mixed regular/irregular tiles, shuffled order."""
    arr = H.make(14, 700, 6, seed=180, mode="unordered", min_tri_angle_deg=0.5)
    rng = np.random.default_rng(9)
    n = arr["obs_cam"].shape[0]
    keep = rng.random(n) < 0.55
    first = np.zeros(n, bool); first[np.unique(arr["obs_pt"], return_index=True)[1]] = True     # >= 1 observation per point
    keep |= first
    perm = rng.permutation(int(keep.sum()))
    for k in ("obs_cam", "obs_pt", "obs_uv"):
        arr[k] = np.ascontiguousarray(arr[k][keep][perm])
    lens = np.bincount(arr["obs_pt"], minlength=700)
    assert lens.min() == 1 and lens.max() == 6 and (lens == 1).sum() > 5
    pr, s_ref, prod, s = _solve_both(arr, dict(linear_solver=solver, max_iterations=15))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert max(np.abs(prod.cam_q - pr.cam_q).max(), np.abs(prod.cam_t - pr.cam_t).max()) < 1e-5


def test_band_with_loop_closures(lib):
    """A sequential problem with a few long-range camera pairs (loop closures): the band ordering must survive them (level
    schedule still shallow) and the solve must still match the oracle."""
    from xrsfm_amd import capi
    arr = H.make(160, 3000, 4, seed=200)
    rng = np.random.default_rng(4)
    # 6 extra tracks, each seen by two far-apart camera pairs (geometry does not matter for the structure: observations
    # that end up behind a camera take the clamp branch of the cost functor)
    n_p = arr["points"].shape[0]
    extra_cam, extra_pt, extra_uv, extra_P = [], [], [], []
    for e in range(6):
        a, b = int(rng.integers(0, 60)), int(rng.integers(90, 150))
        for cidx in (a, a + 1, b, b + 1):
            extra_cam.append(cidx); extra_pt.append(n_p + e); extra_uv.append(rng.uniform([100, 50], [1100, 300]))
        extra_P.append(arr["points"][int(rng.integers(0, n_p))] + rng.normal(0, 0.5, 3))
    arr["points"] = np.concatenate([arr["points"], np.array(extra_P)])
    arr["point_const"] = np.zeros(arr["points"].shape[0], np.uint8)
    arr["obs_cam"] = np.concatenate([arr["obs_cam"], np.array(extra_cam, np.int32)])
    arr["obs_pt"] = np.concatenate([arr["obs_pt"], np.array(extra_pt, np.int32)])
    arr["obs_uv"] = np.concatenate([arr["obs_uv"], np.array(extra_uv)])
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options(profile=1, max_iterations=8))
    prof = ctx.profile()
    levels = prof["k_potrf"][1] / (s.lm_steps_attempted)
    assert levels <= 12, levels                      # 96 tiles would be ~96 sequential panels with the natural order
    ctx.close()
    pr, s_ref, prod, s2 = _solve_both(arr, dict(max_iterations=8))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s2.n_successful, s2.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s2.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
def test_refine_pose_matches_oracle(lib, model):
    """xrsfm_ba_refine_pose (RegisterImage's refinement, pnp.cc:38-71) against the oracle on the same one-camera problem;
    correspondences masked out by the inlier mask must not enter (pnp.cc:43-45)."""
    from xrsfm_amd import capi
    arr = H.make_pose_problem(180, seed=300 + model, model=model)
    rng = np.random.default_rng(model)
    mask = (rng.random(180) < 0.85).astype(np.uint8)
    keep = np.nonzero(mask)[0]
    sub = dict(arr)
    sub["points"] = arr["points"][keep]; sub["point_const"] = arr["point_const"][keep]
    sub["obs_cam"] = arr["obs_cam"][keep]; sub["obs_pt"] = np.arange(keep.size, dtype=np.int32); sub["obs_uv"] = arr["obs_uv"][keep]
    pr = H.to_oracle(sub)
    s_ref = bo.solve(pr, bo.Options(max_iterations=10, function_tolerance=1e-6, parameter_tolerance=1e-8))
    q, t, s = capi.refine_pose(model, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0], inlier_mask=mask)
    assert s.num_residuals == 2 * keep.size and s.num_effective_params == 6
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(s.initial_cost - s_ref.initial_cost) <= 1e-9 * s_ref.initial_cost
    n_res = 2 * keep.size
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(q - pr.cam_q[0]).max() < 1e-5 and np.abs(t - pr.cam_t[0]).max() < 1e-5
    # the refinement must have moved the pose towards a lower cost
    assert s.final_cost < s.initial_cost


@pytest.mark.gpu
@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
def test_refine_pose_persistent_kernel_equals_engine(lib, model):
    """The one-workgroup LM kernel (ba_refine.h, the default) against the same problem through the general engine
    (XRSFM_BA_REFINE_ENGINE=1): same step counts and exits, costs to 1e-10, pose to 1e-9 (summation orders differ)."""
    import os
    from xrsfm_amd import capi
    arr = H.make_pose_problem(500, seed=700 + model, model=model, outlier_frac=0.15)
    args = (model, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0])
    q1, t1, s1 = capi.refine_pose(*args)
    os.environ["XRSFM_BA_REFINE_ENGINE"] = "1"
    try:
        q2, t2, s2 = capi.refine_pose(*args)
    finally:
        del os.environ["XRSFM_BA_REFINE_ENGINE"]
    assert (s1.n_successful, s1.n_unsuccessful, s1.termination, s1.termination_reason, s1.lm_steps_attempted) == \
           (s2.n_successful, s2.n_unsuccessful, s2.termination, s2.termination_reason, s2.lm_steps_attempted)
    assert abs(s1.initial_cost - s2.initial_cost) <= 1e-12 * s2.initial_cost and abs(s1.final_cost - s2.final_cost) <= 1e-10 * s2.final_cost
    assert np.abs(q1 - q2).max() < 1e-9 and np.abs(t1 - t2).max() < 1e-9
    assert (s1.num_residuals, s1.num_effective_params, s1.linear_solver_used) == (s2.num_residuals, s2.num_effective_params, s2.linear_solver_used)


@pytest.mark.gpu
def test_refine_poses_batch_equals_single_calls(lib):
    """xrsfm_ba_refine_poses: several frames in one launch (one workgroup each) = the single-frame calls, bit for bit; frames of
    different sizes and camera models, one with an inlier mask, one without correspondences."""
    from xrsfm_amd import capi
    probs = [H.make_pose_problem(n, seed=800 + i, model=m) for i, (n, m) in enumerate([(300, 2), (40, 4), (1500, 0), (7, 3), (200, 1)])]
    masks = [None, None, (np.arange(1500) % 5 != 0).astype(np.uint8), None, np.zeros(200, np.uint8)]
    frames = [(p["points"], p["obs_uv"], mk) for p, mk in zip(probs, masks)]
    models = [int(p["intr_model"][0]) for p in probs]
    intr = [p["intr_params"][0] for p in probs]
    q0 = np.array([p["cam_q"][0] for p in probs]); t0 = np.array([p["cam_t"][0] for p in probs])
    qb, tb, sb = capi.refine_poses(models, intr, frames, q0, t0)
    for f, p in enumerate(probs):
        q1, t1, s1 = capi.refine_pose(models[f], intr[f], p["points"], p["obs_uv"], q0[f], t0[f], inlier_mask=masks[f])
        assert np.array_equal(qb[f], q1) and np.array_equal(tb[f], t1)
        assert (sb[f].initial_cost, sb[f].final_cost, sb[f].n_successful, sb[f].n_unsuccessful, sb[f].termination_reason, sb[f].num_residuals) == \
               (s1.initial_cost, s1.final_cost, s1.n_successful, s1.n_unsuccessful, s1.termination_reason, s1.num_residuals)
    assert sb[4].num_residuals == 0 and np.array_equal(qb[4], q0[4])
    assert sb[0].final_cost < sb[0].initial_cost


@pytest.mark.gpu
def test_refine_pose_edge_cases(lib):
    from xrsfm_amd import capi
    arr = H.make_pose_problem(40, seed=9, model=2)
    # nothing selected: no residuals, pose untouched, CONVERGENCE by the gradient test like an empty Ceres problem
    q, t, s = capi.refine_pose(2, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0],
                               inlier_mask=np.zeros(40, np.uint8))
    assert s.num_residuals == 0 and np.array_equal(q, arr["cam_q"][0]) and np.array_equal(t, arr["cam_t"][0])
    with pytest.raises(Exception):
        capi.refine_pose(7, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [1, 0], ids=["cholesky", "pcg"])
def test_gradient_tolerance_exit_matches_oracle(lib, solver):
    """Exit by the gradient tolerance after an accepted step.  The product learns the gradient norm of a new linearisation
    one solve late (it travels with the next step's scalars); that extra solve must be neither counted nor applied."""
    arr = H.make(8, 300, 4, seed=410, noise=0.0, outlier_frac=0.0)
    kw = dict(gradient_tolerance=1e-3, function_tolerance=1e-30, parameter_tolerance=1e-30, max_iterations=30)
    pr, s_ref, prod, s = _solve_both(dict(arr), dict(kw, linear_solver=solver))
    assert s_ref.termination.startswith("CONVERGENCE: gradient"), s_ref.termination
    assert s.termination_reason == 1
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert s.lm_steps_attempted == s.n_successful + s.n_unsuccessful
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * max(s_ref.final_cost, 1e-12) + 1e-14
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
    # max_iterations hit while a linearisation is still pending: same counts as the oracle as well
    kw2 = dict(max_iterations=2)
    pr, s_ref, prod, s = _solve_both(dict(arr), dict(kw2, linear_solver=solver))
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful) and s.termination_reason == 5
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost


@pytest.mark.gpu
def test_auto_solver_beyond_the_dense_limit(lib):
    """More than 12288 camera unknowns: a sequential (band) problem still gets the exact tile Cholesky through AUTO (shallow
    elimination tree, S in tile storage) and matches the C restatement; random visibility at that size falls back to PCG, and an
    explicit CHOLESKY request is refused with ETOOBIG instead of running a 200-panel dense factorisation."""
    from oracle import ba_cpu
    from xrsfm_amd import capi
    arr = H.make(2100, 9000, 4, seed=420)
    assert 6 * 2100 > 12288
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 1 and plan["level_schedule"] == 1 and plan["levels"] <= 8
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=10))
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY and s.final_cost < s.initial_cost
    if ba_cpu.available():
        prob = {k: np.array(v, copy=True) for k, v in arr.items()}
        sc = ba_cpu.solve(prob, max_iterations=10, threads=8)
        n_res = 2 * arr["obs_cam"].shape[0]
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
    arr_u = H.make(2100, 6000, 4, seed=421, mode="unordered")
    prod_u = H.to_product(arr_u)
    s_u = capi.solve(prod_u, capi.default_options(max_iterations=3, pcg_tolerance=1e-6))
    assert s_u.linear_solver_used == capi.SOLVER_PCG and s_u.final_cost < s_u.initial_cost
    with pytest.raises(Exception, match="ETOOBIG"):
        capi.solve(H.to_product(arr_u), capi.default_options(max_iterations=1, linear_solver=capi.SOLVER_CHOLESKY))


@pytest.mark.gpu
@pytest.mark.parametrize("n_cams,panel_cols", [(150, 2), (150, 4), (95, 2), (230, 6)])
def test_dense_pattern_panel_schedule_matches_oracle(lib, monkeypatch, n_cams, panel_cols):
    """Unordered visibility -> a full tile pattern with one column per elimination-tree level: the panel schedule (left-looking
    updates in chunks, 128x128 macro tiles over `panel_cols` columns at a time, fused pivot / triangular-solve kernel, push-form
    backward substitution).  The production rule switches macro tiles on from 96 tile columns; the developer switches force
    them here on 10-23 columns (odd counts: the last row pair is half empty) and the result must equal both the oracle and
    the same solve without macro tiles."""
    from xrsfm_amd import capi
    arr = H.make(n_cams, 40 * n_cams, 5, seed=460 + n_cams, mode="unordered")
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["level_schedule"] == 0 and plan["tiles"] == (n_cams + 9) // 10
    kw = dict(max_iterations=8, linear_solver=1)
    monkeypatch.setenv("XRSFM_BA_PANEL_MACRO", "0")
    prod0 = H.to_product(arr)
    s0 = capi.solve(prod0, capi.default_options(**kw))
    monkeypatch.setenv("XRSFM_BA_PANEL_MACRO", "1")
    monkeypatch.setenv("XRSFM_BA_PANEL_COLS", str(panel_cols))
    pr, s_ref, prod, s = _solve_both(dict(arr), kw)
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful) == (s0.n_successful, s0.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
    assert np.abs(prod.cam_q - prod0.cam_q).max() < 1e-9 and np.abs(prod.cam_t - prod0.cam_t).max() < 1e-8
    assert abs(s.final_cost - s0.final_cost) <= 1e-10 * s0.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [1, 0], ids=["cholesky", "pcg"])
def test_ragged_tracks_match_oracle(lib, solver):
    """Windows of 8 frames with 35 % missed detections: many distinct camera tuples, tiles whose tracks see different camera
    subsets (non-dense Gram tiles with zero-filled cells, per-camera sums over scattered lanes).  Reduced matrix, solution of the
    reduced system and the full solve against the oracle."""
    from xrsfm_amd import capi
    arr = H.make(40, 1200, 8, seed=430, dropout=0.35)
    lens = np.bincount(arr["obs_pt"])
    assert lens.min() >= 2 and len(set(lens.tolist())) >= 5
    if solver == 1:
        pr = H.to_oracle(arr)
        ctx = capi.Context(H.to_product(arr))
        ctx.debug_linearize(use_scaling=True)
        y, S = ctx.debug_cholesky_solve(1e4, want_S=True)
        cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
        ctx.close()
        assert np.isfinite(S).all() and np.abs(S - S.T).max() <= 1e-9 * np.abs(S).max()
    pr, s_ref, prod, s = _solve_both(dict(arr), dict(max_iterations=12, linear_solver=solver))
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5


@pytest.mark.gpu
def test_track_observed_twice_by_one_camera(lib):
    """A track with two observations in the same frame (the map format allows it; Ceres adds both residual blocks,
    /root/reference/src/optimization/ba_solver.cc:336-349): AUTO must not abort — it takes the implicit-Schur path, which treats
    every observation on its own — and matches the oracle; an explicit CHOLESKY request is refused with EINVAL."""
    from xrsfm_amd import capi
    arr = H.make(10, 300, 4, seed=191)
    rng = np.random.default_rng(5)
    dup = rng.choice(arr["obs_cam"].shape[0], 12, replace=False)
    arr["obs_cam"] = np.concatenate([arr["obs_cam"], arr["obs_cam"][dup]]).astype(np.int32)
    arr["obs_pt"] = np.concatenate([arr["obs_pt"], arr["obs_pt"][dup]]).astype(np.int32)
    arr["obs_uv"] = np.concatenate([arr["obs_uv"], arr["obs_uv"][dup] + rng.normal(0, 0.7, (12, 2))])
    pr, s_ref, prod, s = _solve_both(dict(arr), dict(max_iterations=12))
    assert s.linear_solver_used == capi.SOLVER_PCG
    n_res = 2 * arr["obs_cam"].shape[0]
    assert s.num_residuals == n_res
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s_ref.final_cost / n_res)) < 1e-6
    assert np.abs(prod.cam_q - pr.cam_q).max() < 1e-5 and np.abs(prod.cam_t - pr.cam_t).max() < 1e-5
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.solve(H.to_product(arr), capi.default_options(max_iterations=2, linear_solver=capi.SOLVER_CHOLESKY))
    # the post-BA filter on the same data: every observation gets its own mask entry
    out = capi.filter_tracks(H.to_product(arr), 4.0, math.radians(1.5))
    assert out["obs_delete"].shape[0] == arr["obs_cam"].shape[0]


@pytest.mark.gpu
def test_failure_paths_of_the_side_entry_points(lib):
    """xrsfm_ba_filter_tracks / xrsfm_ba_refine_pose(s) with broken input on the GPU box: a negative code, no crash, outputs
    untouched (track_processor.cc:321-349 and pnp.cc:38-71 have void/plain-int call sites: the adapters print and carry on)."""
    import ctypes as C
    from xrsfm_amd import capi
    L = capi.load()
    arr = H.make(6, 80, 3, seed=192)
    bad = H.to_product(arr); bad.obs_pt[3] = 10_000
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.filter_tracks(bad, 4.0, 0.02)
    bad = H.to_product(arr); bad.cam_intr[2] = 7
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.filter_tracks(bad, 4.0, 0.02)
    # NULL output arrays / NULL problem arrays straight through the C-ABI
    prob = H.to_product(arr)
    cs = prob.c_struct()
    assert L.xrsfm_ba_filter_tracks(C.byref(cs), 4.0, 0.02, None, None, None, None, None) == -1
    cs2 = prob.c_struct(); cs2.points = None
    od = np.zeros(prob.n_obs, np.uint8); to = np.zeros(prob.n_points, np.uint8)
    assert L.xrsfm_ba_filter_tracks(C.byref(cs2), 4.0, 0.02, od.ctypes.data_as(C.POINTER(C.c_uint8)), to.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    None, None, None) == -1
    assert not od.any() and not to.any()
    # pose refinement: unknown camera model, missing arrays
    pa = H.make_pose_problem(60, seed=3)
    q = pa["cam_q"][0].copy(); t = pa["cam_t"][0].copy(); q0, t0 = q.copy(), t.copy()
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.refine_pose(9, pa["intr_params"][0], pa["points"], pa["obs_uv"], q, t)
    assert np.array_equal(q, q0) and np.array_equal(t, t0)
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.refine_poses([2, 7], np.stack([pa["intr_params"][0]] * 2), [(pa["points"], pa["obs_uv"], None)] * 2, np.stack([q, q]), np.stack([t, t]))


@pytest.mark.gpu
def test_headline_config_camera_parity(lib):
    """BASELINE.json config 4: the camera-parameter criterion that test_headline_config_properties could not state as a plain
    1e-5 bound (round-1 verdict).  Measured (tools/parity_spectrum.py): on this 1000-frame loop with
    4-frame tracks and only two translations fixed (ba_solver.cc:611-614), stopping at a 1e-5 relative cost change (:628)
    leaves the global shape of the loop — scale about the fixed pair, low-frequency bending — undetermined at the 1e-3 level:
    the C restatement differs from ITSELF by 9e-4 in the far-side translations when its points are merely relabelled (another
    summation order), at a Gauss-Newton energy of 1e-12 of the cost.  So the criterion is, deterministic on any box (one CPU
    thread, fixed orders):
      (1) same LM decisions, RMSE within 1e-6 px;
      (2) the objective cannot tell the results apart: 1/2 |J (x_hip - x_cpu)|^2 <= 1e-10 cost  (measured 1e-12);
      (3) what IS determined agrees: relative pose of every covisible camera pair within 1e-6 rad / 1e-4 units
          (measured 7e-9 / 1.4e-5; the north star's 1e-5 on absolute parameters holds on the well-conditioned configs,
          test_full_size_properties);
      (4) the HIP result is no further from the restatement than the restatement is from itself under relabelling (x3)."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, parity, synth
    if not ba_cpu.available():
        pytest.skip("oracle/_build/libba_cpu.so is not built")
    d = synth.make_problem(**synth.CONFIGS["L"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options())
    n_res = 2 * arr["obs_cam"].shape[0]
    c1 = {k: np.array(v, copy=True) for k, v in arr.items()}
    s1 = ba_cpu.solve(c1, threads=1)
    arr_r, perm = H.relabel_points(arr, seed=1)
    c2 = {k: np.array(v, copy=True) for k, v in arr_r.items()}
    s2 = ba_cpu.solve(c2, threads=1)
    P2 = np.empty_like(c2["points"]); P2[perm] = c2["points"]
    for sc in (s1, s2):
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
    pairs = parity.covisible_pairs(arr["obs_cam"], arr["obs_pt"])
    assert pairs.shape[0] >= 2990

    def distance(qa, ta, Pa, qb, tb, Pb):
        dv = parity.tangent_difference(qa, ta, qb, tb).reshape(-1, 6)
        ang, dtr = parity.relative_pose_difference(qa, ta, qb, tb, pairs)
        e, cost = H.gn_energy(dict(arr, cam_q=qa, cam_t=ta, points=Pa), dv, Pa - Pb)
        return float(np.abs(dv).max()), ang, dtr, e / cost

    raw, ang, dtr, erel = distance(prod.cam_q, prod.cam_t, prod.points, c1["cam_q"], c1["cam_t"], c1["points"])
    raw_cc, ang_cc, dtr_cc, erel_cc = distance(c1["cam_q"], c1["cam_t"], c1["points"], c2["cam_q"], c2["cam_t"], P2)
    print(f"hip vs cpu: raw {raw:.2e} rel-pose {ang:.2e} rad / {dtr:.2e} energy/cost {erel:.2e};  cpu vs relabelled cpu: raw {raw_cc:.2e} "
          f"rel-pose {ang_cc:.2e} / {dtr_cc:.2e} energy/cost {erel_cc:.2e}")
    assert erel <= 1e-10
    assert ang <= 1e-6 and dtr <= 1e-4
    assert raw <= 3.0 * raw_cc + 1e-5 and dtr <= 3.0 * dtr_cc + 1e-5
    # ... and the plain absolute difference keeps an explicit bound (the north star's 1e-5 is NOT met here — measured 3e-5 ..
    # 1.3e-3 depending on the CPU run's summation order; this catches a regression of the weak modes beyond that)
    assert raw <= 2e-3


@pytest.mark.gpu
def test_config4_parity_workload_meets_the_literal_north_star_bounds(lib):
    """BASELINE.json config 4 on a workload where the north star's bounds are DECIDABLE.  Config LP
    (synth.py) = the sizes and the code path of config L — 1000 frames / 500 000 points / 2 000 000 observations, band order,
    level schedule, regular 4-camera tiles — plus 24 hub frames sharing 1200 distant-landmark tracks, which tie the loop
    together: the C restatement then agrees with its own relabelled run to 1e-9 in the translations (config L: 9e-4, see
    test_headline_config_camera_parity), so the bounds can be asserted as written: same LM decisions, final RMSE within
    1e-6 px, every camera parameter within 1e-5 — against the restatement at one thread AND against its run on relabelled
    points (another summation order).  The solve also rejects steps (13 + 15 in the restatement), which the L solve never
    does: the cost-only pass and the re-linearisation after a rejection run at full size here."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, synth
    if not ba_cpu.available():
        pytest.skip("oracle/_build/libba_cpu.so is not built")
    d = synth.make_problem(**synth.CONFIGS["LP"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    assert arr["cam_q"].shape[0] == 1000 and arr["points"].shape[0] == 500_000 and arr["obs_cam"].shape[0] == 2_000_000
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 1 and plan["band"] == 3 and plan["hubs"] == 24 and plan["level_schedule"] == 1
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options())
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY and s.n_unsuccessful > 0
    n_res = 2 * arr["obs_cam"].shape[0]
    c1 = {k: np.array(v, copy=True) for k, v in arr.items()}
    s1 = ba_cpu.solve(c1, threads=1)
    arr_r, perm = H.relabel_points(arr, seed=1)
    c2 = {k: np.array(v, copy=True) for k, v in arr_r.items()}
    s2 = ba_cpu.solve(c2, threads=1)
    P2 = np.empty_like(c2["points"]); P2[perm] = c2["points"]
    worst = 0.0
    for sc, cc, Pc in ((s1, c1, c1["points"]), (s2, c2, P2)):
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
        dq = float(np.abs(prod.cam_q - cc["cam_q"]).max()); dt = float(np.abs(prod.cam_t - cc["cam_t"]).max())
        print(f"hip vs cpu: |dq| {dq:.2e} |dt| {dt:.2e} |dP| {np.abs(prod.points - Pc).max():.2e}")
        assert dq <= 1e-5 and dt <= 1e-5
        worst = max(worst, dq, dt)
    # the workload decides: the restatement against its relabelled self is orders of magnitude inside the bound
    assert np.abs(c1["cam_t"] - c2["cam_t"]).max() <= 1e-6 and worst <= 1e-5


@pytest.mark.gpu
def test_config3_shape_properties(lib):
    """BASELINE.json config 3 at size: the shape of a KITTI-00 key-frame global BA (2000 frames / 1M points / 4M observations,
    sequential visibility; synth config K — the real sequence cannot be reconstructed offline): termination, the reported cost
    is the cost of the returned state, bit-reproducibility, LM decisions and RMSE equal to the C restatement's, relative poses
    of covisible frames within 1e-6 rad / 1e-4 (test_headline_config_camera_parity explains the criterion)."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, parity, synth
    d = synth.make_problem(**synth.CONFIGS["K"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options())
    q, t, P = ctx.download()
    n_res = 2 * arr["obs_cam"].shape[0]
    assert s.termination == 0 and s.final_cost < 0.05 * s.initial_cost and s.linear_solver_used == capi.SOLVER_CHOLESKY
    ctx.reset()
    s2 = ctx.run(capi.default_options())
    q2, t2, P2 = ctx.download()
    ctx.close()
    assert s2.final_cost == s.final_cost and np.array_equal(q, q2) and np.array_equal(P, P2)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    if ba_cpu.available():
        prob = {k: np.array(v, copy=True) for k, v in arr.items()}
        sc = ba_cpu.solve(prob, threads=8)
        assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
        assert abs(math.sqrt(sc["final_cost"] / n_res) - math.sqrt(s.final_cost / n_res)) < 1e-6
        ang, dtr = parity.relative_pose_difference(q, t, prob["cam_q"], prob["cam_t"], parity.covisible_pairs(arr["obs_cam"], arr["obs_pt"]))
        print(f"config K: relative pose difference to the C restatement {ang:.2e} rad / {dtr:.2e}")
        assert ang <= 1e-6 and dtr <= 1e-4


@pytest.mark.gpu
def test_config5_shape_properties(lib):
    """BASELINE.json config 5 at the size one GPU holds in the bench: the shape of an unordered internet collection (synth
    config V: 3000 cameras around a scene, random visibility, 18 000 camera unknowns — beyond the dense limit of the exact
    factorisation, so AUTO takes the implicit-Schur PCG path).  The C restatement's dense envelope Cholesky would need ~10
    minutes per LM step at this size, so the checks are the size-independent ones: convergence, the reported cost is the
    cost of the returned state (oracle evaluation), bit-reproducibility, a stationary point of the objective (the gradient
    max-norm fell by > 1e3), and idempotence of a re-solve.  The two linear solvers are compared with each other and with
    the oracle at the sizes the exact path reaches (test_solver_variants_agree, test_auto_solver_beyond_the_dense_limit)."""
    from xrsfm_amd import capi, synth
    d = synth.make_problem(**synth.CONFIGS["V"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    ctx = capi.Context(H.to_product(arr))
    opt = capi.default_options()
    s = ctx.run(opt)
    q, t, P = ctx.download()
    assert s.linear_solver_used == capi.SOLVER_PCG and s.termination == 0 and s.final_cost < 0.05 * s.initial_cost
    assert s.pcg_iterations > 0
    ctx.reset()
    s2 = ctx.run(opt)
    q2, t2, P2 = ctx.download()
    ctx.close()
    assert s2.final_cost == s.final_cost and np.array_equal(q, q2) and np.array_equal(P, P2)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    prod2 = H.to_product(dict(arr, cam_q=q, cam_t=t, points=P))
    s3 = capi.solve(prod2)
    assert s3.n_successful <= 2 and abs(s3.final_cost - s.final_cost) <= 2e-5 * s.final_cost


@pytest.mark.gpu
def test_config_L0_numbers(lib):
    """BASELINE.json config 4 with SURVEY Appendix D read literally (1000 cameras on a radius-40 ring: 0.25-unit baselines, no
    triangulation-angle filter; synth config L0).  A much harder, ill-conditioned problem than the headline workload: the FP64
    LM trajectories of two exact solvers part at accept / reject decisions (round 2 measured 35 LM iterations on the GPU
    against 36 in the C restatement, RMSE difference 1.5e-5 px, relative poses within 1.3e-4 rad / 1.8e-3).  Asserted here so
    that the numbers are a test, not a line in profiles/: iteration counts within +-4 (round 3: the GPU takes 38; the C restatement
    itself takes 35 or 36 from one run to the next — its OpenMP reductions combine the threads' partial sums in arrival order — so
    the +-2 of the first version of this test failed one run in two), reference-style RMSE within 2e-5 px, relative pose of
    covisible pairs within 5e-4 rad / 5e-3 (measured 6.6e-6 px, 7e-5 rad / 7.8e-4), and the cost of the returned state is the
    reported one."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, parity, synth
    if not ba_cpu.available():
        pytest.skip("oracle/_build/libba_cpu.so is not built")
    d = synth.make_problem(**synth.CONFIGS["L0"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options())
    c1 = {k: np.array(v, copy=True) for k, v in arr.items()}
    s1 = ba_cpu.solve(c1, threads=8)
    n_res = 2 * arr["obs_cam"].shape[0]
    it_gpu, it_cpu = s.n_successful + s.n_unsuccessful, s1["n_successful"] + s1["n_unsuccessful"]
    d_rmse = abs(math.sqrt(s.final_cost / n_res) - math.sqrt(s1["final_cost"] / n_res))
    pairs = parity.covisible_pairs(arr["obs_cam"], arr["obs_pt"])
    ang, dtr = parity.relative_pose_difference(prod.cam_q, prod.cam_t, c1["cam_q"], c1["cam_t"], pairs)
    print(f"L0: LM iterations gpu {it_gpu} cpu {it_cpu}, |d rmse| {d_rmse:.2e} px, rel-pose {ang:.2e} rad / {dtr:.2e}")
    assert abs(it_gpu - it_cpu) <= 4
    assert d_rmse <= 2e-5
    assert ang <= 5e-4 and dtr <= 5e-3
    ref = H.to_oracle(dict(arr, cam_q=prod.cam_q, cam_t=prod.cam_t, points=prod.points))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost


@pytest.mark.gpu
def test_config5_size_properties(lib):
    """BASELINE.json config 5 AT ITS SIZE: 7500 photos / 1.8M points / ~8M observations (docs/en/benchmark.md:93,111 give ~7.5k
    registered frames for 1DSfM Trafalgar; the data set is not available offline, so synth.make_collection generates an
    unordered collection of that size with viewpoint clusters, power-law track lengths and shuffled camera ids — 45 000 camera
    unknowns, no band in the natural order: until round 2 only the implicit-Schur PCG ran at this size (block-Jacobi: 1800
    iterations per LM step, 10 s per solve); the exact tile Cholesky does since round 3 — on the reverse Cuthill-McKee chain of
    the camera graph then, since round 4 on its nested dissection (ba_plan.h: ordering 3, a level schedule of ~174 levels; since
    round 5 with the one-launch backward substitution: ~1 s per solve).  No oracle
    finishes at this size, so the checks are the size-independent ones (rec_1dsfm.cc:66-98 runs GBA on exactly this shape):
    termination by a Ceres rule, cost of the returned state (oracle evaluation) = reported cost, bit-reproducibility of a
    second run, the gradient max-norm of the objective falls by > 1e2, the RMSE approaches the noise level."""
    import time
    from xrsfm_amd import capi, synth
    t0 = time.time()
    d = synth.make_collection(**synth.CONFIGS["T"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    n_obs = arr["obs_cam"].shape[0]
    assert arr["cam_q"].shape[0] >= 7500 and arr["points"].shape[0] >= 1_500_000 and n_obs >= 8_000_000
    t1 = time.time()
    ctx = capi.Context(H.to_product(arr))
    opt = capi.default_options(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5)     # GBA fast (ba_solver.cc:630-634)
    s = ctx.run(opt)
    t2 = time.time()
    q, t, P = ctx.download()
    ctx.reset()
    s2 = ctx.run(opt)
    q2, t2_, P2 = ctx.download()
    t3 = time.time()
    print(f"config T: {n_obs} obs, generate {t1 - t0:.1f} s, create+solve {t2 - t1:.1f} s ({s.n_successful}+{s.n_unsuccessful} LM, "
          f"solver {s.linear_solver_used}, {s.pcg_iterations} PCG iterations, solve {s.total_time_s:.2f} s), second run {t3 - t2:.1f} s ({s2.total_time_s:.2f} s), "
          f"rmse {math.sqrt(s.initial_cost / n_obs):.3f} -> {math.sqrt(s.final_cost / n_obs):.3f} px, termination {s.termination}/{s.termination_reason}")
    # the nested dissection of the camera graph (ba_plan.h: ordering 3; round 3: its reverse Cuthill-McKee chain) keeps the exact solve affordable at this size
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1, plan
    assert s.termination in (0, 1) and s.termination_reason in (2, 3, 5)       # tolerance exit, or the iteration cap of GBA-fast
    assert s.final_cost < 0.05 * s.initial_cost
    assert s2.final_cost == s.final_cost and (s2.n_successful, s2.n_unsuccessful) == (s.n_successful, s.n_unsuccessful)
    assert np.array_equal(q, q2) and np.array_equal(t, t2_) and np.array_equal(P, P2)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    # the implicit-Schur PCG (the only path at this size until round 2) from the same start: the same LM decisions, the same
    # cost up to what a truncated linear solve leaves
    ctx.reset()
    sp = ctx.run(capi.default_options(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5, linear_solver=capi.SOLVER_PCG))
    ctx.close()
    print(f"config T, PCG: {sp.n_successful}+{sp.n_unsuccessful} LM, {sp.pcg_iterations} PCG iterations, solve {sp.total_time_s:.2f} s, "
          f"rmse {math.sqrt(sp.final_cost / n_obs):.6f} vs {math.sqrt(s.final_cost / n_obs):.6f} px")
    assert sp.linear_solver_used == capi.SOLVER_PCG and sp.pcg_iterations > 0
    assert (sp.n_successful, sp.n_unsuccessful) == (s.n_successful, s.n_unsuccessful)
    assert abs(math.sqrt(sp.final_cost / n_obs) - math.sqrt(s.final_cost / n_obs)) < 1e-4

    def grad_max(state):
        pr = H.to_oracle(state)
        _, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
        gc = bo._scatter_add(pr.cam_q.shape[0], pr.obs_cam, np.einsum("nij,ni->nj", np.asarray(Fc).reshape(-1, 2, 6), rt))
        gp = bo._scatter_add(pr.points.shape[0], pr.obs_pt, np.einsum("nij,ni->nj", np.asarray(Ep).reshape(-1, 2, 3), rt))
        return max(float(np.abs(gc).max()), float(np.abs(gp).max()))

    g0, g1 = grad_max(arr), grad_max(dict(arr, cam_q=q, cam_t=t, points=P))
    print(f"config T: gradient max-norm {g0:.3e} -> {g1:.3e}")
    assert g1 < 1e-2 * g0


@pytest.mark.gpu
def test_clustered_collection_exact_path_matches_c_restatement(lib, monkeypatch):
    """600 photos in 10 viewpoint clusters, shuffled ids (3600 camera unknowns: inside the dense limit, but the plan takes the
    reverse Cuthill-McKee order because it keeps two thirds of the tiles).  The exact path in that order against (i) the same
    library forced to the natural order (XRSFM_BA_RCM=0: the dense panel schedule the round-2 tests validated) — the
    elimination order must not change the solution beyond rounding — and (ii) the C restatement (natural order,
    block-envelope Cholesky): same LM decisions, RMSE within 1e-6 px, cameras within 1e-5."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, synth
    d = synth.make_collection(600, 30000, seed=5, cams_per_cluster=60)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    assert capi.debug_chol_plan(H.to_product(arr))["ordering"] == 2
    n_res = 2 * arr["obs_cam"].shape[0]
    prod = H.to_product(arr)
    s = capi.solve(prod, capi.default_options(max_iterations=8))
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY
    monkeypatch.setenv("XRSFM_BA_RCM", "0")
    assert capi.debug_chol_plan(H.to_product(arr))["ordering"] == 0
    nat = H.to_product(arr)
    sn = capi.solve(nat, capi.default_options(max_iterations=8))
    monkeypatch.delenv("XRSFM_BA_RCM")
    assert (s.n_successful, s.n_unsuccessful) == (sn.n_successful, sn.n_unsuccessful)
    assert abs(s.final_cost - sn.final_cost) <= 1e-9 * sn.final_cost
    assert np.abs(prod.cam_q - nat.cam_q).max() < 1e-7 and np.abs(prod.cam_t - nat.cam_t).max() < 1e-6
    if ba_cpu.available():
        c1 = {k: np.array(v, copy=True) for k, v in arr.items()}
        s1 = ba_cpu.solve(c1, max_iterations=3, threads=8)
        p3 = H.to_product(arr)
        s3 = capi.solve(p3, capi.default_options(max_iterations=3))
        assert (s3.n_successful, s3.n_unsuccessful) == (s1["n_successful"], s1["n_unsuccessful"])
        assert abs(math.sqrt(s3.final_cost / n_res) - math.sqrt(s1["final_cost"] / n_res)) < 1e-6
        assert np.abs(p3.cam_q - c1["cam_q"]).max() < 1e-5 and np.abs(p3.cam_t - c1["cam_t"]).max() < 1e-5


@pytest.mark.gpu
def test_unordered_collection_nested_dissection_matches_the_chain_order(lib, monkeypatch):
    """2400 photos in 40 small viewpoint clusters on a ring, shuffled ids (244 tile columns): the plan dissects the camera graph
    (ordering 3, ba_plan.h: nd_groups) and factors it on the LEVEL schedule — several tile columns per launch, every level with
    lists split into chunks — instead of the reverse Cuthill-McKee chain on the look-ahead panel schedule (XRSFM_BA_ND=0: what
    round 3 validated).  The elimination order must not change the solution beyond rounding: same LM decisions, same cost to
    1e-9, cameras to 1e-7 / 1e-6; and the result is bit-reproducible."""
    from xrsfm_amd import capi, synth
    d = synth.make_collection(n_cams=2400, n_points=100000, seed=4, cams_per_cluster=60)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1 and 2 * plan["levels"] <= plan["tiles"]
    opt = capi.default_options(max_iterations=6)
    prod = H.to_product(arr)
    s = capi.solve(prod, opt)
    again = H.to_product(arr)
    s_again = capi.solve(again, opt)
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY
    assert s_again.final_cost == s.final_cost and np.array_equal(prod.cam_q, again.cam_q) and np.array_equal(prod.points, again.points)
    # backward substitution of the deep level schedule: ONE launch for all levels (default since round 5: k_lv_bwd_all with its lists
    # walked from the root side) against one launch per level with a column's tiles shared out over workgroups (XRSFM_BA_BWD_ALL=0)
    # and against the push form of the panel schedules (... and XRSFM_BA_BWD_CHUNK=0)
    monkeypatch.setenv("XRSFM_BA_BWD_ALL", "0")
    for chunk_env in (None, "0"):
        if chunk_env is not None:
            monkeypatch.setenv("XRSFM_BA_BWD_CHUNK", chunk_env)
        other = H.to_product(arr)
        so = capi.solve(other, opt)
        assert (s.n_successful, s.n_unsuccessful) == (so.n_successful, so.n_unsuccessful)
        assert abs(s.final_cost - so.final_cost) <= 1e-9 * so.final_cost
        assert np.abs(prod.cam_q - other.cam_q).max() < 1e-7 and np.abs(prod.cam_t - other.cam_t).max() < 1e-6
    monkeypatch.delenv("XRSFM_BA_BWD_CHUNK")
    monkeypatch.delenv("XRSFM_BA_BWD_ALL")
    monkeypatch.setenv("XRSFM_BA_ND", "0")
    plan0 = capi.debug_chol_plan(H.to_product(arr))
    assert plan0["ordering"] == 2 and plan0["level_schedule"] == 0
    chain = H.to_product(arr)
    sc = capi.solve(chain, opt)
    monkeypatch.delenv("XRSFM_BA_ND")
    assert (s.n_successful, s.n_unsuccessful) == (sc.n_successful, sc.n_unsuccessful)
    assert abs(s.final_cost - sc.final_cost) <= 1e-9 * sc.final_cost
    assert np.abs(prod.cam_q - chain.cam_q).max() < 1e-7 and np.abs(prod.cam_t - chain.cam_t).max() < 1e-6


@pytest.mark.gpu
def test_level_look_ahead_of_a_dissected_collection(lib, monkeypatch):
    """Round 6: on the level schedule of a dissected photo collection (BASELINE config 5's shape: ~170 levels of update -> sum ->
    factor) the partial products of a level whose operand columns were factored la_depth + 1 levels earlier ("early": all but a few
    per cent) run on a second stream while the main stream is still at the levels in between; the late ones follow the previous
    level's factor kernel and the fixed-order sum adds both (ba_plan.h: la_depth, sp_slot; xrsfm_ba.hip: chol_factor_solve).
    XRSFM_BA_LA_DEPTH=0 is the round-5 schedule (every chunk in one launch, list order).  The early / late split reorders a
    target's sum, nothing else: same LM decisions, costs to 1e-9, cameras to 1e-7 / 1e-6 — for depth 1, 2 (default) and 3 —
    and a depth run twice is bit-identical (two streams, fixed-order sums: no race)."""
    from xrsfm_amd import capi, synth
    d = synth.make_collection(n_cams=2400, n_points=100000, seed=4, cams_per_cluster=60)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1 and plan["levels"] >= 16
    opt = capi.default_options(max_iterations=6)
    res = {}
    for depth in ("0", "1", "2", "2", "3"):
        monkeypatch.setenv("XRSFM_BA_LA_DEPTH", depth)
        prod = H.to_product(arr)
        s = capi.solve(prod, opt)
        assert s.linear_solver_used == capi.SOLVER_CHOLESKY
        key = depth if depth not in res else depth + "'"
        res[key] = (s, prod.cam_q.copy(), prod.cam_t.copy(), prod.points.copy())
    monkeypatch.delenv("XRSFM_BA_LA_DEPTH")
    s0, q0, t0, P0 = res["0"]
    for key in ("1", "2", "3"):
        s, q, t, P = res[key]
        assert (s.n_successful, s.n_unsuccessful) == (s0.n_successful, s0.n_unsuccessful), key
        assert abs(s.final_cost - s0.final_cost) <= 1e-9 * s0.final_cost, key
        assert np.abs(q - q0).max() < 1e-7 and np.abs(t - t0).max() < 1e-6, key
    a, b = res["2"], res["2'"]
    assert a[0].final_cost == b[0].final_cost and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


@pytest.mark.gpu
def test_twenty_thousand_sequential_cameras_take_the_exact_path(lib):
    """120 000 camera unknowns: the dense tile array of the reduced camera matrix would be 161 GB; the packed form (non-zero tiles of
    the nested-dissection factor only, ba_chol.h: tile_ptr) is 0.24 GB, so AUTO stays on the exact Cholesky path (round 2: PCG).
    Properties only at this size: solver used, convergence, bit-reproducibility, idempotence, and the cost of the returned state
    re-evaluated by the oracle."""
    from xrsfm_amd import capi, synth
    d = synth.make_problem(n_cams=20000, n_points=400000, k_obs=4, seed=13)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 1 and plan["level_schedule"] == 1 and plan["levels"] <= 10 and plan["tiles"] > 2000
    ctx = capi.Context(H.to_product(arr))
    s = ctx.run(capi.default_options())
    q, t, P = ctx.download()
    ctx.reset()
    s2 = ctx.run(capi.default_options())
    q2, t2, P2 = ctx.download()
    ctx.close()
    assert s.linear_solver_used == capi.SOLVER_CHOLESKY and s.termination == 0 and s.final_cost < 0.1 * s.initial_cost
    assert s2.final_cost == s.final_cost and np.array_equal(q, q2) and np.array_equal(P, P2)
    ref = H.to_oracle(dict(arr, cam_q=q, cam_t=t, points=P))
    assert abs(bo.evaluate(ref, ref.cam_q, ref.cam_t, ref.points, want_jac=False) - s.final_cost) <= 1e-9 * s.final_cost
    prod = H.to_product(dict(arr, cam_q=q, cam_t=t, points=P))
    s3 = capi.solve(prod)
    assert s3.n_successful <= 1 and np.abs(prod.cam_q - q).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["random", "sequential", "constant_cameras"])
def test_pcg_gauge_coarse_space_matches_oracle_and_cuts_iterations(lib, monkeypatch, case):
    """Round 5: the PCG path's preconditioner is block-Jacobi + the seven gauge directions of the reconstruction as a coarse space
    (ba_kernels.h: k_pcg_gauge / k_pcg_coarse).  Against the oracle's restatement of the same iteration (Options.pcg_coarse): same
    LM decisions, same results, PCG iteration count within a few per solve; against block-Jacobi alone (XRSFM_BA_PCG_COARSE=0):
    same results, at most 60 % of its iterations on random visibility (where the gauge modes are what PCG spends its time on).
    `constant_cameras`: every rotation and two more translations constant — some gauge directions vanish from the coarse space
    (zero columns of W: their pivots are dropped, the others stay)."""
    from xrsfm_amd import capi, synth
    from oracle import ba_oracle as bo
    if case == "random":
        d = synth.make_problem(n_cams=120, n_points=9000, k_obs=5, seed=31, mode="unordered")
        arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    elif case == "sequential":
        arr = H.make(150, 8000, 4, seed=32)
    else:
        arr = H.make(60, 4000, 4, seed=33)
        cc = arr["cam_const"].copy(); cc[:] |= 1; cc[5] |= 2; cc[17] |= 2; arr["cam_const"] = cc
    opt = dict(max_iterations=8, linear_solver=capi.SOLVER_PCG, pcg_max_iterations=3000)
    ref = H.to_oracle(arr)
    s_ref = bo.solve(ref, bo.Options(max_iterations=8, linear_solver="pcg", pcg_max_iter=3000, pcg_coarse=True))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("XRSFM_BA_PCG_COARSE", mode)
        prod = H.to_product(arr)
        s = capi.solve(prod, capi.default_options(**opt))
        out[mode] = (s, prod)
    monkeypatch.delenv("XRSFM_BA_PCG_COARSE")
    s, prod = out["1"]
    s0, prod0 = out["0"]
    assert s.linear_solver_used == capi.SOLVER_PCG
    assert (s.n_successful, s.n_unsuccessful) == (s_ref.n_successful, s_ref.n_unsuccessful) == (s0.n_successful, s0.n_unsuccessful)
    assert abs(s.final_cost - s_ref.final_cost) <= 1e-9 * s_ref.final_cost
    assert max(np.abs(prod.cam_q - ref.cam_q).max(), np.abs(prod.cam_t - ref.cam_t).max()) < 1e-7
    assert max(np.abs(prod.cam_q - prod0.cam_q).max(), np.abs(prod.cam_t - prod0.cam_t).max()) < 1e-7
    print(f"{case}: PCG iterations {s.pcg_iterations} with the gauge coarse space (oracle {s_ref.pcg_iterations}), {s0.pcg_iterations} with block-Jacobi alone")
    if case == "random":
        assert abs(s.pcg_iterations - s_ref.pcg_iterations) <= max(4, 0.1 * s_ref.pcg_iterations), (s.pcg_iterations, s_ref.pcg_iterations)
        assert s.pcg_iterations <= 0.6 * s0.pcg_iterations
    else:       # (a camera chain takes thousands of iterations either way: rounding decides the exact count; the coarse space must not cost any)
        assert s.pcg_iterations <= 1.05 * s0.pcg_iterations + 8
