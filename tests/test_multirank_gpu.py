"""Two or three ranks of the HIP path on ONE GPU (RCCL refuses two ranks on one device, so the all-reduces go through the library's
test transport hook + torch.distributed/gloo on host copies): points sharded by the length-aware partition of
xrsfm_amd/sharding.py (bench.shard_problem), cameras replicated.  Every rank must
take the same LM decisions and end with the same cameras as the single-rank solve; the union of the ranks' points must equal
the single-rank points.  Covers both linear solvers (the Cholesky path also needs the union block pattern)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H


_rdzv_count = [0]


def _free_port():
    """Rendezvous token of one spawn: a fresh FILE (torch's file:// store) — a TCP port picked by bind(0) can be taken again
    before the workers listen on it (seen once on the GPU box: EADDRINUSE)."""
    import tempfile
    _rdzv_count[0] += 1
    return os.path.join(tempfile.gettempdir(), f"xba_rdzv_{os.getpid()}_{_rdzv_count[0]}")


def _worker(rank, world, port, arr, solver, opt_kw, out_prefix):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import shard_problem
    from xrsfm_amd import capi
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)

    def allreduce(buf, op):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX)

    local = arr[rank] if isinstance(arr, list) else shard_problem(arr, rank, world)      # list: one ready-made shard per rank
    ctx = capi.Context(H.to_product(local))
    ctx.comm_hook(world, rank, allreduce)
    s = ctx.run(capi.default_options(linear_solver=solver, **opt_kw))
    q, t, P = ctx.download()
    np.savez(f"{out_prefix}{rank}.npz", q=q, t=t, P=P, stat=np.array([s.n_successful, s.n_unsuccessful, s.termination_reason]),
             cost=np.array([s.initial_cost, s.final_cost]))
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("solver", [1, 0], ids=["cholesky", "pcg"])
@pytest.mark.parametrize("mode,world", [("sequential", 2), ("unordered", 2), ("sequential", 3), ("ragged", 2),
                                        ("unordered-panels", 2), ("clustered", 2)])
def test_ranks_equal_one_rank(lib, tmp_path, solver, mode, world):
    """(round 3) "unordered-panels": 130 cameras with random visibility = 13 tile columns: the look-ahead panel schedule
    (k_panel_slot) on every rank over the union block pattern; "clustered": a small photo collection with viewpoint clusters in
    the reverse Cuthill-McKee order (every rank must derive the same order from the all-reduced pattern)."""
    from xrsfm_amd import capi, synth
    if mode == "ragged":
        arr = H.make(24, 1500, 8, seed=140, dropout=0.35)
    elif mode == "unordered-panels":
        arr = H.make(130, 4000, 5, seed=141, mode="unordered")
    elif mode == "clustered":
        d = synth.make_collection(n_cams=600, n_points=30000, seed=5, cams_per_cluster=60)
        arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    else:
        arr = H.make(24, 1500, 4, seed=140, mode=mode)
    if solver == 0 and mode in ("unordered-panels", "clustered"):
        pytest.skip("the larger cases are about the exact path's schedules")
    if mode in ("unordered-panels", "clustered"):
        plan = capi.debug_chol_plan(H.to_product(arr))
        assert plan["lookahead"] == 1 and (mode != "clustered" or plan["ordering"] == 2), plan
    opt_kw = dict(max_iterations=8)
    ref = H.to_product(arr)
    s1 = capi.solve(ref, capi.default_options(linear_solver=solver, **opt_kw))
    prefix = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), arr, solver, opt_kw, prefix), nprocs=world, join=True)
    z = [np.load(f"{prefix}{r}.npz") for r in range(world)]
    n_res = 2 * arr["obs_cam"].shape[0]
    for r in range(world):
        assert tuple(z[r]["stat"]) == (s1.n_successful, s1.n_unsuccessful, s1.termination_reason)
        assert abs(z[r]["cost"][0] - s1.initial_cost) <= 1e-12 * s1.initial_cost          # all-reduced: the global cost
        assert abs(np.sqrt(z[r]["cost"][1] / n_res) - np.sqrt(s1.final_cost / n_res)) < 1e-6
        assert np.abs(z[r]["q"] - ref.cam_q).max() < 1e-5 and np.abs(z[r]["t"] - ref.cam_t).max() < 1e-5
    # both ranks hold bit-identical cameras (same reduced system, same factorisation on every rank)
    for r in range(1, world):
        assert np.array_equal(z[0]["q"], z[r]["q"]) and np.array_equal(z[0]["t"], z[r]["t"])
    n_p = arr["points"].shape[0]
    # (ragged tracks leave points seen by two neighbouring frames only: their depth amplifies the 1e-9 differences of the
    # cameras by five orders of magnitude, at no difference in cost)
    ptol = 1e-2 if mode == "ragged" else (1e-4 if mode == "clustered" else 1e-5)
    from xrsfm_amd import sharding
    owner = sharding.partition_points(arr["obs_pt"], n_p, world)          # what bench.shard_problem applied on every rank
    assert sharding.imbalance(arr["obs_pt"], owner, world) <= 0.01
    for r in range(world):
        assert np.abs(z[r]["P"] - ref.points[owner == r]).max() < ptol


@pytest.mark.gpu
@pytest.mark.parametrize("world,fused", [(2, "1"), (4, "0"), (8, "1"), (8, "0")])
def test_config_s_sized_problem_on_2_4_8_ranks(lib, tmp_path, monkeypatch, world, fused):
    """BASELINE.json config 2's size (100 cameras / 50 000 points / 200 000 observations) split over
    2, 4 and 8 ranks that share the GPU through the transport hook, with the one-launch linearisation tail (single-rank form)
    replaced by the multi-rank sequence either way and XRSFM_BA_FUSED = 0 / 1 for the rest: every rank takes the single-rank
    solve's LM decisions, all ranks hold bit-identical cameras, cameras within 1e-5 and RMSE within 1e-6 px of the single-rank
    solve, and the shards' observation counts balance within 1 %."""
    from xrsfm_amd import capi, sharding, synth
    monkeypatch.setenv("XRSFM_BA_FUSED", fused)
    d = synth.make_problem(**synth.CONFIGS["S"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    opt_kw = dict(max_iterations=6)
    ref = H.to_product(arr)
    s1 = capi.solve(ref, capi.default_options(linear_solver=1, **opt_kw))
    prefix = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), arr, 1, opt_kw, prefix), nprocs=world, join=True)
    z = [np.load(f"{prefix}{r}.npz") for r in range(world)]
    n_res = 2 * arr["obs_cam"].shape[0]
    owner = sharding.partition_points(arr["obs_pt"], arr["points"].shape[0], world)
    assert sharding.imbalance(arr["obs_pt"], owner, world) <= 0.01
    for r in range(world):
        assert tuple(z[r]["stat"]) == (s1.n_successful, s1.n_unsuccessful, s1.termination_reason)
        assert abs(np.sqrt(z[r]["cost"][1] / n_res) - np.sqrt(s1.final_cost / n_res)) < 1e-6
        assert np.abs(z[r]["q"] - ref.cam_q).max() < 1e-5 and np.abs(z[r]["t"] - ref.cam_t).max() < 1e-5
        assert np.array_equal(z[0]["q"], z[r]["q"]) and np.array_equal(z[0]["t"], z[r]["t"])
        assert np.abs(z[r]["P"] - ref.points[owner == r]).max() < 1e-5


@pytest.mark.gpu
def test_weak_scaling_shards_equal_their_union(lib, tmp_path):
    """bench.py --scaling weak: every rank holds its own generated shard (synth.make_problem(point_seed=rank)).  Two such ranks
    must take the LM decisions of, and end with the cameras of, the single-rank solve of the union problem."""
    from xrsfm_amd import capi, synth
    cfg = dict(n_cams=24, n_points=1200, k_obs=4, seed=141)
    fields = capi.ProblemArrays.FIELDS
    shards = [{k: v for k, v in synth.make_problem(**cfg, point_seed=r).items() if k in fields} for r in range(2)]
    union = dict(shards[0])
    union["points"] = np.vstack([sh["points"] for sh in shards])
    union["point_const"] = np.concatenate([sh["point_const"] for sh in shards])
    union["obs_cam"] = np.concatenate([sh["obs_cam"] for sh in shards])
    union["obs_pt"] = np.concatenate([shards[0]["obs_pt"], shards[1]["obs_pt"] + shards[0]["points"].shape[0]]).astype(np.int32)
    union["obs_uv"] = np.vstack([sh["obs_uv"] for sh in shards])
    opt_kw = dict(max_iterations=8)
    ref = H.to_product(union)
    s1 = capi.solve(ref, capi.default_options(linear_solver=1, **opt_kw))
    prefix = str(tmp_path / "weak")
    mp.spawn(_worker, args=(2, _free_port(), shards, 1, opt_kw, prefix), nprocs=2, join=True)
    z = [np.load(f"{prefix}{r}.npz") for r in range(2)]
    n_res = 2 * union["obs_cam"].shape[0]
    n0 = shards[0]["points"].shape[0]
    for r in range(2):
        assert tuple(z[r]["stat"]) == (s1.n_successful, s1.n_unsuccessful, s1.termination_reason)
        assert abs(z[r]["cost"][0] - s1.initial_cost) <= 1e-12 * s1.initial_cost
        assert abs(np.sqrt(z[r]["cost"][1] / n_res) - np.sqrt(s1.final_cost / n_res)) < 1e-6
        assert np.abs(z[r]["q"] - ref.cam_q).max() < 1e-5 and np.abs(z[r]["t"] - ref.cam_t).max() < 1e-5
        assert np.abs(z[r]["P"] - ref.points[r * n0:(r + 1) * n0]).max() < 1e-5
    assert np.array_equal(z[0]["q"], z[1]["q"]) and np.array_equal(z[0]["t"], z[1]["t"])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_config4_sized_problem_on_2_and_8_ranks(lib, tmp_path, world):
    """BASELINE.json config 4's size — 1000 cameras / 500 000 points / 2 000 000 observations, the parity
    workload LP (config 4 + 24 hub frames x 50 distant landmarks: the plain ring L leaves 1e-3 of gauge drift, on which no two
    summation orders agree to 1e-5) — split over 2 and 8 ranks that share the GPU through the transport hook:
    the complete multi-rank HIP path (union block pattern, length-aware shards, camS | Sblk and camlin all-reduces, the replicated
    level schedule with its one-launch backward substitution) at the size the driver's 8-GPU run uses.  Every rank takes the
    single-rank solve's LM decisions, all ranks hold bit-identical cameras, cameras within 1e-5 and RMSE within 1e-6 px of the
    single-rank solve, shards balanced within 1 %."""
    from xrsfm_amd import capi, sharding, synth
    d = synth.make_problem(**synth.CONFIGS["LP"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    opt_kw = dict(max_iterations=50)
    ref = H.to_product(arr)
    s1 = capi.solve(ref, capi.default_options(linear_solver=1, **opt_kw))
    assert s1.n_successful + s1.n_unsuccessful >= 10
    prefix = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), arr, 1, opt_kw, prefix), nprocs=world, join=True)
    z = [np.load(f"{prefix}{r}.npz") for r in range(world)]
    n_res = 2 * arr["obs_cam"].shape[0]
    owner = sharding.partition_points(arr["obs_pt"], arr["points"].shape[0], world)
    assert sharding.imbalance(arr["obs_pt"], owner, world) <= 0.01
    for r in range(world):
        assert tuple(z[r]["stat"]) == (s1.n_successful, s1.n_unsuccessful, s1.termination_reason)
        assert abs(np.sqrt(z[r]["cost"][1] / n_res) - np.sqrt(s1.final_cost / n_res)) < 1e-6
        assert np.abs(z[r]["q"] - ref.cam_q).max() < 1e-5 and np.abs(z[r]["t"] - ref.cam_t).max() < 1e-5
        assert np.array_equal(z[0]["q"], z[r]["q"]) and np.array_equal(z[0]["t"], z[r]["t"])
    print(f"config-4-sized LP on {world} ranks: LM {s1.n_successful}+{s1.n_unsuccessful}, max camera difference to one rank "
          f"{max(np.abs(z[0]['q'] - ref.cam_q).max(), np.abs(z[0]['t'] - ref.cam_t).max()):.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_dissected_collection_on_2_4_8_ranks(lib, tmp_path, world):
    """VERDICT round 5, item 3 (its test half): a photo collection in the shape of BASELINE config 5 — 2400 photos in 40 viewpoint
    clusters, 100 000 tracks with a power-law length distribution — split over 2, 4 and 8 ranks that share the GPU through the
    transport hook.  Every rank derives the SAME nested dissection of the camera graph from the all-reduced block pattern
    (ordering 3), the same level schedule with its level look-ahead on a second stream (round 6) and the same stored-operand
    block sums; the factorisation is replicated.  Every rank takes the single-rank solve's LM decisions, all ranks hold
    bit-identical cameras, cameras within 1e-5 and RMSE within 1e-6 px of the single-rank solve, shards balanced within 1 %."""
    from xrsfm_amd import capi, sharding, synth
    d = synth.make_collection(n_cams=2400, n_points=100000, seed=4, cams_per_cluster=60)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1 and plan["levels"] >= 16
    opt_kw = dict(max_iterations=6)
    ref = H.to_product(arr)
    s1 = capi.solve(ref, capi.default_options(linear_solver=1, **opt_kw))
    prefix = str(tmp_path / "rank")
    mp.spawn(_worker, args=(world, _free_port(), arr, 1, opt_kw, prefix), nprocs=world, join=True)
    z = [np.load(f"{prefix}{r}.npz") for r in range(world)]
    n_res = 2 * arr["obs_cam"].shape[0]
    owner = sharding.partition_points(arr["obs_pt"], arr["points"].shape[0], world)
    assert sharding.imbalance(arr["obs_pt"], owner, world) <= 0.01
    for r in range(world):
        assert tuple(z[r]["stat"]) == (s1.n_successful, s1.n_unsuccessful, s1.termination_reason)
        assert abs(np.sqrt(z[r]["cost"][1] / n_res) - np.sqrt(s1.final_cost / n_res)) < 1e-6
        assert np.abs(z[r]["q"] - ref.cam_q).max() < 1e-5 and np.abs(z[r]["t"] - ref.cam_t).max() < 1e-5
        assert np.array_equal(z[0]["q"], z[r]["q"]) and np.array_equal(z[0]["t"], z[r]["t"])
