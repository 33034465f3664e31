"""Host-side plan of the reduced-camera Cholesky (ba_plan.h) through xrsfm_ba_debug_chol_plan: no GPU needed."""
import numpy as np
import pytest

from tests import helpers as H
from xrsfm_amd import capi


def _check_layout(plan, n_cams):
    off = plan["cam_offset"]
    assert len(set(off.tolist())) == n_cams                       # every camera has its own 6 rows
    assert np.all(off % 64 % 6 == 0) and np.all(off % 64 <= 54)   # 10 cameras per 64-row tile, never straddling a tile
    assert off.max() < 64 * plan["tiles"]
    assert 1 <= plan["levels"] <= plan["tiles"]
    assert plan["tiles_nz"] <= plan["tiles"] * (plan["tiles"] + 1) // 2


def _add_closures(arr, n_extra, seed=4):
    rng = np.random.default_rng(seed)
    n_p = arr["points"].shape[0]
    n_c = arr["cam_q"].shape[0]
    ec, ep, eu, eP = [], [], [], []
    for e in range(n_extra):
        a, b = int(rng.integers(0, n_c // 3)), int(rng.integers(n_c // 2, n_c - 10))
        for cidx in (a, a + 1, b, b + 1):
            ec.append(cidx); ep.append(n_p + e); eu.append(rng.uniform([100, 50], [1100, 300]))
        eP.append(arr["points"][int(rng.integers(0, n_p))] + rng.normal(0, 0.5, 3))
    arr["points"] = np.concatenate([arr["points"], np.array(eP)])
    arr["point_const"] = np.zeros(arr["points"].shape[0], np.uint8)
    arr["obs_cam"] = np.concatenate([arr["obs_cam"], np.array(ec, np.int32)])
    arr["obs_pt"] = np.concatenate([arr["obs_pt"], np.array(ep, np.int32)])
    arr["obs_uv"] = np.concatenate([arr["obs_uv"], np.array(eu)])
    return arr


def test_band_gets_nested_dissection():
    arr = H.make(160, 3000, 4, seed=200)
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 160)
    assert plan["ordering"] == 1 and plan["hubs"] == 0 and plan["band"] == 3
    assert plan["level_schedule"] == 1 and plan["levels"] <= 5
    assert plan["blocks"] == 160 * 3                               # ring of 160 cameras, each sharing tracks with 3 successors


def test_loop_closures_become_hubs():
    arr = _add_closures(H.make(160, 3000, 4, seed=200), 6)
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 160)
    assert plan["ordering"] == 1 and plan["band"] == 3
    assert 0 < plan["hubs"] <= 24
    assert plan["level_schedule"] == 1 and plan["levels"] <= 9
    off = plan["cam_offset"]
    # hub cameras are eliminated last: the ends of a long-range pair sit in the last tiles
    ec = arr["obs_cam"][-24:]
    assert off[ec].min() // 64 >= plan["tiles"] - 3


def test_too_many_closures_fall_back_to_natural_order():
    arr = _add_closures(H.make(160, 3000, 4, seed=200), 40)
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 160)
    assert plan["ordering"] == 0 and plan["hubs"] == 0
    assert plan["tiles"] == 16
    assert np.array_equal(plan["cam_offset"], 64 * (np.arange(160) // 10) + 6 * (np.arange(160) % 10))


def test_dense_visibility_keeps_natural_order():
    arr = H.make(30, 400, 12, seed=3, mode="unordered")
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 30)
    assert plan["tiles"] == 3


def test_duplicate_frame_in_track_is_rejected():
    arr = H.make(12, 60, 3, seed=1)
    arr["obs_cam"] = arr["obs_cam"].copy()
    pt0 = arr["obs_pt"][0]
    idx = np.nonzero(arr["obs_pt"] == pt0)[0]
    arr["obs_cam"][idx[1]] = arr["obs_cam"][idx[0]]
    with pytest.raises(Exception):
        capi.debug_chol_plan(H.to_product(arr))


def test_empty_problem_plan():
    arr = H.make(12, 60, 3, seed=1)
    for k in ("obs_cam", "obs_pt"):
        arr[k] = arr[k][:0]
    arr["obs_uv"] = arr["obs_uv"][:0]
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["blocks"] == 0 and plan["tiles"] >= 1


@pytest.mark.parametrize("n_c,n_p,limit_s", [(20000, 60000, 20.0), (45000, 135000, 8.0)])
def test_large_unordered_problem_is_refused_before_the_symbolic_factorisation(lib, n_c, n_p, limit_s):
    """20 000 cameras with random visibility: the natural-order tile pattern fills in completely, and the symbolic
    factorisation of 1875 x 1875 tiles would need ~10^9 list entries.  The plan must stop with XRSFM_BA_ETOOBIG right after
    the ordering decision (xrsfm_ba_run then takes the PCG path), in well under a second of plan time.
    45 000 cameras (4500 tile columns > kPlanMaxTiles): no order can be accepted, so not even the reverse Cuthill-McKee probe
    (a T x T map and an O(T^2) walk) may run — refused before any T^2 work."""
    import time
    from xrsfm_amd import capi
    rng = np.random.default_rng(3)
    ks = rng.integers(2, 6, n_p)
    obs_pt = np.repeat(np.arange(n_p, dtype=np.int32), ks)
    obs_cam = np.concatenate([rng.choice(n_c, size=k, replace=False) for k in ks]).astype(np.int32)
    arr = dict(cam_q=np.tile([0, 0, 0, 1.0], (n_c, 1)), cam_t=np.zeros((n_c, 3)), cam_const=np.zeros(n_c, np.uint8),
               cam_intr=np.zeros(n_c, np.int32), intr_model=np.array([2], np.int32), intr_params=np.array([[700.0, 600, 200, 0, 0, 0, 0, 0]]),
               points=rng.normal(size=(n_p, 3)), point_const=np.zeros(n_p, np.uint8), obs_cam=obs_cam, obs_pt=obs_pt,
               obs_uv=rng.normal(size=(len(obs_cam), 2)))
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="-6"):
        capi.debug_chol_plan(capi.ProblemArrays(**arr))
    assert time.perf_counter() - t0 < limit_s


@pytest.mark.parametrize("mode,n_cams,k_obs,env", [
    ("unordered", 150, 5, {}),                                                        # panel schedule, chunks
    ("unordered", 150, 5, {"XRSFM_BA_PANEL_MACRO": "1"}),                             # + 128x128 macro tiles, 2 columns
    ("unordered", 150, 5, {"XRSFM_BA_PANEL_MACRO": "1", "XRSFM_BA_PANEL_COLS": "4"}),
    ("unordered", 95, 5, {"XRSFM_BA_PANEL_MACRO": "1"}),                              # odd / even numbers of tile columns
    ("unordered", 230, 5, {"XRSFM_BA_PANEL_MACRO": "1", "XRSFM_BA_PANEL_COLS": "6"}),
    ("unordered", 60, 12, {}),
    ("unordered", 150, 5, {"XRSFM_BA_LOOKAHEAD": "0"}),                               # the round-2 panel schedule (no second stream)
    ("unordered", 150, 5, {"XRSFM_BA_LOOKAHEAD": "0", "XRSFM_BA_PANEL_MACRO": "1"}),
    ("sequential", 130, 4, {}), ("sequential", 257, 3, {}), ("sequential", 400, 4, {}), ("sequential", 560, 4, {}),   # level schedules
])
def test_schedule_covers_the_factorisation(monkeypatch, mode, n_cams, k_obs, env):
    """XRSFM_BA_PLAN_CHECK makes the plan verify itself (ba_plan.h): every structurally non-zero tile (i,k) receives each
    contribution L_ij L_kj^T, j < k, exactly once — from a macro-tile entry, a chunk of a split level or its own list in the
    fused factor kernel — and the forward substitution of row k each L_kj y_j exactly once.  (A mutation that drops one macro
    chunk makes the call fail with EINTERNAL.)"""
    monkeypatch.setenv("XRSFM_BA_PLAN_CHECK", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    arr = H.make(n_cams, 20 * n_cams, k_obs, seed=300 + n_cams, mode=mode)
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["tiles"] <= 96                      # the check covers plans of up to 96 tile columns
    if mode == "unordered":
        assert plan["level_schedule"] == 0 and plan["levels"] == plan["tiles"]
        # a pure chain of >= 8 columns takes the look-ahead form of the panel schedule: partial products cover j < k - 2 only
        assert plan["lookahead"] == (0 if env.get("XRSFM_BA_LOOKAHEAD") == "0" or plan["tiles"] < 8 else 1)
    else:
        assert plan["level_schedule"] == 1 and plan["lookahead"] == 0


def test_clustered_collection_takes_the_rcm_order():
    """An unordered photo collection with viewpoint clusters (synth.make_collection: 10 landmarks on a ring, shuffled camera ids):
    in the natural order the tile pattern of the reduced camera matrix fills in completely; the reverse Cuthill-McKee order of
    the camera graph recovers the band of neighbouring landmarks (ordering 2) and the symbolic factorisation keeps a fraction
    of the tiles.  Random visibility (config U: no clusters) keeps the natural order."""
    from xrsfm_amd import synth
    d = synth.make_collection(1500, 60000, seed=5)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 1500)
    dense = plan["tiles"] * (plan["tiles"] + 1) // 2
    assert plan["ordering"] == 2 and plan["tiles"] == 150 and plan["tiles_nz"] < 0.75 * dense
    # the order follows the landmarks: cameras of one landmark are (nearly) contiguous in the elimination order
    rank = np.argsort(np.argsort(plan["cam_offset"]))
    spread = [np.ptp(rank[d["cluster_of_cam"] == k]) for k in range(10)]
    assert np.median(spread) < 450                                   # a landmark has ~150 cameras, its neighbours 300 more
    u = H.make(500, 20000, 5, seed=5, mode="unordered")
    assert capi.debug_chol_plan(H.to_product(u))["ordering"] == 0


def test_pair_keys_of_a_large_collection_are_sorted_in_parallel():
    """Above 2 M pair keys the key list is radix-sorted on up to 16 threads (ba_plan.h: chol_local_keys; config T has 61.8 M keys).
    The number of distinct camera pairs the plan reports must be the number an independent count over the tracks gives."""
    from xrsfm_amd import synth
    d = synth.make_collection(n_cams=1200, n_points=150000, seed=3)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    order = np.lexsort((arr["obs_cam"], arr["obs_pt"]))
    cam, pt = arr["obs_cam"][order].astype(np.int64), arr["obs_pt"][order]
    ptr = np.searchsorted(pt, np.arange(arr["points"].shape[0] + 1))
    n_pairs = int(sum((ptr[j + 1] - ptr[j]) * (ptr[j + 1] - ptr[j] - 1) // 2 for j in range(len(ptr) - 1)))
    assert n_pairs > 2_000_000                       # the parallel path
    keys = set()
    for j in range(len(ptr) - 1):
        c = cam[ptr[j]:ptr[j + 1]]
        if len(c) > 1:
            a, b = np.triu_indices(len(c), 1)
            keys.update((c[b] * 1200 + c[a]).tolist())
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert plan["blocks"] == len(keys)


def test_schedule_of_a_clustered_collection_covers_the_factorisation(monkeypatch):
    """The self-check of the plan (XRSFM_BA_PLAN_CHECK) on the shape of BASELINE config 5 at a quarter of its size: ~190 tile
    columns in the reverse Cuthill-McKee order, look-ahead panel schedule with sparse columns — chunks of split levels, late
    partials (column k-2) and in-kernel lists (column k-1) must cover every product exactly once."""
    from xrsfm_amd import synth
    monkeypatch.setenv("XRSFM_BA_PLAN_CHECK", "1")
    d = synth.make_collection(n_cams=1900, n_points=120000, seed=4)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    assert 150 <= plan["tiles"] <= 256 and plan["ordering"] == 2 and plan["lookahead"] == 1 and plan["levels"] == plan["tiles"]
    # the same pattern through the round-2 panel schedule (no look-ahead)
    monkeypatch.setenv("XRSFM_BA_LOOKAHEAD", "0")
    plan0 = capi.debug_chol_plan(H.to_product(arr))
    assert plan0["lookahead"] == 0 and plan0["tiles_nz"] == plan["tiles_nz"]


def test_ring_of_small_clusters_is_dissected(monkeypatch):
    """Unordered collection, 40 viewpoint clusters of 60 photos on a ring (244 tile columns): George's nested dissection of the
    camera graph (ordering 3) — elimination tree at most half as deep as the reverse Cuthill-McKee chain, no more tile products than
    1.15 x the chain's, level schedule; the plan's self-check confirms that chunks + in-kernel lists cover every product once.
    XRSFM_BA_ND=0 keeps the chain (look-ahead panel schedule); cameras of one landmark stay close in the elimination order."""
    from xrsfm_amd import synth
    monkeypatch.setenv("XRSFM_BA_PLAN_CHECK", "1")
    d = synth.make_collection(n_cams=2400, n_points=100000, seed=4, cams_per_cluster=60)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, 2400)
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1 and plan["lookahead"] == 0
    assert 2 * plan["levels"] <= plan["tiles"] <= 256
    monkeypatch.setenv("XRSFM_BA_ND", "0")
    chain = capi.debug_chol_plan(H.to_product(arr))
    assert chain["ordering"] == 2 and chain["levels"] == chain["tiles"] and chain["lookahead"] == 1
    assert plan["tiles_nz"] <= 1.15 * chain["tiles_nz"]
    # same plan twice (the dissection is deterministic)
    monkeypatch.delenv("XRSFM_BA_ND")
    again = capi.debug_chol_plan(H.to_product(arr))
    assert np.array_equal(again["cam_offset"], plan["cam_offset"])


def test_dissection_of_a_disconnected_collection(monkeypatch):
    """Two photo collections that share nothing (two rings of 20 viewpoint clusters each, merged into one problem with interleaved
    camera ids): the camera graph has two components.  The dissection treats the components as parts of their own; the plan must
    cover the factorisation (self-check), keep every camera, and put no tile of one collection into a column list of the other —
    the non-zero tile count is the sum of the two separate plans' counts up to the padding of the parts to tile boundaries."""
    from xrsfm_amd import synth
    monkeypatch.setenv("XRSFM_BA_PLAN_CHECK", "1")
    a = synth.make_collection(n_cams=1200, n_points=50000, seed=7, cams_per_cluster=60)
    b = synth.make_collection(n_cams=1200, n_points=50000, seed=8, cams_per_cluster=60)
    na, pa = a["cam_q"].shape[0], a["points"].shape[0]
    perm = np.random.default_rng(3).permutation(2 * na)            # new id of camera c of a: perm[c]; of b: perm[na + c]
    inv = np.argsort(perm)
    merged = {k: np.concatenate([a[k], b[k]])[inv] for k in ("cam_q", "cam_t", "cam_const", "cam_intr")}      # row perm[c] = camera c
    merged.update({k: a[k] for k in ("intr_model", "intr_params")})                                         # (one shared intrinsics group)
    merged.update({k: np.concatenate([a[k], b[k]]) for k in ("points", "point_const", "obs_uv")})
    merged["obs_cam"] = np.concatenate([perm[a["obs_cam"]], perm[na + b["obs_cam"]]]).astype(np.int32)
    merged["obs_pt"] = np.concatenate([a["obs_pt"], pa + b["obs_pt"]]).astype(np.int32)
    plan = capi.debug_chol_plan(H.to_product(merged))
    _check_layout(plan, 2 * na)
    assert plan["ordering"] == 3 and plan["level_schedule"] == 1
    pa_ = capi.debug_chol_plan(H.to_product({k: a[k] for k in capi.ProblemArrays.FIELDS}))
    pb_ = capi.debug_chol_plan(H.to_product({k: b[k] for k in capi.ProblemArrays.FIELDS}))
    assert plan["blocks"] == pa_["blocks"] + pb_["blocks"]
    # no coupling between the collections: their elimination trees stand side by side (depth ~ the deeper one — not exactly: the part
    # size follows the camera count and ties break by id), they are not chained (depth = the sum)
    assert plan["levels"] <= 1.25 * max(pa_["levels"], pb_["levels"]) < pa_["levels"] + pb_["levels"]
    off = plan["cam_offset"]
    tiles_a, tiles_b = set((off[perm[:na]] // 64).tolist()), set((off[perm[na:]] // 64).tolist())
    assert not (tiles_a & tiles_b)                                   # parts start on tile boundaries: no tile mixes the collections


@pytest.mark.parametrize("n_cams,cpc,seed", [(1000, 25, 1), (1600, 40, 2), (2200, 110, 3), (2400, 50, 9)])
def test_dissected_plans_are_valid_permutations_and_cover_the_factorisation(monkeypatch, n_cams, cpc, seed):
    """Collections of different cluster sizes (25 .. 110 photos per landmark, 100 .. 250 tile columns): whichever order the plan
    takes (dissection or chain), every camera gets its own six rows, the plan's self-check passes (every product of the symbolic
    factorisation covered exactly once by chunks / in-kernel lists), and forcing the other order (XRSFM_BA_ND=0) gives a valid plan
    over the same blocks."""
    from xrsfm_amd import synth
    monkeypatch.setenv("XRSFM_BA_PLAN_CHECK", "1")
    d = synth.make_collection(n_cams=n_cams, n_points=40 * n_cams, seed=seed, cams_per_cluster=cpc)
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    plan = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(plan, n_cams)
    assert plan["ordering"] in (2, 3) and plan["tiles"] <= 256
    if plan["ordering"] == 3:
        assert plan["level_schedule"] == 1 and 2 * plan["levels"] <= plan["tiles"]
    monkeypatch.setenv("XRSFM_BA_ND", "0")
    chain = capi.debug_chol_plan(H.to_product(arr))
    _check_layout(chain, n_cams)
    assert chain["ordering"] == 2 and chain["blocks"] == plan["blocks"]
