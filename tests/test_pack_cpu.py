"""Host logic without a GPU: the track-tile packing of xrsfm_ba_create (xrsfm_amd/csrc/ba_pack.h) through xrsfm_ba_debug_pack."""
import numpy as np
import pytest

from tests import helpers as H


def _check_invariants(arr, st):
    n_obs = arr["obs_cam"].shape[0]
    so = st["slot_obs"]
    assert st["slots"] == 64 * st["tiles"] == so.shape[0]
    used = so[so >= 0]
    assert np.array_equal(np.sort(used), np.arange(n_obs))                   # every observation exactly once
    pt = np.where(so >= 0, arr["obs_pt"][np.clip(so, 0, None)], -1)
    cam = np.where(so >= 0, arr["obs_cam"][np.clip(so, 0, None)], -1)
    # a track is contiguous, ordered by camera, and never straddles a tile unless it is a long item of whole tiles
    for j in np.unique(arr["obs_pt"]):
        idx = np.nonzero(pt == j)[0]
        assert idx[-1] - idx[0] + 1 == len(idx)
        assert np.all(np.diff(cam[idx]) > 0)
        if len(idx) <= 64:
            assert idx[0] // 64 == idx[-1] // 64
        else:
            assert idx[0] % 64 == 0
    # padding only at the end of a tile
    for t in range(st["tiles"]):
        v = so[64 * t:64 * t + 64] >= 0
        assert v[0] and not np.any(v[1:] & ~v[:-1])
    assert st["active_points"] == len(np.unique(arr["obs_pt"]))


@pytest.mark.parametrize("case", ["sequential", "unordered", "ragged", "long"])
def test_packing_invariants(lib, case):
    from xrsfm_amd import capi
    if case == "sequential":
        arr = H.make(40, 3000, 4, seed=190)
    elif case == "unordered":
        arr = H.make(25, 800, 5, seed=191, mode="unordered")
    elif case == "ragged":
        arr = H.make(14, 700, 6, seed=192, mode="unordered", min_tri_angle_deg=0.5)
        rng = np.random.default_rng(1)
        keep = rng.random(arr["obs_cam"].shape[0]) < 0.6
        keep[np.unique(arr["obs_pt"], return_index=True)[1]] = True
        perm = rng.permutation(int(keep.sum()))
        for k in ("obs_cam", "obs_pt", "obs_uv"):
            arr[k] = np.ascontiguousarray(arr[k][keep][perm])
    else:
        arr = H.make(150, 9, 140, seed=193, mode="unordered", min_tri_angle_deg=0.5)
    st = capi.debug_pack(H.to_product(arr))
    _check_invariants(arr, st)
    if case == "sequential":
        assert st["regular_tiles"] > 0.8 * st["tiles"] and st["long_items"] == 0 and st["longest_track"] == 4
        assert st["cam_entries"] < 0.3 * arr["obs_cam"].shape[0]            # regular tiles emit one partial per camera
    if case == "long":
        assert st["long_items"] == 9 and st["longest_track"] == 140 and st["items"] == 9 and st["tiles"] == 27
        assert st["cam_entries"] == arr["obs_cam"].shape[0]


def test_packing_edge_cases(lib):
    from xrsfm_amd import capi
    empty = capi.ProblemArrays(cam_q=np.zeros((0, 4)), cam_t=np.zeros((0, 3)), cam_intr=np.zeros(0, np.int32),
                               intr_model=np.zeros(0, np.int32), intr_params=np.zeros((0, 8)), points=np.zeros((0, 3)),
                               obs_cam=np.zeros(0, np.int32), obs_pt=np.zeros(0, np.int32), obs_uv=np.zeros((0, 2)))
    st = capi.debug_pack(empty)
    assert st["tiles"] == 0 and st["items"] == 0
    arr = H.make(6, 40, 3, seed=194)
    bad = H.to_product(arr); bad.obs_pt[3] = 10 ** 6
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.debug_pack(bad)
    bad = H.to_product(arr); bad.intr_model[0] = 7
    with pytest.raises(RuntimeError, match="EINVAL"):
        capi.debug_pack(bad)
    # points without observations are not active
    arr["points"] = np.concatenate([arr["points"], np.zeros((3, 3))]); arr["point_const"] = np.zeros(43, np.uint8)
    assert capi.debug_pack(H.to_product(arr))["active_points"] == 40


@pytest.mark.parametrize("kind", ["regular", "ragged", "wide"])
def test_gram_tiles_and_item_classes(kind):
    """S-assembly side of the packing (ba_pack.h / ba_plan.h): Gram tiles, their camera indices, the per-camera writers and the
    item classes of k_schur_pairs."""
    from xrsfm_amd import capi
    if kind == "regular":
        arr = H.make(60, 3000, 4, seed=11)                       # 50 tracks per tuple: groups start on tile boundaries
    elif kind == "ragged":
        arr = H.make(60, 3000, 8, seed=12, dropout=0.35)
    else:
        arr = H.make(40, 60, 30, seed=13, mode="unordered")      # 30-camera tracks: more than 10 distinct cameras per tile
    g = capi.debug_pack_gram(H.to_product(arr))
    so, ncam, cidx, cpg = g["slot_obs"], g["tile_ncam"], g["slot_cidx"], g["slot_campos_g"]
    cam = np.where(so >= 0, arr["obs_cam"][np.clip(so, 0, None)], -1)
    pt = np.where(so >= 0, arr["obs_pt"][np.clip(so, 0, None)], -1)
    n_tiles = ncam.shape[0]
    assert g["items_small"] + g["items_big"] + g["items_other"] == g["items"]
    assert g["items_small"] + g["items_big"] == g["gram_tiles"] == int((ncam > 0).sum())
    expect_entries, expect_writes = 0, 0
    for t in range(n_tiles):
        sl = slice(64 * t, 64 * t + 64)
        c, p_, ci, cp = cam[sl], pt[sl], cidx[sl], cpg[sl]
        valid = c >= 0
        if ncam[t] > 0:
            distinct = np.unique(c[valid])
            assert ncam[t] == len(distinct) <= 10 and len(distinct) >= 2
            assert np.array_equal(distinct[ci[valid]], c[valid])                  # index into the ascending camera list
            # exactly one writer per distinct camera: its first lane
            for k, cc in enumerate(distinct):
                lanes = np.nonzero(valid & (c == cc))[0]
                assert cp[lanes[0]] >= 0 and (cp[lanes[1:]] < 0).all()
            expect_entries += len(distinct)
            pairs = set()
            for j in np.unique(p_[valid]):
                cs_ = np.sort(c[valid & (p_ == j)])
                pairs.update((a, b) for i, a in enumerate(cs_) for b in cs_[i + 1:])
            expect_writes += len(pairs)                                           # one partial block per co-visible pair
        else:
            assert (ci[valid] == 255).all() if valid.any() else True
            expect_entries += int((cp >= 0).sum())
            for j in np.unique(p_[valid]):
                n = int((valid & (p_ == j)).sum())
                expect_writes += n * (n - 1) // 2 if g["long_items"] == 0 else 0
    assert g["cam_entries_g"] == expect_entries
    if g["long_items"] == 0:
        assert g["block_writes"] == expect_writes
    if kind == "regular":
        assert g["items_other"] <= 0.05 * g["items"] and g["items_big"] == 0     # one launch class for sequential data
    if kind == "ragged":
        assert g["gram_tiles"] >= 0.9 * n_tiles
    if kind == "wide":
        assert g["gram_tiles"] == 0 and g["items_other"] == g["items"]


def test_bal9_problems_pack_per_observation_and_plan_with_nine_rows_per_camera(lib):
    """bal9 mode (cam_const bit 2): no regular-tile pre-reductions (the 9-wide linearisation works per observation), Gram tiles of
    at most 7 cameras (63 operand rows: k9_pairs_gram, round 4) — a tile with more cameras takes the per-pair path —, 7 cameras x 9
    rows per 64-row tile in the plan; the bit is refused for the reference's camera models and for shared intrinsics entries."""
    from xrsfm_amd import capi
    arr = H.make_bal9(40, 2000, 4, seed=5)
    st = capi.debug_pack(H.to_product(arr))
    g = capi.debug_pack_gram(H.to_product(arr))
    assert st["regular_tiles"] == 0 and st["cam_entries"] == arr["obs_cam"].shape[0]
    assert g["gram_tiles"] >= 0.9 * st["tiles"] and 2 <= g["max_cams"] <= 7 and g["items_small"] + g["items_big"] == g["gram_tiles"]
    assert g["cam_entries_g"] < 0.5 * st["cam_entries"]            # one S-assembly entry per distinct camera of a Gram tile
    wide_tracks = H.make_bal9(40, 200, 9, seed=6)                  # 9-camera tracks: more than 7 distinct cameras per tile
    g9 = capi.debug_pack_gram(H.to_product(wide_tracks))
    assert g9["gram_tiles"] == 0 and g9["items_other"] == g9["items"]
    plan = capi.debug_chol_plan(H.to_product(arr))
    off = plan["cam_offset"]
    assert plan["tiles"] == 6 and np.all(off % 64 % 9 == 0) and np.all(off % 64 <= 54) and len(set(off.tolist())) == 40
    bad = dict(arr); bad["cam_intr"] = np.zeros(40, np.int32)
    with pytest.raises(RuntimeError):
        capi.debug_pack(H.to_product(bad))
    bad = dict(arr); bad["intr_model"] = np.full(40, 2, np.int32)
    with pytest.raises(RuntimeError):
        capi.debug_pack(H.to_product(bad))


@pytest.mark.parametrize("C", list(range(1, 11)))
def test_gram_schedule_of_4x4_blocks_covers_every_camera_pair_element_once(lib, C):
    """Round 5 (ba_chol.h: gram_tile4): the camera-pair blocks of a Gram tile of C cameras are formed from 4x4 result blocks of the
    [6C x 6C] Gram matrix, four per v_mfma_f64_4x4x4 instruction.  The schedule (a constexpr table of the library, read back through
    xrsfm_ba_debug_gram_schedule) must list every block (row group >= column group) that holds an element of a camera pair
    (camera of the row > camera of the column) exactly once and no block without one, padded to whole instructions with block
    (0, 0) — which lies inside camera 0 and stores nothing; tiles with more than 6 instructions keep the 16x16 form."""
    from xrsfm_amd import capi
    s = capi.debug_gram_schedule(C)
    R = 6 * C
    wanted = {(r, c) for r in range(R) for c in range(R) if r // 6 > c // 6}
    blocks = s["blocks"]
    assert len(blocks) == 4 * s["n_inst_all"]
    real = [b for b in blocks if b != (0, 0)]
    assert blocks[:len(real)] == real and all(b == (0, 0) for b in blocks[len(real):]) and len(blocks) - len(real) < 4
    assert len(set(real)) == len(real) and all(rg >= cg for rg, cg in real)
    covered = []
    for rg, cg in real:
        el = [(4 * rg + i, 4 * cg + j) for i in range(4) for j in range(4) if 4 * rg + i < R and 4 * cg + j < R and (4 * rg + i) // 6 > (4 * cg + j) // 6]
        assert el, (rg, cg)                 # no block without a wanted element
        covered += el
    assert len(covered) == len(set(covered)) and set(covered) == wanted
    assert s["n_inst_all"] == (len(real) + 3) // 4
    assert s["n_inst"] == (s["n_inst_all"] if s["n_inst_all"] <= 6 else 0)
    if C == 4:
        assert (len(real), s["n_inst"]) == (17, 5)          # the tiles of a sequential map: 17 blocks of 16 for 6 blocks of 36
