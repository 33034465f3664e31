"""Oracle (numpy) and C restatement against the committed golden fixtures; C restatement against the numpy oracle."""
import glob
import math
import os

import numpy as np
import pytest

from oracle import ba_cpu, ba_oracle as bo
from tests import helpers as H

GOLD_DIR = os.path.join(os.path.dirname(__file__), "golden")
OTHER = {"track_filter.npz", "tag_refine.npz", "pose_graph.npz", "lba_selection.npz", "wide_bal9.npz"}   # fixtures of other rows, tested elsewhere
GOLD = sorted(p for p in glob.glob(os.path.join(GOLD_DIR, "*.npz")) if os.path.basename(p) not in OTHER and not os.path.basename(p).startswith("ceres_"))
CERES = sorted(glob.glob(os.path.join(GOLD_DIR, "ceres_*.npz")))          # real Ceres runs (bench/make_ceres_golden.py); none committed yet


def _load(path):
    z = np.load(path)
    arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    mi, ft, pt, rad = z["opt"]
    return z, arr, dict(max_iterations=int(mi), function_tolerance=float(ft), parameter_tolerance=float(pt), initial_radius=float(rad))


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not ba_cpu.available():
        ba_cpu.build()


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_numpy_oracle_reproduces_golden(path):
    z, arr, opt = _load(path)
    pr = H.to_oracle(arr)
    cost0, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    assert abs(cost0 - float(z["init_cost"])) <= 1e-12 * cost0
    assert H.rel_err(rt, z["init_r"]) < 1e-12 and H.rel_err(Fc, z["init_Jc"]) < 1e-12 and H.rel_err(Ep, z["init_Jp"]) < 1e-12
    s = bo.solve(pr, bo.Options(**opt))
    assert (s.n_successful, s.n_unsuccessful) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(s.final_cost - float(z["final_cost"])) <= 1e-9 * s.final_cost
    assert np.abs(pr.cam_q - z["out_cam_q"]).max() < 1e-8 and np.abs(pr.cam_t - z["out_cam_t"]).max() < 1e-8


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[:-4] for p in GOLD])
def test_c_restatement_reproduces_golden(path):
    z, arr, opt = _load(path)
    prob = {k: np.array(v, copy=True) for k, v in arr.items()}
    s = ba_cpu.solve(prob, threads=2, **opt)
    n_res = 2 * arr["obs_cam"].shape[0]
    assert (s["n_successful"], s["n_unsuccessful"]) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(math.sqrt(s["final_cost"] / n_res) - float(z["rmse_ref_style"])) < 1e-9
    assert np.abs(prob["cam_q"] - z["out_cam_q"]).max() < 1e-7 and np.abs(prob["cam_t"] - z["out_cam_t"]).max() < 1e-7


@pytest.mark.parametrize("mode,n_cams,n_pts,k_obs", [("sequential", 30, 1500, 4), ("unordered", 24, 600, 5)])
def test_c_restatement_matches_numpy_oracle(mode, n_cams, n_pts, k_obs):
    arr = H.make(n_cams, n_pts, k_obs, seed=111, mode=mode)
    pr = H.to_oracle(arr)
    s = bo.solve(pr, bo.Options())
    prob = {k: np.array(v, copy=True) for k, v in arr.items()}
    sc = ba_cpu.solve(prob, threads=4)
    assert (sc["n_successful"], sc["n_unsuccessful"]) == (s.n_successful, s.n_unsuccessful)
    assert abs(sc["final_cost"] - s.final_cost) <= 1e-9 * s.final_cost
    assert sc["num_effective_params"] == s.num_effective_params
    assert np.abs(prob["cam_q"] - pr.cam_q).max() < 1e-8 and np.abs(prob["cam_t"] - pr.cam_t).max() < 1e-8


def test_oracle_pcg_equals_exact():
    arr = H.make(10, 300, 4, seed=112)
    a = H.to_oracle(arr); b = H.to_oracle(arr)
    sa = bo.solve(a, bo.Options()); sb = bo.solve(b, bo.Options(linear_solver="pcg"))
    assert (sa.n_successful, sa.n_unsuccessful) == (sb.n_successful, sb.n_unsuccessful)
    assert np.abs(a.cam_q - b.cam_q).max() < 1e-8 and np.abs(a.cam_t - b.cam_t).max() < 1e-8


def test_lm_properties():
    """Size-independent properties: accepted costs decrease monotonically; a converged state re-solves in 0 steps."""
    arr = H.make(12, 500, 4, seed=113)
    pr = H.to_oracle(arr)
    s = bo.solve(pr, bo.Options())
    costs = [t["cost"] for t in s.trace if t.get("ok")]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    s2 = bo.solve(pr, bo.Options())
    assert s2.n_successful <= 1


def test_track_filter_oracle_reproduces_golden():
    z = np.load(os.path.join(GOLD_DIR, "track_filter.npz"))
    arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    ref = bo.filter_tracks(H.to_oracle(arr), float(z["max_re"]), float(z["min_angle"]))
    for k in ("obs_delete", "track_outlier", "num_filtered"):
        assert np.array_equal(ref[k], z["out_" + k]), k
    assert np.allclose(ref["track_error"], z["out_track_error"], rtol=0, atol=1e-12)
    assert np.allclose(ref["track_angle"], z["out_track_angle"], rtol=0, atol=1e-14)
    assert set(np.unique(z["out_track_outlier"])) == {0, 1, 2}


def test_tag_refine_oracle_and_product_reproduce_golden(lib):
    """tag_refine.npz: inputs + the scipy oracle's minima of both stages.  The oracle must reproduce them; the product
    (xrsfm_tag_refine, host code: runs without a GPU) reaches them with tight tolerances."""
    from oracle import tag_oracle as to
    from scipy.spatial.transform import Rotation
    from xrsfm_amd import capi
    z = np.load(os.path.join(GOLD_DIR, "tag_refine.npz"))
    tag_obs = (z["tag_obs_tag"], z["tag_obs_frame"], z["tag_obs_xy"])
    obs = (z["obs_frame"], z["obs_pt"], z["obs_xy"])
    q1, t1, s1, c1 = to.solve_stage1(z["corners"], float(z["tag_length"]))
    assert abs(s1 - float(z["stage1_scale"])) < 1e-9 and abs(c1 - float(z["stage1_cost"])) <= 1e-9 * c1
    tight = dict(function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-13)
    out = capi.tag_refine(z["frame_q"], z["frame_t"], z["corners"], *tag_obs, float(z["tag_length"]), points=z["points"],
                          obs_frame=obs[0], obs_pt=obs[1], obs_xy=obs[2], stages=2, **tight)
    s_1, s_2 = out["summaries"]
    assert abs(s_1.final_cost - float(z["stage1_cost"])) <= 1e-9 * s_1.final_cost
    assert abs(s_2.final_cost - float(z["stage2_cost"])) <= 1e-6 * s_2.final_cost
    assert abs(out["scale"] - float(z["stage2_scale"])) < 1e-5 * out["scale"]
    assert np.max((Rotation.from_quat(out["tag_q"]).inv() * Rotation.from_quat(z["stage2_q"])).magnitude()) < 1e-4
    assert np.abs(out["tag_corners"] - z["stage2_corners"]).max() < 1e-4 and np.abs(out["points"] - z["stage2_points"]).max() < 1e-4


def test_pose_graph_oracle_and_product_reproduce_golden(lib):
    """pose_graph.npz: a drifted loop with covisibility, loop and scale edges + the bounded least-squares minimum of the oracle.
    The oracle must reproduce it; the product (xrsfm_pg_solve, host code) ends at the same minimum within what its function
    tolerance (1e-6 per step, Ceres' default) allows, and reaches it with tight tolerances."""
    from oracle import pg_oracle as po
    from xrsfm_amd import capi
    z = np.load(os.path.join(GOLD_DIR, "pose_graph.npz"))
    edges = dict(a=z["edge_a"], b=z["edge_b"], sa=z["edge_sa"], sb=z["edge_sb"], q_mea=z["edge_q_mea"], p_mea=z["edge_p_mea"])
    sc_costs = [(int(a), int(b), float(c)) for a, b, c in z["scale_costs"]]
    kw = dict(weight_o=float(z["weight_o"]), scale_costs=sc_costs, pos_const=z["pos_const"], scale_const=z["scale_const"], scale_lower=z["scale_lower"])
    p_ref, s_ref, cost_ref, cost0 = po.solve(z["rot_q"], z["pos"], z["scale"], edges, **kw)
    assert abs(cost_ref - float(z["out_cost"])) <= 1e-9 * cost_ref and abs(cost0 - float(z["init_cost"])) <= 1e-12 * cost0
    assert np.abs(p_ref - z["out_pos"]).max() < 1e-6 and np.abs(s_ref - z["out_scale"]).max() < 1e-6
    pos, sc, s = capi.pose_graph_solve(z["rot_q"], z["pos"], z["scale"], edges, **kw)
    assert abs(s.initial_cost - float(z["init_cost"])) <= 1e-9 * s.initial_cost
    assert abs(s.final_cost - float(z["out_cost"])) <= 2e-3 * s.final_cost
    pos, sc, s = capi.pose_graph_solve(z["rot_q"], z["pos"], z["scale"], edges, function_tolerance=1e-14, parameter_tolerance=1e-13,
                                       gradient_tolerance=1e-12, max_iterations=500, **kw)
    assert abs(s.final_cost - float(z["out_cost"])) <= 1e-6 * s.final_cost
    assert np.abs(pos - z["out_pos"]).max() < 1e-3 and np.abs(sc - z["out_scale"]).max() < 1e-3


def test_oracle_against_real_ceres_runs():
    """If tests/golden/ceres_*.npz exist (bench/ceres_harness.cc run on a box with Ceres < 2.2, bench/make_ceres_golden.py), the
    numpy oracle must reproduce Ceres' step counts, final cost (1e-9) and cameras (1e-5, the north star's tolerances).  No such
    file is committed: Ceres / Eigen are absent from the build image and from every GPU box probed (__graft_entry__.py probe),
    so this test SKIPS and parity stays UNPINNED against the real reference."""
    if not CERES:
        pytest.skip("parity unpinned: no Ceres run available (bench/ceres_harness.cc needs Ceres < 2.2 + Eigen)")
    for path in CERES:
        z = np.load(path)
        arr = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
        pr = H.to_oracle(arr)
        s = bo.solve(pr, bo.Options())
        n_res = 2 * arr["obs_cam"].shape[0]
        assert (s.n_successful, s.n_unsuccessful) == (int(z["n_successful"]), int(z["n_unsuccessful"])), path
        assert abs(math.sqrt(s.final_cost / n_res) - math.sqrt(float(z["final_cost"]) / n_res)) < 1e-6
        assert np.abs(pr.cam_q - z["out_cam_q"]).max() < 1e-5 and np.abs(pr.cam_t - z["out_cam_t"]).max() < 1e-5


def test_numpy_oracle_reproduces_the_bal9_golden():
    """tests/golden/wide_bal9.npz (make_golden_bal9.py): 9-wide camera blocks, intrinsics {f, k1, k2} of model 5 variable."""
    z, arr, opt = _load(os.path.join(GOLD_DIR, "wide_bal9.npz"))
    pr = H.to_oracle(arr)
    cost0, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    assert Fc.shape[1:] == (2, 9) and abs(cost0 - float(z["init_cost"])) <= 1e-12 * cost0
    assert H.rel_err(rt, z["init_r"]) < 1e-12 and H.rel_err(Fc, z["init_Jc"]) < 1e-12 and H.rel_err(Ep, z["init_Jp"]) < 1e-12
    s = bo.solve(pr, bo.Options(**opt))
    assert (s.n_successful, s.n_unsuccessful) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(s.final_cost - float(z["final_cost"])) <= 1e-9 * s.final_cost
    assert np.abs(pr.cam_q - z["out_cam_q"]).max() < 1e-8 and np.abs(pr.intr_params - z["out_intr"]).max() < 1e-6


def test_c_restatement_reproduces_the_bal9_golden_and_the_numpy_oracle():
    """oracle/ba_cpu.c with 9-wide camera blocks (CW = 9 as soon as one camera keeps {f, k1, k2} variable): the committed golden,
    and the numpy oracle on a ragged problem with constant blocks — same LM decisions, cost to 1e-10, intrinsics included."""
    z, arr, opt = _load(os.path.join(GOLD_DIR, "wide_bal9.npz"))
    prob = {k: np.array(v, copy=True) for k, v in arr.items()}
    s = ba_cpu.solve(prob, threads=2, **opt)
    assert (s["n_successful"], s["n_unsuccessful"]) == (int(z["n_successful"]), int(z["n_unsuccessful"]))
    assert abs(s["final_cost"] - float(z["final_cost"])) <= 1e-10 * float(z["final_cost"])
    assert np.abs(prob["cam_q"] - z["out_cam_q"]).max() < 1e-8 and np.abs(prob["intr_params"] - z["out_intr"]).max() < 1e-6
    arr = H.make_bal9(40, 2000, 6, seed=6, dropout=0.3, min_tri_angle_deg=0.5)
    arr["point_const"] = (np.arange(2000) % 7 == 0).astype(np.uint8)
    cc = arr["cam_const"].copy(); cc[5] &= 3; cc[9] &= 3; cc[11] |= 1; arr["cam_const"] = cc
    pr = H.to_oracle(arr)
    s_ref = bo.solve(pr, bo.Options(max_iterations=12))
    prob = {k: np.array(v, copy=True) for k, v in arr.items()}
    s = ba_cpu.solve(prob, max_iterations=12, threads=1)
    assert (s["n_successful"], s["n_unsuccessful"]) == (s_ref.n_successful, s_ref.n_unsuccessful)
    assert s["num_effective_params"] == s_ref.num_effective_params
    assert abs(s["final_cost"] - s_ref.final_cost) <= 1e-10 * s_ref.final_cost
    assert np.abs(prob["cam_q"] - pr.cam_q).max() < 1e-8 and np.abs(prob["cam_t"] - pr.cam_t).max() < 1e-7
    assert np.abs(prob["intr_params"] - pr.intr_params).max() < 1e-6
    # shared intrinsics entry / one of the reference's models with the bit: refused
    bad = {k: np.array(v, copy=True) for k, v in arr.items()}; bad["cam_intr"] = np.zeros(40, np.int32)
    with pytest.raises(RuntimeError):
        ba_cpu.solve(bad, threads=1)
