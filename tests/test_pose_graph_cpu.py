"""the scaled pose graph of ScalePoseGraphUnorder behind xrsfm_pg_solve (host code: runs without a GPU).
Checked against oracle/pg_oracle.py (numpy residuals + scipy bounded least squares) — parity unpinned against real Ceres."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import pg_oracle as po
from xrsfm_amd import capi


def _loop_problem(n=40, seed=0, drift=0.08, n_loop=2, weight_o=0.5, scale_obs=None):
    """A circular trajectory whose estimate has accumulated scale drift; covisibility edges are built from the CURRENT
    estimate (zero residual at the start, ba_solver.cc:100-104), loop edges from the corrected pose of the last frame
    (AddLoopEdge, :117-145), so the optimiser has to spread the correction over the per-frame scales and positions."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 1.9 * np.pi, n)
    true_pos = np.stack([10 * np.cos(ang), 0.2 * rng.normal(size=n), 10 * np.sin(ang)], 1)
    rot = Rotation.from_euler("y", -ang) * Rotation.from_rotvec(0.02 * rng.normal(size=(n, 3)))
    rot_q = rot.as_quat()
    # estimate with drift: steps shrink progressively
    est = np.zeros_like(true_pos); est[0] = true_pos[0]
    for i in range(1, n):
        est[i] = est[i - 1] + (true_pos[i] - true_pos[i - 1]) * (1.0 - drift * i / n)
    R = rot.as_matrix()
    a, b, sa, sb, qm, pm = [], [], [], [], [], []
    for i in range(n):
        for j in range(max(0, i - 3), i):          # frame.id > cor_id (:96-97)
            a.append(i); b.append(j); sa.append(i); sb.append(j)
            qm.append((rot[i].inv() * rot[j]).as_quat()); pm.append(R[i].T @ (est[j] - est[i]))
    # loop: last frame re-localised against the first frames: its corrected pose sits at the true position
    loop_frame = n - 1
    n_scales = n + n_loop
    for k in range(n_loop):
        mea_pos = true_pos[loop_frame] + 0.01 * rng.normal(size=3)
        for j in range(3 * k, 3 * k + 3):
            a.append(loop_frame); b.append(j); sa.append(n + k); sb.append(j)
            qm.append((rot[loop_frame].inv() * rot[j]).as_quat()); pm.append(R[loop_frame].T @ (est[j] - mea_pos))
    edges = dict(a=np.array(a), b=np.array(b), sa=np.array(sa), sb=np.array(sb), q_mea=np.array(qm), p_mea=np.array(pm))
    pos_const = np.zeros(n, np.uint8); pos_const[[0, 1]] = 1              # init_id1/2 (:254-257)
    scale_const = np.zeros(n_scales, np.uint8); scale_const[[0, 1]] = 1
    lower = np.full(n_scales, 0.2); lower[loop_frame] = -np.inf           # (:245-247)
    sc = [(n, n + 1, scale_obs)] if scale_obs is not None else []
    return dict(rot_q=rot_q, pos=est, scale=np.ones(n_scales), edges=edges, weight_o=weight_o, scale_costs=sc,
                pos_const=pos_const, scale_const=scale_const, scale_lower=lower), true_pos


def test_residual_rows_match_oracle_at_start():
    """Covisibility edges are consistent with the estimate they were built from: only the loop edges carry error."""
    prob, _ = _loop_problem()
    r = po.residuals(prob["rot_q"], prob["pos"], prob["scale"], prob["edges"], prob["weight_o"], prob["scale_costs"])
    n_cov = len(prob["edges"]["a"]) - 6
    assert np.abs(r[:8 * n_cov]).max() < 1e-9
    assert np.abs(r[8 * n_cov:]).max() > 0.1
    pos, sc, s = capi.pose_graph_solve(**prob, max_iterations=0)
    assert abs(s.initial_cost - 0.5 * np.sum(r ** 2)) <= 1e-12 * s.initial_cost and s.final_cost == s.initial_cost
    assert np.array_equal(pos, prob["pos"])


@pytest.mark.parametrize("scale_obs", [None, 1.03])
def test_minimum_matches_bounded_least_squares(scale_obs):
    prob, true_pos = _loop_problem(seed=1, scale_obs=scale_obs)
    p_ref, s_ref, cost_ref, cost0 = po.solve(**prob)
    pos, sc, s = capi.pose_graph_solve(**prob)
    assert s.termination in (1, 2, 3) and s.n_successful >= 1
    assert abs(s.initial_cost - cost0) <= 1e-9 * cost0
    assert s.final_cost < 0.2 * s.initial_cost
    assert s.final_cost <= cost_ref * (1 + 1e-4) + 1e-12            # at least as good as the independent solver ...
    assert abs(s.final_cost - cost_ref) <= 2e-3 * cost_ref + 1e-10  # ... and the same minimum (function tolerance 1e-6 per step)
    assert np.abs(pos - p_ref).max() < 5e-2 and np.abs(sc - s_ref).max() < 1e-2
    # the loop is closed better than before
    assert np.linalg.norm(pos[-1] - true_pos[-1]) < 0.5 * np.linalg.norm(prob["pos"][-1] - true_pos[-1])
    # gauge and constants respected
    assert np.array_equal(pos[:2], prob["pos"][:2]) and np.array_equal(sc[:2], [1.0, 1.0])


def test_lower_bound_is_respected():
    prob, _ = _loop_problem(seed=2, drift=0.9, weight_o=0.0)       # heavy drift pushes some scales towards the bound
    prob["scale_lower"] = np.where(np.isfinite(prob["scale_lower"]), 0.8, -np.inf)
    p_ref, s_ref, cost_ref, _ = po.solve(**prob)
    pos, sc, s = capi.pose_graph_solve(**prob)
    free = np.isfinite(prob["scale_lower"])
    assert (sc[free] >= 0.8 - 1e-12).all()
    assert s.final_cost < s.initial_cost
    assert s.final_cost <= cost_ref * 1.05 + 1e-9


def test_active_lower_bounds_reach_the_constrained_minimum():
    """Scales that want to shrink below their bound (ba_solver.cc:245-252 bounds them at 0.2; here 0.97 so that several are
    active).  Default = Ceres' handling of bounds (projection + projected line search only): feasible, decreases the cost and
    stays within 2x of the constrained minimum — upstream's loop has no active set and may stop above it.  With
    bounds_active_set=1 (opt-in deviation) the held scales let the remaining variables reach the constrained minimum of
    scipy's bounded least squares."""
    prob, _ = _loop_problem(seed=2, drift=0.9, weight_o=0.0)
    prob["scale_lower"] = np.where(np.isfinite(prob["scale_lower"]), 0.97, -np.inf)
    p_ref, s_ref, cost_ref, _ = po.solve(**prob)
    free = np.isfinite(prob["scale_lower"])
    n_active_ref = int((s_ref[free] <= 0.97 + 1e-9).sum())
    assert n_active_ref >= 2                                         # the case really has active bounds
    # Ceres-faithful default
    pos0, sc0, s0 = capi.pose_graph_solve(**prob)
    assert (sc0[free] >= 0.97 - 1e-12).all()
    assert s0.final_cost < s0.initial_cost and cost_ref * (1 - 1e-9) <= s0.final_cost <= 2.0 * cost_ref
    # opt-in active set
    pos, sc, s = capi.pose_graph_solve(**prob, function_tolerance=1e-13, parameter_tolerance=1e-13, bounds_active_set=1)
    assert (sc[free] >= 0.97 - 1e-12).all()
    assert abs(s.final_cost - cost_ref) <= 1e-6 * cost_ref
    assert int((sc[free] <= 0.97 + 1e-9).sum()) == n_active_ref and np.abs(sc - s_ref).max() < 1e-4 and np.abs(pos - p_ref).max() < 1e-3
    assert s.final_cost <= s0.final_cost * (1 + 1e-12)


def test_degenerate_inputs():
    prob, _ = _loop_problem(n=6, n_loop=1)
    e = {k: v[:0] for k, v in prob["edges"].items()}
    pos, sc, s = capi.pose_graph_solve(prob["rot_q"], prob["pos"], prob["scale"], e)
    assert s.initial_cost == 0.0 and s.iterations == 0 and np.array_equal(pos, prob["pos"])
    bad = dict(prob["edges"]); bad["a"] = bad["a"].copy(); bad["a"][0] = 99
    with pytest.raises(Exception):
        capi.pose_graph_solve(prob["rot_q"], prob["pos"], prob["scale"], bad)
