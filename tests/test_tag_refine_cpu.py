"""Tag refinement (host code): xrsfm_tag_refine = the two ceres::Solve calls of tag_refine
(/root/reference/src/tag/tag_extract.hpp:193-265) against the independent scipy restatement in oracle/tag_oracle.py."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import tag_oracle as to


@pytest.fixture(scope="module")
def capi(lib):
    from xrsfm_amd import capi
    return capi


def make_scene(n_frames=14, n_tags=3, n_points=300, s_true=2.7, tag_length=0.113, seed=0, obs_noise=2e-4, corner_noise=2e-3,
               point_noise=2e-2):
    """A ring of cameras looking inwards at tags and points; the map is s_true map units per metre."""
    rng = np.random.default_rng(seed)
    ang = np.linspace(0, 2 * np.pi, n_frames, endpoint=False)
    centres = np.stack([4 * np.cos(ang), 0.3 * rng.normal(size=n_frames), 4 * np.sin(ang)], 1) * s_true
    frame_q, frame_t = [], []
    for c in centres:
        zc = -c / np.linalg.norm(c)
        xc = np.cross([0, 1.0, 0], zc); xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        Rcw = np.stack([xc, yc, zc])                       # rows: camera axes in the world
        Rcw = Rotation.from_rotvec(0.05 * rng.normal(size=3)).as_matrix() @ Rcw
        frame_q.append(Rotation.from_matrix(Rcw).as_quat()); frame_t.append(-Rcw @ c)
    frame_q, frame_t = np.array(frame_q), np.array(frame_t)
    frame_R = Rotation.from_quat(frame_q).as_matrix()
    tag_R = Rotation.from_rotvec(rng.normal(0, 1.2, (n_tags, 3)))
    tag_t = rng.uniform(-0.8, 0.8, (n_tags, 3)) * s_true
    corners = s_true * np.einsum("kij,cj->kci", tag_R.as_matrix(), to.tag_points(tag_length)) + tag_t[:, None, :]
    ok, of, oxy = [], [], []
    for k in range(n_tags):
        for f in rng.choice(n_frames, size=rng.integers(4, n_frames), replace=False):
            pc = np.einsum("ij,cj->ci", frame_R[f], corners[k]) + frame_t[f]
            ok.append(k); of.append(f); oxy.append(pc[:, :2] / pc[:, 2:] + obs_noise * rng.normal(size=(4, 2)))
    points = rng.uniform(-1.0, 1.0, (n_points, 3)) * s_true
    pf, pp, pxy = [], [], []
    for j in range(n_points):
        for f in rng.choice(n_frames, size=rng.integers(2, 7), replace=False):
            pc = frame_R[f] @ points[j] + frame_t[f]
            pf.append(f); pp.append(j); pxy.append(pc[:2] / pc[2] + obs_noise * rng.normal(size=2))
    return dict(frame_q=frame_q, frame_t=frame_t, tag_length=tag_length, s_true=s_true,
                tag_q_true=tag_R.as_quat(), tag_t_true=tag_t, corners_true=corners,
                corners=corners + corner_noise * rng.normal(size=corners.shape),
                tag_obs=(np.array(ok, np.int32), np.array(of, np.int32), np.array(oxy)),
                points_true=points, points=points + point_noise * rng.normal(size=points.shape),
                obs=(np.array(pf, np.int32), np.array(pp, np.int32), np.array(pxy)))


def _rot_dist(qa, qb):
    return np.max((Rotation.from_quat(qa).inv() * Rotation.from_quat(qb)).magnitude())


def test_stage1_recovers_scale_and_tag_poses(capi):
    sc = make_scene(seed=1)
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1)
    s1 = out["summaries"][0]
    assert s1.termination in (1, 2, 3) and s1.final_cost < 1e-3 * s1.initial_cost
    q, t, s, c = to.solve_stage1(sc["corners"], sc["tag_length"])
    # the solver stops on Ceres' function tolerance (1e-6 relative cost change): close to, not at, the minimum
    assert abs(s1.final_cost - c) <= 1e-4 * c + 1e-12
    assert abs(out["scale"] - s) < 2e-3 * s and _rot_dist(out["tag_q"], q) < 5e-3 and np.abs(out["tag_t"] - t).max() < 2e-3
    assert abs(out["scale"] - sc["s_true"]) < 0.1 * sc["s_true"]
    assert np.array_equal(out["tag_corners"].reshape(-1), np.asarray(sc["corners"]).reshape(-1))        # constant in stage 1
    # reported cost = cost of the returned state
    assert abs(to.cost(sc["frame_q"], sc["frame_t"], out["tag_q"], out["tag_t"], out["scale"], sc["corners"], sc["tag_length"], 1)
               - s1.final_cost) <= 1e-12 * max(1.0, s1.final_cost)


def test_stage1_tight_tolerances_reach_the_oracle_minimum(capi):
    sc = make_scene(seed=2, n_tags=5)
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1,
                          function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-13)
    q, t, s, c = to.solve_stage1(sc["corners"], sc["tag_length"])
    assert abs(out["summaries"][0].final_cost - c) <= 1e-9 * c
    assert abs(out["scale"] - s) < 1e-7 * s and _rot_dist(out["tag_q"], q) < 1e-6 and np.abs(out["tag_t"] - t).max() < 1e-6


def test_scale_lower_bound_is_enforced(capi):
    sc = make_scene(seed=3, s_true=0.05)            # the unconstrained optimum is far below the bound
    q, t, s, c = to.solve_stage1(sc["corners"], sc["tag_length"])
    assert abs(s - 0.2) < 1e-9
    # Default = Ceres' handling of the bound (tag_extract.hpp:227 -> SetParameterLowerBound; projected Plus + projected Armijo
    # search only): feasible, on the bound, large cost decrease; its loop may stop up to 2x above the constrained minimum (the
    # model keeps promising the infeasible decrease), as upstream's does.
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1)
    s1 = out["summaries"][0]
    assert out["scale"] >= 0.2 and abs(out["scale"] - 0.2) < 1e-9
    assert s1.final_cost < 0.1 * s1.initial_cost and c * (1 - 1e-9) <= s1.final_cost <= 2.0 * c
    # Opt-in active set (bounds_active_set=1, a documented deviation): the scale is held while it sits on the bound and the
    # gradient pushes it down, the tags take the step of the problem restricted to scale = 0.2, and the solve reaches the
    # constrained minimum of the independent bounded least-squares solver (scipy) — within the function tolerance by default,
    # to round-off with tight tolerances.
    act = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1, bounds_active_set=1)
    a1 = act["summaries"][0]
    assert abs(act["scale"] - 0.2) < 1e-9 and c * (1 - 1e-9) <= a1.final_cost <= c * (1 + 1e-4)
    tight = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1,
                            function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-13, bounds_active_set=1)
    assert tight["scale"] == 0.2 and abs(tight["summaries"][0].final_cost - c) <= 1e-9 * c
    assert _rot_dist(tight["tag_q"], q) < 1e-6 and np.abs(tight["tag_t"] - t).max() < 1e-6


def test_both_stages_match_the_oracle(capi):
    sc = make_scene(seed=4)
    tight = dict(function_tolerance=1e-15, parameter_tolerance=1e-14, gradient_tolerance=1e-13)
    one = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], stages=1, **tight)
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], points=sc["points"],
                          obs_frame=sc["obs"][0], obs_pt=sc["obs"][1], obs_xy=sc["obs"][2], stages=2, **tight)
    s1, s2 = out["summaries"]
    assert s1.final_cost == one["summaries"][0].final_cost
    q, t, s, c, pts = to.solve_stage2(sc["frame_q"], sc["frame_t"], one["tag_q"], one["tag_t"], one["scale"], sc["corners"],
                                      sc["tag_length"], sc["tag_obs"], sc["points"], sc["obs"])
    ref_cost = to.cost(sc["frame_q"], sc["frame_t"], q, t, s, c, sc["tag_length"], 2, sc["tag_obs"], pts, sc["obs"])
    got_cost = to.cost(sc["frame_q"], sc["frame_t"], out["tag_q"], out["tag_t"], out["scale"], out["tag_corners"], sc["tag_length"], 2,
                       sc["tag_obs"], out["points"], sc["obs"])
    assert abs(got_cost - s2.final_cost) <= 1e-12 * s2.final_cost
    assert abs(s2.final_cost - ref_cost) <= 1e-6 * ref_cost
    assert abs(out["scale"] - s) < 1e-5 * s and _rot_dist(out["tag_q"], q) < 1e-4
    assert np.abs(out["tag_corners"] - c).max() < 1e-4 and np.abs(out["points"] - pts).max() < 1e-4
    # the refinement pulls corners and points towards the truth and the scale to the metric one
    assert np.abs(out["tag_corners"] - sc["corners_true"]).max() < np.abs(sc["corners"] - sc["corners_true"]).max()
    assert np.abs(out["points"] - sc["points_true"]).mean() < 0.2 * np.abs(sc["points"] - sc["points_true"]).mean()
    assert abs(out["scale"] - sc["s_true"]) < 0.05 * sc["s_true"]


def test_default_options_are_the_reference_settings(capi):
    import ctypes as C
    o = capi.CPgOptions()
    capi.load().xrsfm_tag_default_options(C.byref(o))
    assert (o.max_iterations, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance, o.initial_radius) == (500, 1e-6, 1e-8, 1e-10, 1e4)


def test_outlier_observations_are_down_weighted_and_inputs_validated(capi):
    sc = make_scene(seed=5)
    oxy = sc["obs"][2].copy()
    oxy[::37] += 0.05                                  # ~35 px at f = 700: beyond sigma, the sqrt(sigma/|r|) branch
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], points=sc["points"],
                          obs_frame=sc["obs"][0], obs_pt=sc["obs"][1], obs_xy=oxy, stages=2)
    s2 = out["summaries"][1]
    assert s2.termination in (1, 2, 3, 4) and s2.final_cost < s2.initial_cost
    got = to.cost(sc["frame_q"], sc["frame_t"], out["tag_q"], out["tag_t"], out["scale"], out["tag_corners"], sc["tag_length"], 2,
                  sc["tag_obs"], out["points"], (sc["obs"][0], sc["obs"][1], oxy))
    assert abs(got - s2.final_cost) <= 1e-12 * s2.final_cost
    assert np.median(np.abs(out["points"] - sc["points_true"])) < np.median(np.abs(sc["points"] - sc["points_true"]))
    bad = sc["tag_obs"][0].copy(); bad[0] = 99
    with pytest.raises(RuntimeError):
        capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], bad, sc["tag_obs"][1], sc["tag_obs"][2], sc["tag_length"], stages=1)
    with pytest.raises(RuntimeError):
        capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], 0.0, stages=1)


def test_compat_refine_map_with_tags_equals_c_abi(capi, tmp_path):
    """RefineMapWithTags (compat/tag/tag_refine_solve.h; replaces tag_extract.hpp:193-275) on a shim Map keyed by sparse ids
    = xrsfm_tag_refine on the flat arrays in the same visiting order, followed by the division by the scale; the
    unregistered frame and the outlier track are ignored by the solve but rescaled with the rest of the map."""
    import os, struct, subprocess
    shim = os.path.join(os.path.dirname(__file__), "shim")
    subprocess.run(["make", "-C", shim, "_build/tag_main"], check=True, capture_output=True)
    sc = make_scene(seed=6, n_points=120)
    tk, tf, txy = sc["tag_obs"]
    of, op, oxy = sc["obs"]
    inp, outp = str(tmp_path / "tag_in.bin"), str(tmp_path / "tag_out.bin")
    nf, nt, npt = sc["frame_q"].shape[0], sc["corners"].shape[0], sc["points"].shape[0]
    with open(inp, "wb") as f:
        f.write(struct.pack("5i", nf, nt, len(tk), npt, len(of)))
        f.write(struct.pack("d", sc["tag_length"]))
        f.write(np.hstack([sc["frame_q"], sc["frame_t"]]).astype("f8").tobytes())
        f.write(sc["corners"].astype("f8").tobytes())
        f.write(np.stack([tk, tf], 1).astype("i4").tobytes()); f.write(txy.astype("f8").tobytes())
        f.write(sc["points"].astype("f8").tobytes())
        f.write(np.stack([of, op], 1).astype("i4").tobytes()); f.write(oxy.astype("f8").tobytes())
    p = subprocess.run([os.path.join(shim, "_build", "tag_main"), inp, outp], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    raw = np.frombuffer(open(outp, "rb").read(), dtype="f8")
    scale = raw[0]; o = 1
    ft = raw[o:o + 3 * nf].reshape(nf, 3); o += 3 * nf
    pts = raw[o:o + 3 * npt].reshape(npt, 3); o += 3 * npt
    corners = raw[o:o + 12 * nt].reshape(nt, 4, 3); o += 12 * nt
    tq = raw[o:o + 4 * nt].reshape(nt, 4); o += 4 * nt
    tt = raw[o:o + 3 * nt].reshape(nt, 3); o += 3 * nt
    extra = raw[o:o + 2]
    # the same problem in the adapter's visiting order: one extra (unregistered, unused) frame first, observations by frame,
    # tracks by first appearance, tag observations by (tag, frame)
    fq = np.vstack([[0, 0, 0, 1.0], sc["frame_q"]]); ftin = np.vstack([[8.0, 0, 0], sc["frame_t"]])
    order = np.argsort(of, kind="stable")
    of2, op2, oxy2 = of[order] + 1, op[order], oxy[order]
    first = {}
    for j in op2:
        first.setdefault(int(j), len(first))
    perm = np.array(sorted(first, key=first.get))             # slot -> original track
    slot = np.empty(npt, np.int64); slot[perm] = np.arange(len(perm))
    torder = np.lexsort((tf, tk))
    ref = capi.tag_refine(fq, ftin, sc["corners"], tk[torder], tf[torder] + 1, txy[torder], sc["tag_length"], points=sc["points"][perm],
                          obs_frame=of2, obs_pt=slot[op2], obs_xy=oxy2, stages=2)
    assert scale == ref["scale"] and abs(scale - sc["s_true"]) < 0.05 * sc["s_true"]
    assert np.array_equal(corners, ref["tag_corners"]) and np.array_equal(tq, ref["tag_q"]) and np.array_equal(tt, ref["tag_t"])
    assert np.array_equal(pts[perm], ref["points"] / scale) and len(perm) == npt
    assert np.array_equal(ft, sc["frame_t"] / scale)
    assert extra[0] == 8.0 / scale and extra[1] == 3.0 / scale
    assert p.stdout.count("xrsfm_ba Report:") == 2 and "tag refine stage 2" in p.stdout


def test_degenerate_inputs_do_not_crash(capi):
    sc = make_scene(seed=21, n_tags=1, n_points=10)
    # no tags: stage 1 has nothing to do, stage 2 refines the track points, the scale stays 1
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], np.zeros((0, 4, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 4, 2)), 0.1,
                          points=sc["points"], obs_frame=sc["obs"][0], obs_pt=sc["obs"][1], obs_xy=sc["obs"][2], stages=2)
    assert out["scale"] == 1.0 and out["summaries"][0].iterations == 0 and out["summaries"][1].final_cost < out["summaries"][1].initial_cost
    # a point nobody observes keeps its value
    pts = np.vstack([sc["points"], [[100.0, 0, 0]]])
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], sc["corners"], *sc["tag_obs"], sc["tag_length"], points=pts,
                          obs_frame=sc["obs"][0], obs_pt=sc["obs"][1], obs_xy=sc["obs"][2], stages=2)
    assert np.array_equal(out["points"][-1], [100.0, 0, 0])
    # non-finite input: reported as a failure (termination 6), nothing changes
    bad = sc["corners"].copy(); bad[0, 0, 0] = np.nan
    out = capi.tag_refine(sc["frame_q"], sc["frame_t"], bad, *sc["tag_obs"], sc["tag_length"], stages=1)
    assert out["summaries"][0].termination == 6 and out["scale"] == 1.0
