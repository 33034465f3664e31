#!/usr/bin/env python3
"""Golden fixtures of the "next" rows: post-BA track filter (f1), scaled pose graph and tag refinement (f4).

Run from the repo root:  python tests/golden/make_golden_extra.py
Inputs come from the seeded generators of the tests, expected outputs from the oracles (oracle/ba_oracle.py
filter_tracks; oracle/tag_oracle.py with tight tolerances).  Like make_golden.py these vectors pin OUR restatements —
"parity unpinned" against the real reference, whose arithmetic (Ceres) cannot run here.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_oracle as bo  # noqa: E402
from oracle import pg_oracle as po  # noqa: E402
from oracle import tag_oracle as to  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_pose_graph_cpu import _loop_problem  # noqa: E402
from tests.test_tag_refine_cpu import make_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def track_filter():
    # a nearly converged state: what the filter sees after a global BA (gross outliers, a few points behind the cameras or
    # far away, low-parallax tracks)
    arr = H.make(10, 400, 4, seed=151, outlier_frac=0.08, min_tri_angle_deg=0.2, perturb=(0.0005, 0.002, 0.005))
    arr["points"][::40] += np.array([0.0, 0.0, -70.0])
    arr["points"][5::45] *= 40.0
    max_re, min_angle = 4.0, math.radians(1.5)
    ref = bo.filter_tracks(H.to_oracle(arr), max_re, min_angle)
    np.savez_compressed(os.path.join(OUT, "track_filter.npz"), **{"in_" + k: v for k, v in arr.items()},
                        max_re=max_re, min_angle=min_angle, **{"out_" + k: v for k, v in ref.items()})
    print("track_filter", ref["num_filtered"], np.bincount(ref["track_outlier"]))


def tag_refine():
    sc = make_scene(seed=11, n_frames=10, n_tags=2, n_points=60)
    q1, t1, s1, c1 = to.solve_stage1(sc["corners"], sc["tag_length"])
    q2, t2, s2, corners2, pts2 = to.solve_stage2(sc["frame_q"], sc["frame_t"], q1, t1, s1, sc["corners"], sc["tag_length"], sc["tag_obs"],
                                                 sc["points"], sc["obs"])
    c2 = to.cost(sc["frame_q"], sc["frame_t"], q2, t2, s2, corners2, sc["tag_length"], 2, sc["tag_obs"], pts2, sc["obs"])
    np.savez_compressed(
        os.path.join(OUT, "tag_refine.npz"), frame_q=sc["frame_q"], frame_t=sc["frame_t"], tag_length=sc["tag_length"],
        corners=sc["corners"], tag_obs_tag=sc["tag_obs"][0], tag_obs_frame=sc["tag_obs"][1], tag_obs_xy=sc["tag_obs"][2],
        points=sc["points"], obs_frame=sc["obs"][0], obs_pt=sc["obs"][1], obs_xy=sc["obs"][2],
        stage1_q=q1, stage1_t=t1, stage1_scale=s1, stage1_cost=c1,
        stage2_q=q2, stage2_t=t2, stage2_scale=s2, stage2_corners=corners2, stage2_points=pts2, stage2_cost=c2)
    print("tag_refine", s1, c1, s2, c2)


def pose_graph():
    prob, _ = _loop_problem(n=30, seed=3, scale_obs=1.03)
    pos, sc, cost, cost0 = po.solve(prob["rot_q"], prob["pos"], prob["scale"], prob["edges"], prob["weight_o"], prob["scale_costs"],
                                    prob["pos_const"], prob["scale_const"], prob["scale_lower"])
    e = prob["edges"]
    np.savez_compressed(
        os.path.join(OUT, "pose_graph.npz"), rot_q=prob["rot_q"], pos=prob["pos"], scale=prob["scale"], weight_o=prob["weight_o"],
        edge_a=e["a"], edge_b=e["b"], edge_sa=e["sa"], edge_sb=e["sb"], edge_q_mea=e["q_mea"], edge_p_mea=e["p_mea"],
        scale_costs=np.array(prob["scale_costs"], float), pos_const=prob["pos_const"], scale_const=prob["scale_const"],
        scale_lower=prob["scale_lower"], out_pos=pos, out_scale=sc, out_cost=cost, init_cost=cost0)
    print("pose_graph", cost0, cost)


if __name__ == "__main__":
    track_filter()
    tag_refine()
    pose_graph()
