"""Point (track) partition of one BA problem over the ranks of a node ("length-aware greedy by
sum of track length, so that observations per GPU balance within 1 %").

Every residual touches one camera and one point, so all observations of a point stay on one rank and the cameras are
replicated; the streaming kernels of a rank cost time in proportion to its OBSERVATIONS, not its points.  `j % world` balances
points; with power-law track lengths (unordered photo collections, BASELINE.json config 5) or an input sorted by track length
it does not balance observations.  Host-side caller code (the library takes one shard per context): bench.py and the tests.
"""
from __future__ import annotations

import numpy as np


def partition_points(obs_pt: np.ndarray, n_points: int, world: int) -> np.ndarray:
    """owner[j] in [0, world) for every point j.  Longest-processing-time greedy by track length, evaluated class by class:
    the points of one length L (ascending index inside a class) go round-robin to the ranks in ascending order of their
    current load, so that after every class the loads differ by at most L; classes in descending L, so the final spread is
    bounded by the shortest tracks.  Deterministic (stable sorts, ties by rank id); O(N log N)."""
    owner = np.zeros(n_points, np.int32)
    if world <= 1 or n_points == 0:
        return owner
    length = np.bincount(np.asarray(obs_pt, np.int64), minlength=n_points)
    order = np.argsort(-length, kind="stable")                 # descending length, ascending index inside a class
    ls = length[order]
    bounds = np.flatnonzero(np.diff(ls)) + 1
    starts = np.concatenate([[0], bounds]); ends = np.concatenate([bounds, [n_points]])
    load = np.zeros(world, np.int64)
    for a, b in zip(starts, ends):
        L = int(ls[a])
        m = b - a
        ranks = np.argsort(load, kind="stable")                # least loaded first
        # full rounds keep the loads' differences; the remainder goes to the least loaded ranks
        who = ranks[np.arange(m) % world]
        owner[order[a:b]] = who
        if L > 0:
            load += (m // world) * L
            load[ranks[:m % world]] += L
    return owner


def shard_problem(arr: dict, rank: int, world: int, balance: str = "length") -> dict:
    """The shard of rank `rank`: its points, their observations (point indices renumbered), all cameras.
    balance = "length": partition_points (default); "modulo": point j on rank j % world (the round-1..3 rule)."""
    if world == 1:
        return arr
    n_points = arr["points"].shape[0]
    if balance == "modulo":
        owner = (np.arange(n_points) % world).astype(np.int32)
    else:
        owner = partition_points(arr["obs_pt"], n_points, world)
    keep_pt = owner == rank
    new_idx = np.cumsum(keep_pt) - 1
    keep_obs = keep_pt[arr["obs_pt"]]
    out = dict(arr)
    out["points"] = np.ascontiguousarray(arr["points"][keep_pt])
    out["point_const"] = np.ascontiguousarray(arr["point_const"][keep_pt])
    out["obs_cam"] = np.ascontiguousarray(arr["obs_cam"][keep_obs])
    out["obs_pt"] = np.ascontiguousarray(new_idx[arr["obs_pt"][keep_obs]].astype(np.int32))
    out["obs_uv"] = np.ascontiguousarray(arr["obs_uv"][keep_obs])
    return out


def imbalance(obs_pt: np.ndarray, owner: np.ndarray, world: int) -> float:
    """max over ranks of (observations of the rank) / mean - 1."""
    length = np.bincount(np.asarray(obs_pt, np.int64), minlength=owner.shape[0])
    loads = np.bincount(owner, weights=length, minlength=world)
    return float(loads.max() / loads.mean() - 1.0)
