"""Gauge- and conditioning-aware comparison of two bundle-adjustment results (test / measurement helper; the solver itself
lives in csrc/).

XRSfM's BA fixes only the translations of the two initial frames (/root/reference/src/optimization/ba_solver.cc:611-614)
and stops at a relative cost change of 1e-5 (:628).  On a 1000-frame loop with 4-frame tracks the global shape of the
trajectory (scale about the fixed pair, low-frequency bending of the loop) is then only weakly determined: two exact FP64
solvers — or ONE solver with its residual blocks added in a different order — agree in the cost to 1e-10 and end up to
1e-3 apart in the absolute translations of the far side of the loop, with the points moving along (the Gauss-Newton energy
of the whole difference is 1e-12 of the cost).  What IS determined to the north star's 1e-5 is local: the relative pose of
cameras that share tracks.  This module measures both:

  tangent_difference          per-camera difference in the tangent space of the parameterisation
  relative_pose_difference    difference of T_i T_j^-1 over covisible camera pairs (gauge invariant, drift free)
  covisible_pairs             the camera pairs that share a track
  camera_difference_spectrum  (diagnostic, tools/parity_spectrum.py) the difference in the eigenbasis of the reduced matrix"""
from __future__ import annotations

import numpy as np


def tangent_difference(q_a, t_a, q_b, t_b):
    """d in R^{6 Nc}: per camera the rotation vector w with q_a = Plus(q_b, w) (EigenQuaternionParameterization: full angle,
    left multiplication) and t_a - t_b."""
    qa = np.asarray(q_a, float); qb = np.asarray(q_b, float)
    # dq = q_a * conj(q_b)   (x,y,z,w)
    ax, ay, az, aw = qa.T; bx, by, bz, bw = (-qb[:, 0], -qb[:, 1], -qb[:, 2], qb[:, 3])
    dx = aw * bx + bw * ax + (ay * bz - az * by)
    dy = aw * by + bw * ay + (az * bx - ax * bz)
    dz = aw * bz + bw * az + (ax * by - ay * bx)
    dw = aw * bw - (ax * bx + ay * by + az * bz)
    v = np.stack([dx, dy, dz], 1)
    n = np.linalg.norm(v, axis=1)
    ang = np.arctan2(n, dw)                      # Plus uses the full angle: dq = (sin|w| w/|w|, cos|w|)
    w = np.where(n[:, None] > 0, v / np.maximum(n, 1e-300)[:, None] * ang[:, None], v)
    return np.concatenate([w, np.asarray(t_a, float) - np.asarray(t_b, float)], 1).reshape(-1)


def _rot(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)


def covisible_pairs(obs_cam, obs_pt, max_track=64):
    """Unique (i, j), i < j, of cameras that observe a common track."""
    order = np.lexsort((obs_cam, obs_pt))
    cam = np.asarray(obs_cam)[order].astype(np.int64); pt = np.asarray(obs_pt)[order]
    keys = []
    for d in range(1, max_track):
        same = pt[d:] == pt[:-d]
        if not same.any():
            break
        keys.append(cam[:-d][same] * (1 << 32) + cam[d:][same])
    if not keys:
        return np.zeros((0, 2), np.int64)
    k = np.unique(np.concatenate(keys))
    return np.stack([k >> 32, k & 0xffffffff], 1)


def relative_pose_difference(q_a, t_a, q_b, t_b, pairs):
    """For every pair (i, j): T_ij = T_i T_j^-1 (the pose of camera i relative to camera j; R_ij = R_i R_j^T,
    t_ij = t_i - R_ij t_j) in result a and in result b.  Returns (max rotation angle between the two R_ij [rad],
    max |t_ij(a) - t_ij(b)|): invariant under a change of the world frame and insensitive to drift accumulated far away."""
    pairs = np.asarray(pairs)
    if pairs.shape[0] == 0:
        return 0.0, 0.0
    i, j = pairs[:, 0], pairs[:, 1]
    out = []
    for q, t in ((np.asarray(q_a, float), np.asarray(t_a, float)), (np.asarray(q_b, float), np.asarray(t_b, float))):
        R = _rot(q)
        Rij = np.einsum("nab,ncb->nac", R[i], R[j])
        tij = t[i] - np.einsum("nab,nb->na", Rij, t[j])
        out.append((Rij, tij))
    dR = np.einsum("nab,ncb->nac", out[0][0], out[1][0])
    ang = np.arccos(np.clip((np.trace(dR, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0))
    # arccos loses accuracy near 0: use the antisymmetric part for small angles
    small = 0.5 * np.sqrt(((dR - dR.transpose(0, 2, 1)) ** 2).sum((1, 2)) / 2.0)
    ang = np.where(ang < 1e-4, small, ang)
    return float(ang.max()), float(np.abs(out[0][1] - out[1][1]).max())


def camera_difference_spectrum(problem, q_other, t_other, device="cuda"):
    """`problem`: capi.ProblemArrays holding result A (its cam_q / cam_t / points are the linearisation point);
    q_other / t_other: result B.  Returns eigenvalues (ascending) of S restricted to the free camera coordinates, the
    coefficients of the tangent difference in that basis, norms, the energy d^T S d and rest_inf(k) = the largest component
    of the difference once the k weakest modes are projected out."""
    import torch
    from xrsfm_amd import capi
    ctx = capi.Context(problem)
    ctx.debug_linearize(5.99, False)              # unscaled columns: S is in the tangent coordinates of the parameters
    _, S = ctx.debug_cholesky_solve(1e30, want_S=True)      # radius 1e30: no LM damping
    ctx.close()
    d = tangent_difference(problem.cam_q, problem.cam_t, q_other, t_other)
    const = np.asarray(problem.cam_const if problem.cam_const is not None else np.zeros(problem.n_cams, np.uint8))
    free = np.ones((problem.n_cams, 6), bool)
    free[(const & 1) != 0, :3] = False
    free[(const & 2) != 0, 3:] = False
    active = np.bincount(problem.obs_cam, minlength=problem.n_cams) > 0
    free[~active] = False
    idx = np.nonzero(free.reshape(-1))[0]
    Sf = torch.from_numpy(S[np.ix_(idx, idx)]).to(device)
    Sf = 0.5 * (Sf + Sf.T)
    lam, V = torch.linalg.eigh(Sf)
    df = torch.from_numpy(d[idx]).to(device)
    c = V.T @ df
    energy = float(df @ (Sf @ df))
    Vc = V * c            # column i = c_i v_i

    def rest_inf(k: int) -> float:
        k = max(0, min(int(k), c.shape[0]))
        return float((df - Vc[:, :k].sum(1)).abs().max())

    return {"eigenvalues": lam.cpu().numpy(), "coefficients": c.cpu().numpy(), "d_inf": float(np.abs(d).max()),
            "d_2": float(np.linalg.norm(d)), "energy": energy, "rest_inf": rest_inf, "n_free": int(idx.shape[0])}
