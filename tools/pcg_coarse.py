#!/usr/bin/env python3
"""Developer tool (CPU, no GPU): what the PCG path's preconditioner has to deal with, on the dense reduced camera matrix of a small
synthetic problem built with the oracle (oracle/ba_oracle.py — test infrastructure; this tool is not part of the product).

  python tools/pcg_coarse.py unordered 300 30000 5 1e4,1e8,1e12      # random visibility (the shape of bench.py --config V)
  python tools/pcg_coarse.py collection 600 60000 100 1e4,1e8        # viewpoint clusters (the shape of --config T), 100 photos per cluster

For every trust-region radius: the smallest eigenvalues of the block-Jacobi-preconditioned matrix (up to seven of them sit at the
level of the LM damping: the gauge directions of the reconstruction), and the PCG iteration counts to |r| <= 1e-12 |b| with
block-Jacobi alone, with the seven global gauge vectors as an additive coarse space (what the HIP path runs: ba_kernels.h
k_pcg_gauge / k_pcg_coarse), and with one set of gauge vectors per group of cameras (not built: see DESIGN.md section 4).
Round 5 numbers: random visibility 41-53 -> 14 iterations whatever the radius; clusters 119-217 -> 81-90 (global) -> 63-66 (per cluster)."""
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import ba_oracle as bo          # noqa: E402
from xrsfm_amd import capi, synth           # noqa: E402


def build(d, radius):
    """Dense S + D^2 (Jacobi-scaled), right-hand side, the problem and the camera scales at the initial point."""
    pr = bo.Problem(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})
    q, t, P = pr.cam_q, pr.cam_t, pr.points
    _, rt, Fc, Ep = bo.evaluate(pr, q, t, P, bo.HUBER_A)
    ci, pi = pr.obs_cam, pr.obs_pt
    sc_c = 1.0 / (1.0 + np.sqrt(bo._scatter_add(q.shape[0], ci, np.sum(Fc * Fc, axis=1))))
    sc_p = 1.0 / (1.0 + np.sqrt(bo._scatter_add(P.shape[0], pi, np.sum(Ep * Ep, axis=1))))
    lin = bo._Linearization(pr, rt, Fc * sc_c[ci][:, None, :], Ep * sc_p[pi][:, None, :])
    Nc = q.shape[0]
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
    b = lin.gc - bo._scatter_add(Nc, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    S = np.zeros((Nc * 6, Nc * 6))
    for c in range(Nc):
        S[6 * c:6 * c + 6, 6 * c:6 * c + 6] = lin.Hcc[c] + np.diag(Dc2[c])
    order = np.argsort(pi, kind="stable")
    ptr = np.searchsorted(pi[order], np.arange(P.shape[0] + 1))
    lens = np.diff(ptr)
    i6 = np.arange(6)
    rows_l, cols_l, vals_l = [], [], []
    for L in np.unique(lens):
        if L == 0:
            continue
        idx = order[ptr[np.nonzero(lens == L)[0]][:, None] + np.arange(L)[None, :]]
        for a_ in range(L):
            for b_ in range(L):
                blk = np.einsum("nij,nkj->nik", WH[idx[:, a_]], lin.W[idx[:, b_]])
                rows = np.broadcast_to(6 * ci[idx[:, a_]][:, None, None] + i6[None, :, None], blk.shape)
                cols = np.broadcast_to(6 * ci[idx[:, b_]][:, None, None] + i6[None, None, :], blk.shape)
                rows_l.append(rows.reshape(-1)); cols_l.append(cols.reshape(-1)); vals_l.append(blk.reshape(-1))
    S -= sp.coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))), shape=S.shape).toarray()
    return S, b.reshape(-1), pr, sc_c


def pcg(S, b, precond, tol=1e-12, maxit=5000):
    x = np.zeros_like(b); r = b.copy(); z = precond(r); p = z.copy(); rz = r @ z; bn = np.linalg.norm(b); it = 0
    while it < maxit and np.linalg.norm(r) > tol * bn:
        q = S @ p; a = rz / (p @ q); x += a * p; r -= a * q; z = precond(r); rzn = r @ z; p = z + (rzn / rz) * p; rz = rzn; it += 1
    return x, it


def main():
    mode, n_cams, n_pts, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    radii = [float(x) for x in sys.argv[5].split(",")]
    if mode == "collection":
        d = synth.make_collection(n_cams=n_cams, n_points=n_pts, seed=12, cams_per_cluster=k, shuffle_ids=False)
    else:
        d = synth.make_problem(n_cams=n_cams, n_points=n_pts, k_obs=k, seed=8, mode=mode)
    for radius in radii:
        t0 = time.time()
        S, b, pr, sc_c = build(d, radius)
        n = S.shape[0]
        Minv = np.stack([np.linalg.inv(S[6 * c:6 * c + 6, 6 * c:6 * c + 6]) for c in range(n_cams)])
        W = bo.gauge_vectors(pr.cam_q, pr.cam_t, pr.cam_const, sc_c).reshape(n, 7)

        def bj(r):
            return np.einsum("nij,nj->ni", Minv, r.reshape(-1, 6)).reshape(-1)

        def two_level(Wm):
            E = Wm.T @ (S @ Wm)
            Einv = np.linalg.pinv(0.5 * (E + E.T), hermitian=True)
            return lambda r: bj(r) + Wm @ (Einv @ (Wm.T @ r))

        _, it_bj = pcg(S, b, bj)
        _, it_g = pcg(S, b, two_level(W))
        line = f"radius {radius:.0e}: block-Jacobi {it_bj} iterations | + global gauge {it_g}"
        if mode == "collection":
            ng = (n_cams + k - 1) // k
            Wg = np.zeros((n, 7 * ng))
            for g in range(ng):
                rows = slice(6 * g * k, min(n, 6 * (g + 1) * k))
                Wg[rows, 7 * g:7 * g + 7] = W[rows]
            _, it_c = pcg(S, b, two_level(Wg))
            line += f" | + gauge per cluster ({7 * ng} coarse unknowns) {it_c}"
        if n <= 4000:
            Lb = [np.linalg.cholesky(S[6 * c:6 * c + 6, 6 * c:6 * c + 6]) for c in range(n_cams)]
            A = S.copy()
            for c in range(n_cams):
                A[6 * c:6 * c + 6, :] = np.linalg.solve(Lb[c], A[6 * c:6 * c + 6, :])
            for c in range(n_cams):
                A[:, 6 * c:6 * c + 6] = np.linalg.solve(Lb[c], A[:, 6 * c:6 * c + 6].T).T
            w = np.linalg.eigvalsh(0.5 * (A + A.T))
            line += "\n    block-Jacobi-preconditioned spectrum: smallest 9 " + np.array2string(w[:9], precision=3) + f", bulk [{w[9]:.3f}, {w[-1]:.3f}]"
        print(line + f"   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
