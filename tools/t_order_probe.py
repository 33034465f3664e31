#!/usr/bin/env python3
"""Developer aid (host only, no GPU): is the elimination ORDER what config T's factorisation costs?  Builds the camera graph of the
collection, takes the plan's tiles (cam_offset // 64) as super-nodes, and counts — by symbolic elimination on the tile graph — the
non-zero tiles, the tile products sum_k c_k (c_k + 1) / 2 and the elimination-tree height of (a) the plan's order (must reproduce
the plan's tiles_nz) and (b) an exact minimum-degree order of the same tiles.
usage: python tools/t_order_probe.py [config]      (default T; ~40 s)"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrsfm_amd import capi, synth      # noqa: E402


def symbolic(G, order):
    n = G.shape[0]
    pos = np.empty(n, int); pos[order] = np.arange(n)
    adj = [set() for _ in range(n)]
    Gc = G.tocoo()
    for i, j in zip(pos[Gc.row], pos[Gc.col]):
        if i > j:
            adj[j].add(i)
    prod = nnz = 0
    parent = [-1] * n
    cnt = np.zeros(n, int)
    for k in range(n):
        s = adj[k]; c = len(s); cnt[k] = c
        nnz += c + 1; prod += c * (c + 1) // 2
        if c:
            m = min(s); parent[k] = m
            adj[m] |= (s - {m})
    h = [0] * n
    for k in range(n):
        if parent[k] >= 0:
            h[parent[k]] = max(h[parent[k]], h[k] + 1)
    return prod, nnz, max(h) + 1, cnt


def min_degree(G):
    n = G.shape[0]
    adj = [set(G.indices[G.indptr[i]:G.indptr[i + 1]]) - {i} for i in range(n)]
    alive = set(range(n)); order = []
    while alive:
        k = min(alive, key=lambda v: (len(adj[v]), v))
        order.append(k); alive.discard(k)
        nb = adj[k]
        for u in nb:
            adj[u] |= nb; adj[u].discard(u); adj[u].discard(k)
    return order


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "T"
    mk = synth.make_collection if cfg == "T" else synth.make_problem
    d = mk(**synth.CONFIGS[cfg])
    p = capi.ProblemArrays(**{k: d[k] for k in capi.ProblemArrays.FIELDS})
    plan = capi.debug_chol_plan(p)
    print("plan:", {k: v for k, v in plan.items() if k != "cam_offset"})
    tile = plan["cam_offset"] // 64
    B = sp.csr_matrix((np.ones(len(d["obs_cam"]), np.float32), (d["obs_cam"], d["obs_pt"])), shape=(p.n_cams, d["points"].shape[0]))
    A = (B @ B.T).tocoo()
    T = int(tile.max()) + 1
    G = sp.csr_matrix((np.ones(A.nnz, np.int8), (tile[A.row], tile[A.col])), shape=(T, T)); G.data[:] = 1
    deg = np.diff(G.indptr) - 1
    print(f"tile graph: {T} tiles, mean degree {deg.mean():.1f}, max {deg.max()}")
    prod, nnz, h, cnt = symbolic(G, list(range(T)))
    print(f"plan order:      {prod} tile products ({prod * 2 * 64 ** 3 / 1e12:.2f} TFLOP), {nnz} non-zero tiles, tree height {h}; tiles below a pivot: "
          f"median {int(np.median(cnt))}, 90 % {int(np.percentile(cnt, 90))}")
    prod, nnz, h, cnt = symbolic(G, min_degree(G))
    print(f"minimum degree:  {prod} tile products, {nnz} non-zero tiles, tree height {h}")


if __name__ == "__main__":
    main()
