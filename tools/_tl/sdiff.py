import math, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import ba_oracle as bo
from tests import helpers as H
from tests.test_gpu_fuzz import _problem
from tests.test_gpu_parity import _lin_oracle
from xrsfm_amd import capi
seed = int(sys.argv[1])
arr, _ = _problem(seed)
n_cams = arr["cam_q"].shape[0]; n_pts = arr["points"].shape[0]
pr = H.to_oracle(arr)
ctx = capi.Context(H.to_product(arr))
ctx.debug_linearize(5.99, True)
_, _, _, _, lin = _lin_oracle(pr, True)
radius = 1e4
Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
ci, pi = pr.obs_cam, pr.obs_pt
n = 6 * n_cams
S_ref = np.zeros((n, n))
WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
for c in range(n_cams):
    S_ref[6 * c:6 * c + 6, 6 * c:6 * c + 6] = lin.Hcc[c] + np.diag(Dc2[c])
order = np.argsort(pi, kind="stable")
ptr = np.searchsorted(pi[order], np.arange(n_pts + 1))
for j in range(n_pts):
    ids = order[ptr[j]:ptr[j + 1]]
    for a in ids:
        for b2 in ids:
            S_ref[6 * ci[a]:6 * ci[a] + 6, 6 * ci[b2]:6 * ci[b2] + 6] -= WH[a] @ lin.W[b2].T
y, S = ctx.debug_cholesky_solve(radius, want_S=True)
print("rel err S", H.rel_err(S, S_ref), "finite", np.isfinite(S).all())
D = np.abs(S - S_ref).reshape(n_cams, 6, n_cams, 6).max(axis=(1, 3))
R = np.abs(S_ref).reshape(n_cams, 6, n_cams, 6).max(axis=(1, 3))
bad = np.argwhere(D > 1e-9 * R.max())
print("bad blocks", len(bad), "of", n_cams * n_cams, "first:", bad[:20].tolist())
print("diag bad", [int(c) for c in range(n_cams) if D[c, c] > 1e-9 * R.max()][:40])
cc = arr["cam_const"]; print("const cams", np.nonzero(cc)[0].tolist(), cc[np.nonzero(cc)[0]].tolist())
ctx.close()
