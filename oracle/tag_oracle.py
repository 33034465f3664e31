"""TEST INFRASTRUCTURE — CPU restatement of the two solves of tag_refine (/root/reference/src/tag/tag_extract.hpp:193-265).

Residuals restated in numpy from /root/reference/src/optimization/cost_factor_ceres.h: TagCost (:223-260), ProjectionCost
(:66-112, including the functor's own sqrt(sigma/|r|) down-weighting beyond sigma = 5.99/700) and the right-multiplied
quaternion update of QuatParam (:262-282).  The minimiser is scipy.optimize.least_squares (trust-region reflective with the
reference's lower bound on the scale), an implementation independent of xrsfm_amd/csrc/tag_refine.h.  Ceres is not available
here: PARITY UNPINNED against the real Levenberg-Marquardt trajectory; what is pinned is the minimum.
Note: ProjectionCost hands Ceres a Jacobian that ignores the derivative of its down-weighting factor; with residuals beyond
sigma the fixed point of Ceres' iteration is then the root of J^T r with that truncated J, not the minimiser of sum r^2.
least_squares gets the same truncated Jacobian through `jac`, so both converge to the same kind of point.
Only tests/ may import this module.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

SIGMA = 5.99 / 700.0


def tag_points(tag_length):
    L = tag_length
    return np.array([[0, 0, 0], [L, 0, 0], [L, 0, L], [0, 0, L]], float)


def projection(frame_R, frame_t, f, xy, Pw, want_jac=False):
    """ProjectionCost for observation arrays: residual [n,2] (and d r/d Pw [n,2,3])."""
    R = frame_R[f]
    pc = np.einsum("nij,nj->ni", R, Pw) + frame_t[f]
    z = pc[:, 2]
    r = pc[:, :2] / z[:, None] - xy
    nr = np.sqrt((r * r).sum(1))
    hf = np.where(nr > SIGMA, np.sqrt(SIGMA / np.maximum(nr, 1e-300)), 1.0)
    r = r * hf[:, None]
    if not want_jac:
        return r
    d = np.zeros((len(z), 2, 3))
    d[:, 0, 0] = 1 / z; d[:, 0, 2] = -pc[:, 0] / z ** 2
    d[:, 1, 1] = 1 / z; d[:, 1, 2] = -pc[:, 1] / z ** 2
    d *= hf[:, None, None]
    return r, np.einsum("nij,njk->nik", d, R)


def tag_residuals(tag_q, tag_t, scale, corners, tag_length):
    """TagCost rows [n_tags,4,3]."""
    R = Rotation.from_quat(tag_q).as_matrix()
    p0 = tag_points(tag_length)
    return corners - (scale * np.einsum("kij,cj->kci", R, p0) + tag_t[:, None, :])


def cost(frame_q, frame_t, tag_q, tag_t, scale, corners, tag_length, stage, tag_obs=None, points=None, obs=None):
    """1/2 sum r^2 over the residual blocks that have a variable parameter in `stage`."""
    c = (tag_residuals(tag_q, tag_t, scale, corners, tag_length) ** 2).sum()
    if stage == 2:
        frame_R = Rotation.from_quat(frame_q).as_matrix()
        k, f, xy = tag_obs
        for cidx in range(4):
            c += (projection(frame_R, frame_t, f, xy[:, cidx], corners[k, cidx]) ** 2).sum()
        if points is not None and len(obs[0]):
            c += (projection(frame_R, frame_t, obs[0], obs[2], points[obs[1]]) ** 2).sum()
    return 0.5 * float(c)


def solve_stage1(corners, tag_length, scale_lower=0.2):
    """Tag poses and the common scale against constant corners.  Returns tag_q, tag_t, scale, cost."""
    corners = np.asarray(corners, float).reshape(-1, 4, 3)
    T = corners.shape[0]

    def unpack(x):
        return Rotation.from_rotvec(x[1:1 + 3 * T].reshape(T, 3)).as_quat(), x[1 + 3 * T:].reshape(T, 3), x[0]

    def fun(x):
        q, t, s = unpack(x)
        return tag_residuals(q, t, s, corners, tag_length).ravel()

    x0 = np.concatenate([[1.0], np.zeros(3 * T), np.zeros(3 * T)])
    lo = np.full(x0.shape, -np.inf); lo[0] = scale_lower
    best = None
    # the rotation part of a similarity alignment has one minimum but rotvec charts have singularities: a few starts
    for seed in range(4):
        xs = x0.copy()
        if seed:
            xs[1:1 + 3 * T] = np.random.default_rng(seed).normal(0, 1.0, 3 * T)
        res = least_squares(fun, xs, bounds=(lo, np.full(x0.shape, np.inf)), xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=4000)
        if best is None or res.cost < best.cost:
            best = res
    q, t, s = unpack(best.x)
    return q, t, s, float(best.cost)


def solve_stage2(frame_q, frame_t, tag_q, tag_t, scale, corners, tag_length, tag_obs, points, obs, scale_lower=0.2):
    """Everything but the frames variable.  The track points do not couple with the tag unknowns (all cameras constant), so
    they are refined one by one (3 unknowns each) and the tag part (scale, poses, corners) in one small problem."""
    frame_R = Rotation.from_quat(frame_q).as_matrix()
    corners = np.array(corners, float).reshape(-1, 4, 3)
    T = corners.shape[0]
    k, f, xy = tag_obs
    q0 = Rotation.from_quat(tag_q)

    def unpack(x):
        dq = Rotation.from_rotvec(x[1:1 + 3 * T].reshape(T, 3))
        return (q0 * dq).as_quat(), x[1 + 3 * T:1 + 6 * T].reshape(T, 3), x[0], x[1 + 6 * T:].reshape(T, 4, 3)

    def fun(x):
        q, t, s, c = unpack(x)
        out = [tag_residuals(q, t, s, c, tag_length).ravel()]
        for cidx in range(4):
            out.append(projection(frame_R, frame_t, f, xy[:, cidx], c[k, cidx]).ravel())
        return np.concatenate(out)

    def jac(x):            # the Jacobian Ceres is given: analytic, down-weighting factor treated as a constant
        q, t, s, c = unpack(x)
        R = Rotation.from_quat(q).as_matrix()
        p0 = tag_points(tag_length)
        n_rows = 12 * T + 8 * len(k)
        J = np.zeros((n_rows, x.size))
        row = 0
        for kk in range(T):
            for cidx in range(4):
                v = s * p0[cidx]
                K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
                J[row:row + 3, 0] = -R[kk] @ p0[cidx]
                J[row:row + 3, 1 + 3 * kk:4 + 3 * kk] = R[kk] @ K
                J[row:row + 3, 1 + 3 * T + 3 * kk:4 + 3 * T + 3 * kk] = -np.eye(3)
                o = 1 + 6 * T + 12 * kk + 3 * cidx
                J[row:row + 3, o:o + 3] = np.eye(3)
                row += 3
        for cidx in range(4):
            _, Jp = projection(frame_R, frame_t, f, xy[:, cidx], c[k, cidx], want_jac=True)
            for i in range(len(k)):
                o = 1 + 6 * T + 12 * k[i] + 3 * cidx
                J[row:row + 2, o:o + 3] = Jp[i]
                row += 2
        return J

    x0 = np.concatenate([[scale], np.zeros(3 * T), np.asarray(tag_t, float).ravel(), corners.ravel()])
    lo = np.full(x0.shape, -np.inf); lo[0] = scale_lower
    res = least_squares(fun, x0, jac=jac, bounds=(lo, np.full(x0.shape, np.inf)), xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=4000)
    q, t, s, c = unpack(res.x)
    pts = None
    if points is not None:
        pts = np.array(points, float)
        of, op, oxy = obs
        order = np.argsort(op, kind="stable"); ptr = np.searchsorted(op[order], np.arange(pts.shape[0] + 1))
        for j in range(pts.shape[0]):
            ids = order[ptr[j]:ptr[j + 1]]
            if len(ids) == 0:
                continue
            fj = lambda P: projection(frame_R, frame_t, of[ids], oxy[ids], np.tile(P, (len(ids), 1))).ravel()
            jj = lambda P: projection(frame_R, frame_t, of[ids], oxy[ids], np.tile(P, (len(ids), 1)), want_jac=True)[1].reshape(-1, 3)
            pts[j] = least_squares(fj, pts[j], jac=jj, xtol=1e-15, ftol=1e-15, gtol=1e-14, max_nfev=400).x
    return q, t, s, c, pts
