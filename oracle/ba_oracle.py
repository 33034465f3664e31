"""CPU oracle (numpy/scipy, FP64) for XRSfM's bundle-adjustment hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``xrsfm_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do.  The product path is the HIP library behind
``include/xrsfm_ba.h``.

PARITY UNPINNED: the arithmetic of the reference path lives in Ceres-Solver
(``ceres::Solve`` called at /root/reference/src/optimization/ba_solver.cc:591,
636,672), which is neither vendored under /root/reference nor installed in
this image, and the reference ships no tests or golden vectors for this path
(SURVEY.md section 4 and 8c).  The trust-region semantics below restate the
published Ceres 2.0/2.1 algorithm (TrustRegionMinimizer,
LevenbergMarquardtStrategy, SchurComplementSolver, HuberLoss, Corrector,
EigenQuaternionParameterization) — SURVEY.md Appendix A.  What *is* pinned to
reference source is the residual model:

* residual                /root/reference/src/optimization/cost_factor_ceres.h:19-40
* camera models           /root/reference/src/base/camera_model.hpp:57-68, 93-209
* problem construction    /root/reference/src/optimization/ba_solver.cc:330-391,594-678
* solver options          /root/reference/src/optimization/ba_solver.cc:70-77,586-589,626-634,667-670

The Jacobians here are cross-checked against torch.autograd (FP64) in
tests/test_oracle_jacobian.py, i.e. against what Ceres' autodiff of the
reference functor produces.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import numpy as np

HUBER_A = 5.99          # ba_solver.cc:343,374
MIN_DEPTH = 1e-2        # cost_factor_ceres.h:29
CLAMP_RES = 12.0        # cost_factor_ceres.h:31

# camera_model.hpp: model id -> (num params, ifx, ify, icx, icy, idistort)
MODEL_INFO = {
    0: (3, 0, 0, 1, 2, -1),   # SIMPLE_PINHOLE :93-110
    1: (4, 0, 1, 2, 3, -1),   # PINHOLE        :112-129
    2: (4, 0, 0, 1, 2, 3),    # SIMPLE_RADIAL  :131-153
    3: (5, 0, 1, 2, 3, 4),    # RADIAL         :155-177 (single k)
    4: (8, 0, 1, 2, 3, 4),    # OPENCV         :179-209
}
# Extension (NOT one of the reference's camera models): id 5 = BAL-style radial camera {f, k1, k2}, no principal point,
# uv = f (1 + k1 r^2 + k2 r^4) xy with the reference's sign convention xy = pc.hnormalized().  It exists for the
# "bal9" mode of SURVEY.md section 8(d) / BASELINE.json north_star ("2x9 camera blocks"): a camera whose cam_const has bit 2
# (value 4) set keeps its intrinsics block VARIABLE — the reference always holds it constant (ba_solver.cc:602-606, 655-659,
# 389) — and then contributes a 9-wide block {rotation 3, translation 3, f, k1, k2}.
MODEL_BAL = 5
INTR_VARIABLE = 4


@dataclasses.dataclass
class Problem:
    """Flat SoA view of one BA call (the same arrays the C-ABI takes)."""
    cam_q: np.ndarray        # [Nc,4] x,y,z,w (Eigen coeffs order, ba_solver.cc:346)
    cam_t: np.ndarray        # [Nc,3]
    cam_const: np.ndarray    # [Nc] uint8  bit0: q constant, bit1: t constant, bit2: intrinsics VARIABLE (model 5 only: bal9 mode)
    cam_intr: np.ndarray     # [Nc] int32
    intr_model: np.ndarray   # [Ni] int32
    intr_params: np.ndarray  # [Ni,8]
    points: np.ndarray       # [Np,3]
    point_const: np.ndarray  # [Np] uint8
    obs_cam: np.ndarray      # [No] int32
    obs_pt: np.ndarray       # [No] int32
    obs_uv: np.ndarray       # [No,2]

    def copy(self) -> "Problem":
        return Problem(**{f.name: np.array(getattr(self, f.name), copy=True)
                          for f in dataclasses.fields(self)})


@dataclasses.dataclass
class Options:
    """Ceres options as set by the reference (ba_solver.cc) + library defaults."""
    max_iterations: int = 50             # GBA accurate  :627
    function_tolerance: float = 1e-5     # :628
    parameter_tolerance: float = 1e-6    # :629
    gradient_tolerance: float = 1e-10    # Ceres default
    initial_radius: float = 1e4          # Ceres default (KGBA 1e6, :667)
    max_radius: float = 1e16
    min_radius: float = 1e-32
    min_relative_decrease: float = 1e-3
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    max_consecutive_invalid_steps: int = 5
    huber_a: float = HUBER_A
    linear_solver: str = "exact"         # "exact" (Schur + Cholesky) | "pcg"
    pcg_tol: float = 1e-12
    pcg_max_iter: int = 500
    pcg_coarse: bool = False             # two-level preconditioner of the HIP PCG path (block-Jacobi + the seven gauge vectors); False = block-Jacobi
                                         # alone, what the committed goldens were generated with (the converged solutions agree either way)
    # Sensitivity switches (tools/oracle_sensitivity.py; every default = the recalled Ceres 2.0/2.1 behaviour of SURVEY.md
    # Appendix A, which is UNPINNED against real Ceres): each flips ONE recalled detail so that its effect on step counts and
    # the final RMSE can be tabulated.  Never set by tests of the product.
    alt: str = ""        # one of ALT_DETAILS, or "" for the restatement as recalled


# recalled detail -> what the alternative does
ALT_DETAILS = {
    "accept_before_tolerance": "a step that triggers the parameter / function tolerance exit is ACCEPTED first if rho allows it (Ceres "
                               "<= 1.11 order) instead of being discarded (TrustRegionMinimizer::Minimize of 1.12+: tolerance tests precede IsStepSuccessful)",
    "min_relative_decrease_1e-4": "min_relative_decrease 1e-4 instead of the default 1e-3",
    "jacobi_scaling_off": "jacobi_scaling = false",
    "jacobi_scaling_plain_inverse": "column scaling 1/|col| (guarded) instead of 1/(1 + |col|)",
    "jacobi_scaling_unrobustified": "column norms of the Jacobian BEFORE the loss correction",
    "lm_diagonal_unclamped": "LM diagonal diag(J^T J)/radius without the clamp to [1e-6, 1e32]",
    "lm_diagonal_min_1e-9": "min_lm_diagonal 1e-9",
    "radius_halving_on_reject": "rejected step: radius /= 2 every time (no doubling decrease factor)",
    "radius_factor_capped_at_2": "accepted step: radius grows by at most 2 (min(3, .) -> min(2, .))",
    "invalid_step_is_plain_reject": "a step with model_cost_change <= 0 is handled like any rejected step (no invalid-step counter)",
    "gradient_norm_plain": "gradient tolerance on |g|_inf instead of |x - Plus(x, -g)|_inf",
    "x_norm_includes_constant_blocks": "the parameter-tolerance test uses |x| over ALL parameters, constant blocks included",
    "function_tolerance_vs_candidate_cost": "function tolerance |dcost| <= ftol * candidate cost instead of the current cost",
    "iteration_zero_counted": "iteration 0 (the evaluation at the initial point) is counted as a step: FinalizeIterationAndCheckIfMinimizerCanContinue "
                              "also runs after IterationZero() in Ceres >= 1.12 — if it increments a counter there, the reference's printed "
                              "'Iterations' (num_successful_steps + num_unsuccessful_steps, ba_solver.cc:22-25) and the numerator of the "
                              "bench metric are one higher per solve; no state changes (the restatement and include/xrsfm_ba.h count LM "
                              "steps only)",
}


@dataclasses.dataclass
class Summary:
    initial_cost: float = 0.0
    final_cost: float = 0.0
    num_residuals: int = 0
    num_effective_params: int = 0
    n_successful: int = 0
    n_unsuccessful: int = 0
    termination: str = ""
    trace: list = dataclasses.field(default_factory=list)
    pcg_iterations: int = 0


# ----------------------------------------------------------------------------
# residual + Jacobian (Appendix A.1 / A.2)
# ----------------------------------------------------------------------------
def rotation_from_quat(q: np.ndarray) -> np.ndarray:
    """M(q) = I + 2w[u]x + 2[u]x^2 for q = (x,y,z,w); equals R(q) for unit q.

    This is Eigen's QuaternionBase::_transformVector written as a matrix (the
    reference applies ``qcw * pw`` without normalising, cost_factor_ceres.h:28).
    """
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    M = np.empty(q.shape[:-1] + (3, 3))
    M[..., 0, 0] = 1 - 2 * (y * y + z * z)
    M[..., 0, 1] = 2 * (x * y - w * z)
    M[..., 0, 2] = 2 * (x * z + w * y)
    M[..., 1, 0] = 2 * (x * y + w * z)
    M[..., 1, 1] = 1 - 2 * (x * x + z * z)
    M[..., 1, 2] = 2 * (y * z - w * x)
    M[..., 2, 0] = 2 * (x * z - w * y)
    M[..., 2, 1] = 2 * (y * z + w * x)
    M[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return M


def _intrinsics_per_obs(model: np.ndarray, prm: np.ndarray):
    """Return fx, fy, cx, cy and the per-model distortion parameters."""
    n = model.shape[0]
    fx = np.empty(n); fy = np.empty(n); cx = np.empty(n); cy = np.empty(n)
    for mid, (_, ifx, ify, icx, icy, _) in MODEL_INFO.items():
        m = model == mid
        if m.any():
            fx[m] = prm[m, ifx]; fy[m] = prm[m, ify]
            cx[m] = prm[m, icx]; cy[m] = prm[m, icy]
    m = model == MODEL_BAL
    if m.any():
        fx[m] = prm[m, 0]; fy[m] = prm[m, 0]; cx[m] = 0.0; cy[m] = 0.0
    return fx, fy, cx, cy


def project(problem: Problem, q=None, t=None, P=None, want_jac=True, intr=None, want_intr_jac=False):
    """Per-observation residual and (unrobustified) Jacobian blocks.

    Returns r [No,2], valid [No] and, if want_jac, J_rot [No,2,3] (w.r.t. the
    3-dim local rotation increment of EigenQuaternionParameterization),
    J_t [No,2,3], J_P [No,2,3].  Clamp case z < 1e-2: r = (12,12), J = 0
    (cost_factor_ceres.h:29-31).
    """
    q = problem.cam_q if q is None else q
    t = problem.cam_t if t is None else t
    P = problem.points if P is None else P
    ci, pi = problem.obs_cam, problem.obs_pt
    M = rotation_from_quat(q)[ci]                       # [No,3,3]
    Pw = P[pi]
    RP = np.einsum("nij,nj->ni", M, Pw)
    Pc = RP + t[ci]
    z = Pc[:, 2]
    valid = ~(z < MIN_DEPTH)
    zs = np.where(valid, z, 1.0)
    iz = 1.0 / zs
    xn = Pc[:, 0] * iz
    yn = Pc[:, 1] * iz

    intr_idx = problem.cam_intr[ci]
    model = problem.intr_model[intr_idx]
    prm = (problem.intr_params if intr is None else intr)[intr_idx]
    fx, fy, cx, cy = _intrinsics_per_obs(model, prm)

    n = ci.shape[0]
    du = np.zeros(n); dv = np.zeros(n)
    # D = I + d(du,dv)/d(xn,yn)
    D00 = np.ones(n); D01 = np.zeros(n); D10 = np.zeros(n); D11 = np.ones(n)
    r2 = xn * xn + yn * yn

    m01 = (model == 0) | (model == 1)           # reference quirk: duv = xy (camera_model.hpp:102-105,121-124)
    du[m01] = xn[m01]; dv[m01] = yn[m01]
    D00[m01] = 2.0; D11[m01] = 2.0

    for mid, ik in ((2, 3), (3, 4)):            # SIMPLE_RADIAL / RADIAL: k*r2*(xn,yn)
        m = model == mid
        if m.any():
            k = prm[m, ik]
            du[m] = xn[m] * (k * r2[m]); dv[m] = yn[m] * (k * r2[m])
            D00[m] = 1 + k * r2[m] + 2 * k * xn[m] ** 2
            D11[m] = 1 + k * r2[m] + 2 * k * yn[m] ** 2
            D01[m] = 2 * k * xn[m] * yn[m]
            D10[m] = D01[m]

    m = model == 4                              # OPENCV :188-204
    if m.any():
        k1, k2, p1, p2 = prm[m, 4], prm[m, 5], prm[m, 6], prm[m, 7]
        x, y, rr = xn[m], yn[m], r2[m]
        rad = k1 * rr + k2 * rr * rr
        du[m] = x * rad + 2 * p1 * x * y + p2 * (rr + 2 * x * x)
        dv[m] = y * rad + 2 * p2 * x * y + p1 * (rr + 2 * y * y)
        rad_x = 2 * k1 * x + 4 * k2 * rr * x
        rad_y = 2 * k1 * y + 4 * k2 * rr * y
        D00[m] = 1 + rad + x * rad_x + 2 * p1 * y + 6 * p2 * x
        D01[m] = x * rad_y + 2 * p1 * x + 2 * p2 * y
        D10[m] = y * rad_x + 2 * p2 * y + 2 * p1 * x
        D11[m] = 1 + rad + y * rad_y + 2 * p2 * x + 6 * p1 * y

    m = model == MODEL_BAL                      # extension: f (1 + k1 r2 + k2 r2^2) xy
    if m.any():
        k1, k2 = prm[m, 1], prm[m, 2]
        x, y, rr = xn[m], yn[m], r2[m]
        rad = k1 * rr + k2 * rr * rr
        du[m] = x * rad; dv[m] = y * rad
        rad_x = 2 * k1 * x + 4 * k2 * rr * x
        rad_y = 2 * k1 * y + 4 * k2 * rr * y
        D00[m] = 1 + rad + x * rad_x; D01[m] = x * rad_y
        D10[m] = y * rad_x; D11[m] = 1 + rad + y * rad_y

    r = np.empty((n, 2))
    r[:, 0] = fx * (xn + du) + cx - problem.obs_uv[:, 0]
    r[:, 1] = fy * (yn + dv) + cy - problem.obs_uv[:, 1]
    r[~valid] = CLAMP_RES
    if not want_jac:
        return r, valid

    A00 = fx * D00; A01 = fx * D01; A10 = fy * D10; A11 = fy * D11
    Jproj = np.zeros((n, 2, 3))
    Jproj[:, 0, 0] = A00 * iz; Jproj[:, 0, 1] = A01 * iz
    Jproj[:, 0, 2] = -(A00 * xn + A01 * yn) * iz
    Jproj[:, 1, 0] = A10 * iz; Jproj[:, 1, 1] = A11 * iz
    Jproj[:, 1, 2] = -(A10 * xn + A11 * yn) * iz
    Jproj[~valid] = 0.0
    J_t = Jproj
    J_P = np.einsum("nij,njk->nik", Jproj, M)
    # row^T * (-2 [RP]x) = -2 * (row x RP)
    J_rot = -2.0 * np.cross(Jproj, RP[:, None, :])
    if want_intr_jac:
        # d r / d (f, k1, k2) of model 5 (zero for the reference's models, whose intrinsics are always constant)
        J_i = np.zeros((n, 2, 3))
        m = (model == MODEL_BAL) & valid
        if m.any():
            x, y, rr, f = xn[m], yn[m], r2[m], fx[m]
            J_i[m, 0, 0] = x + du[m]; J_i[m, 1, 0] = y + dv[m]
            J_i[m, 0, 1] = f * x * rr; J_i[m, 1, 1] = f * y * rr
            J_i[m, 0, 2] = f * x * rr * rr; J_i[m, 1, 2] = f * y * rr * rr
        return r, valid, J_rot, J_t, J_P, J_i
    return r, valid, J_rot, J_t, J_P


def huber(s: np.ndarray, a: float = HUBER_A):
    """ceres::HuberLoss(a): rho, rho' on s = |r|^2 (Appendix A.3)."""
    b = a * a
    out = s > b
    sq = np.sqrt(np.where(out, s, 1.0))
    rho = np.where(out, 2 * a * sq - b, s)
    rho1 = np.where(out, np.maximum(np.finfo(float).tiny, a / sq), 1.0)
    return rho, rho1


def quat_plus(q: np.ndarray, d: np.ndarray) -> np.ndarray:
    """EigenQuaternionParameterization::Plus: dq(full angle) (x) q, q = xyzw."""
    n = np.linalg.norm(d, axis=-1)
    safe = np.where(n > 0, n, 1.0)
    s = np.where(n > 0, np.sin(safe) / safe, 0.0)
    av = d * s[..., None]
    aw = np.where(n > 0, np.cos(n), 1.0)
    bv = q[..., :3]; bw = q[..., 3]
    out = np.empty_like(q)
    out[..., 3] = aw * bw - np.sum(av * bv, axis=-1)
    out[..., :3] = aw[..., None] * bv + bw[..., None] * av + np.cross(av, bv)
    ident = ~(n > 0)
    out[ident] = q[ident]
    return out


# ----------------------------------------------------------------------------
# evaluation: cost, robustified residuals and tangent Jacobian blocks
# ----------------------------------------------------------------------------
def wide(problem: Problem) -> bool:
    """bal9 mode: some camera keeps its intrinsics variable (cam_const bit 2) -> 9-wide camera blocks for the whole problem."""
    return bool(((problem.cam_const & INTR_VARIABLE) != 0).any())


def evaluate(problem: Problem, q, t, P, a=HUBER_A, want_jac=True, intr=None):
    is_wide = wide(problem)
    if want_jac and is_wide:
        r, valid, Jr, Jt, JP, Ji = project(problem, q, t, P, True, intr=intr, want_intr_jac=True)
    elif want_jac:
        r, valid, Jr, Jt, JP = project(problem, q, t, P, True, intr=intr)
    else:
        r, valid = project(problem, q, t, P, False, intr=intr)
    s = np.sum(r * r, axis=1)
    rho, rho1 = huber(s, a)
    cost = 0.5 * float(np.sum(rho))
    if not want_jac:
        return cost
    sw = np.sqrt(rho1)
    rt = r * sw[:, None]
    Fc = np.concatenate([Jr, Jt] + ([Ji] if is_wide else []), axis=2) * sw[:, None, None]   # [No,2,6] (rot, t) | [No,2,9] (+ f, k1, k2)
    Ep = JP * sw[:, None, None]                                 # [No,2,3]
    # constant blocks are removed from the program (A.4): zero their columns
    qc = (problem.cam_const[problem.obs_cam] & 1) != 0
    tc = (problem.cam_const[problem.obs_cam] & 2) != 0
    Fc[qc, :, 0:3] = 0.0
    Fc[tc, :, 3:6] = 0.0
    if is_wide:
        Fc[(problem.cam_const[problem.obs_cam] & INTR_VARIABLE) == 0, :, 6:9] = 0.0
    pc = problem.point_const[problem.obs_pt] != 0
    Ep[pc] = 0.0
    return cost, rt, Fc, Ep


def rmse_pair(problem: Problem, q=None, t=None, P=None, a=HUBER_A):
    """(reference-style sqrt(cost/num_residuals), plain sqrt(sum|r|^2/N_obs))."""
    r, _ = project(problem, q, t, P, False)
    s = np.sum(r * r, axis=1)
    rho, _ = huber(s, a)
    n = r.shape[0]
    return math.sqrt(0.5 * rho.sum() / (2 * n)), math.sqrt(s.sum() / n)


class _Linearization:
    """Normal-equation blocks at one point, in Jacobi-scaled coordinates."""

    def __init__(self, problem: Problem, rt, Fs, Es):
        self.rt, self.Fs, self.Es = rt, Fs, Es
        Nc, Np = problem.cam_q.shape[0], problem.points.shape[0]
        ci, pi = problem.obs_cam, problem.obs_pt
        W = Fs.shape[2]
        self.Hcc = np.zeros((Nc, W, W)); np.add.at(self.Hcc, ci, np.einsum("nki,nkj->nij", Fs, Fs))
        self.Hpp = np.zeros((Np, 3, 3)); np.add.at(self.Hpp, pi, np.einsum("nki,nkj->nij", Es, Es))
        self.gc = np.zeros((Nc, W)); np.add.at(self.gc, ci, np.einsum("nki,nk->ni", Fs, rt))
        self.gp = np.zeros((Np, 3)); np.add.at(self.gp, pi, np.einsum("nki,nk->ni", Es, rt))
        self.W = np.einsum("nki,nkj->nij", Fs, Es)          # [No,W,3] = F^T E


def _active_points(problem: Problem) -> np.ndarray:
    m = np.zeros(problem.points.shape[0], bool)
    m[problem.obs_pt] = True
    return m


def _solve_exact(problem: Problem, lin: _Linearization, Dc2, Dp2):
    """(Js^T Js + D^2) y = Js^T r via exact Schur complement (Appendix A.7)."""
    import scipy.linalg as sla
    Nc = problem.cam_q.shape[0]
    Wc = lin.Hcc.shape[1]                                  # camera block width: 6, or 9 in bal9 mode
    ci, pi = problem.obs_cam, problem.obs_pt
    Hpp_d = lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3))
    Hinv = np.linalg.inv(Hpp_d)
    # dense reduced camera matrix
    S = np.zeros((Nc * Wc, Nc * Wc))
    for c in range(Nc):
        S[Wc * c:Wc * c + Wc, Wc * c:Wc * c + Wc] = lin.Hcc[c] + np.diag(Dc2[c])
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])        # [No,6,3]
    b = lin.gc - _scatter_add(Nc, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    order = np.argsort(pi, kind="stable")
    ptr = np.searchsorted(pi[order], np.arange(problem.points.shape[0] + 1))
    # group tracks by length to vectorise the pair loop; duplicates are summed by coo->dense
    import scipy.sparse as sp
    lens = np.diff(ptr)
    rows_l, cols_l, vals_l = [], [], []
    i6 = np.arange(Wc)
    for L in np.unique(lens):
        if L == 0:
            continue
        pts = np.nonzero(lens == L)[0]
        idx = order[ptr[pts][:, None] + np.arange(L)[None, :]]     # [n,L] obs ids
        for a_ in range(L):
            for b_ in range(L):
                blk = np.einsum("nij,nkj->nik", WH[idx[:, a_]], lin.W[idx[:, b_]])  # [n,6,6]
                ca, cb = ci[idx[:, a_]], ci[idx[:, b_]]
                rows = np.broadcast_to(Wc * ca[:, None, None] + i6[None, :, None], blk.shape)
                cols = np.broadcast_to(Wc * cb[:, None, None] + i6[None, None, :], blk.shape)
                rows_l.append(rows.reshape(-1)); cols_l.append(cols.reshape(-1)); vals_l.append(blk.reshape(-1))
    if vals_l:
        S -= sp.coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))),
                           shape=S.shape).toarray()
    cf = sla.cho_factor(S, lower=True, check_finite=False)
    yc = sla.cho_solve(cf, b.reshape(-1), check_finite=False).reshape(Nc, Wc)
    yp = _back_substitute(problem, lin, Hinv, yc)
    return yc, yp, 0


def _scatter_add(n, idx, vals):
    out = np.zeros((n,) + vals.shape[1:])
    np.add.at(out, idx, vals)
    return out


def _back_substitute(problem, lin, Hinv, yc):
    pi, ci = problem.obs_pt, problem.obs_cam
    Np = problem.points.shape[0]
    Wty = _scatter_add(Np, pi, np.einsum("nij,ni->nj", lin.W, yc[ci]))
    return np.einsum("nij,nj->ni", Hinv, lin.gp - Wty)


def gauge_vectors(q, t, cam_const, sc_c):
    """The seven gauge directions of a reconstruction restricted to the cameras, in Jacobi-scaled tangent coordinates [Nc,6,7]:
    world X' = X + w x X + tau + sigma X  =>  camera (R, t): delta_q = -1/2 R w (Plus(q, d) = dq(d) * q), delta_t = sigma t - R tau;
    columns 0-2 translation, 3-5 rotation, 6 scale; rows of constant blocks zero (ba_kernels.h: k_pcg_gauge)."""
    R = rotation_from_quat(q)
    Nc = q.shape[0]
    W = np.zeros((Nc, 6, 7))
    for k in range(3):
        W[:, 3:6, k] = -R[:, :, k]
        W[:, 0:3, 3 + k] = -0.5 * R[:, :, k]
    W[:, 3:6, 6] = t
    W[(cam_const & 1) != 0, 0:3, :] = 0.0
    W[(cam_const & 2) != 0, 3:6, :] = 0.0
    return W / sc_c[:, :6, None]


def _solve_pcg(problem: Problem, lin: _Linearization, Dc2, Dp2, tol, max_iter, coarse=None):
    """Implicit-Schur PCG with block-Jacobi (6x6) preconditioner; coarse = (q, t, sc_c): + the gauge coarse space,
    M^-1 = blockdiag(S_cc)^-1 + W (W^T S W)^-1 W^T (the HIP path's two-level preconditioner).

    Same iteration the HIP path runs: x0 = 0, stop when
    |r|_2 <= tol * |b|_2.
    """
    Nc, Np = problem.cam_q.shape[0], problem.points.shape[0]
    ci, pi = problem.obs_cam, problem.obs_pt
    Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
    b = lin.gc - _scatter_add(Nc, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    Scc = lin.Hcc + np.einsum("ni,ij->nij", Dc2, np.eye(lin.Hcc.shape[1])) \
        - _scatter_add(Nc, ci, np.einsum("nij,nkj->nik", WH, lin.W))
    Minv = np.linalg.inv(Scc)

    def matvec(p):
        v = np.einsum("nki,ni->nk", lin.Fs, p[ci])
        tj = _scatter_add(Np, pi, np.einsum("nki,nk->ni", lin.Es, v))
        u = np.einsum("nij,nj->ni", Hinv, tj)
        zz = v - np.einsum("nki,ni->nk", lin.Es, u[pi])
        return Dc2 * p + _scatter_add(Nc, ci, np.einsum("nki,nk->ni", lin.Fs, zz))

    if coarse is not None:
        Wg = gauge_vectors(coarse[0], coarse[1], problem.cam_const, coarse[2])          # [Nc,6,7]
        SW = np.stack([matvec(Wg[:, :, g]) for g in range(7)], axis=2)
        E = np.einsum("nig,nih->gh", Wg, SW); E = 0.5 * (E + E.T)
        keep = np.diag(E) > 0
        Einv = np.zeros((7, 7))
        if keep.any():
            Einv[np.ix_(keep, keep)] = np.linalg.inv(E[np.ix_(keep, keep)])
        bj = Minv

        def precond(r_):
            return np.einsum("nij,nj->ni", bj, r_) + np.einsum("nig,g->ni", Wg, Einv @ np.einsum("nig,ni->g", Wg, r_))
    else:
        def precond(r_):
            return np.einsum("nij,nj->ni", Minv, r_)
    x = np.zeros((Nc, lin.Hcc.shape[1])); r = b.copy()
    z = precond(r); p = z.copy()
    rz = float(np.sum(r * z)); bnorm = float(np.linalg.norm(b)); it = 0
    if bnorm == 0.0:
        return x, _back_substitute(problem, lin, Hinv, x), 0
    while it < max_iter:
        if float(np.linalg.norm(r)) <= tol * bnorm:
            break
        qv = matvec(p)
        alpha = rz / float(np.sum(p * qv))
        x += alpha * p; r -= alpha * qv
        z = precond(r)
        rz_new = float(np.sum(r * z))
        p = z + (rz_new / rz) * p; rz = rz_new; it += 1
    return x, _back_substitute(problem, lin, Hinv, x), it


# ----------------------------------------------------------------------------
# trust-region loop (Appendix A.5 / A.6)
# ----------------------------------------------------------------------------
def solve(problem: Problem, opt: Optional[Options] = None) -> Summary:
    """Levenberg-Marquardt on `problem`, results written in place (like Ceres)."""
    opt = opt or Options()
    summ = Summary()
    q = problem.cam_q.copy(); t = problem.cam_t.copy(); P = problem.points.copy()
    intr = np.array(problem.intr_params, dtype=float, copy=True)      # state too in bal9 mode (intrinsics of cameras with bit 2)
    is_wide = wide(problem)
    ivar = (problem.cam_const & INTR_VARIABLE) != 0
    if is_wide:
        idx = problem.cam_intr[ivar]
        if (problem.intr_model[idx] != MODEL_BAL).any() or np.unique(problem.cam_intr, return_counts=True)[1][np.searchsorted(np.unique(problem.cam_intr), idx)].max() > 1:
            raise ValueError("variable intrinsics: model 5 with one intrinsics entry per camera")
    qvar = (problem.cam_const & 1) == 0
    tvar = (problem.cam_const & 2) == 0
    act = _active_points(problem)
    pvar = (problem.point_const == 0) & act
    # cameras without observations are not part of the program (ba_solver.cc:350-352)
    cam_act = np.zeros(q.shape[0], bool); cam_act[problem.obs_cam] = True
    qvar &= cam_act; tvar &= cam_act; ivar = ivar & cam_act
    summ.num_residuals = 2 * problem.obs_cam.shape[0]
    summ.num_effective_params = int(3 * qvar.sum() + 3 * tvar.sum() + 3 * pvar.sum() + 3 * ivar.sum())
    iv_rows = problem.cam_intr[ivar]             # rows of intr_params that are variable blocks {f, k1, k2}

    def x_norm(q_, t_, P_):
        i2 = float((intr[iv_rows, 0:3] ** 2).sum()) if is_wide else 0.0
        return math.sqrt(float((q_[qvar] ** 2).sum() + (t_[tvar] ** 2).sum() + (P_[pvar] ** 2).sum()) + i2)

    a = opt.huber_a
    alt = opt.alt
    if alt and alt not in ALT_DETAILS:
        raise ValueError(f"unknown sensitivity switch {alt!r}")
    min_rel_decrease = 1e-4 if alt == "min_relative_decrease_1e-4" else opt.min_relative_decrease
    min_lm_diag = 1e-9 if alt == "lm_diagonal_min_1e-9" else opt.min_lm_diagonal
    if alt == "x_norm_includes_constant_blocks":
        def x_norm(q_, t_, P_):          # noqa: F811
            return math.sqrt(float((q_[cam_act] ** 2).sum() + (t_[cam_act] ** 2).sum() + (P_[act] ** 2).sum()))
    cost, rt, Fc, Ep = evaluate(problem, q, t, P, a, intr=intr)
    summ.initial_cost = cost
    # Jacobi scaling, computed once at iteration 0: 1/(1+||col||)
    ci, pi = problem.obs_cam, problem.obs_pt
    Fn, En = Fc, Ep
    if alt == "jacobi_scaling_unrobustified":
        r_, _valid, Jr_, Jt_, JP_ = project(problem, q, t, P, True, intr=intr)
        Fn = np.concatenate([Jr_, Jt_], axis=2) * (Fc != 0).any(axis=1, keepdims=True)     # constant blocks stay zero
        En = JP_ * (Ep != 0).any(axis=1, keepdims=True)
    cn_c = np.sqrt(_scatter_add(q.shape[0], ci, np.sum(Fn * Fn, axis=1)))
    cn_p = np.sqrt(_scatter_add(P.shape[0], pi, np.sum(En * En, axis=1)))
    sc_c = 1.0 / (1.0 + cn_c); sc_p = 1.0 / (1.0 + cn_p)
    if alt == "jacobi_scaling_off":
        sc_c = np.ones_like(cn_c); sc_p = np.ones_like(cn_p)
    elif alt == "jacobi_scaling_plain_inverse":
        sc_c = 1.0 / np.where(cn_c > 0, cn_c, 1.0); sc_p = 1.0 / np.where(cn_p > 0, cn_p, 1.0)

    def linearize(rt_, Fc_, Ep_):
        return _Linearization(problem, rt_, Fc_ * sc_c[ci][:, None, :], Ep_ * sc_p[pi][:, None, :])

    def grad_max(lin_, q_, t_, P_):
        # Ceres: |x - Plus(x, -g)|_inf with the unscaled gradient
        gc = lin_.gc / sc_c; gp = lin_.gp / sc_p
        m = 0.0
        if qvar.any() and alt == "gradient_norm_plain":
            m = max(m, float(np.abs(gc[qvar, 0:3]).max()))
        elif qvar.any():
            m = max(m, float(np.abs(q_[qvar] - quat_plus(q_[qvar], -gc[qvar, 0:3])).max()))
        if tvar.any():
            m = max(m, float(np.abs(gc[tvar, 3:6]).max()))
        if is_wide and ivar.any():
            m = max(m, float(np.abs(gc[ivar, 6:9]).max()))
        if pvar.any():
            m = max(m, float(np.abs(gp[pvar]).max()))
        return m

    lin = linearize(rt, Fc, Ep)
    summ.trace.append(dict(it=0, cost=cost, radius=opt.initial_radius, ok=True))

    def finish(term, cost_):
        summ.termination = term
        summ.final_cost = cost_
        if alt == "iteration_zero_counted":
            summ.n_successful += 1       # (a count only: which of the two counters Ceres would use is part of what is unpinned)
        problem.cam_q[:] = q; problem.cam_t[:] = t; problem.points[:] = P
        if is_wide:
            problem.intr_params[:] = intr
        return summ

    if grad_max(lin, q, t, P) <= opt.gradient_tolerance:
        return finish("CONVERGENCE: gradient tolerance", cost)

    radius = opt.initial_radius
    decrease = 2.0
    xn = x_norm(q, t, P)
    it = 0
    invalid = 0
    while True:
        if it >= opt.max_iterations:
            return finish("NO_CONVERGENCE: max iterations", cost)
        it += 1
        diag_c = np.clip(np.einsum("nii->ni", lin.Hcc), min_lm_diag, opt.max_lm_diagonal)
        diag_p = np.clip(np.einsum("nii->ni", lin.Hpp), min_lm_diag, opt.max_lm_diagonal)
        if alt == "lm_diagonal_unclamped":
            diag_c = np.einsum("nii->ni", lin.Hcc).copy(); diag_p = np.einsum("nii->ni", lin.Hpp).copy()
            diag_c[diag_c == 0] = 1e-300; diag_p[diag_p == 0] = 1e-300      # (constant blocks: keep the system non-singular)
        Dc2 = diag_c / radius; Dp2 = diag_p / radius
        if opt.linear_solver == "exact":
            yc, yp, k = _solve_exact(problem, lin, Dc2, Dp2)
        else:
            yc, yp, k = _solve_pcg(problem, lin, Dc2, Dp2, opt.pcg_tol, opt.pcg_max_iter,
                                   coarse=(q, t, sc_c) if (opt.pcg_coarse and not is_wide) else None)
        summ.pcg_iterations += k
        ok = np.isfinite(yc).all() and np.isfinite(yp).all()
        # step = -y; model residual m = Js*step
        mres = -(np.einsum("nki,ni->nk", lin.Fs, yc[ci]) + np.einsum("nki,ni->nk", lin.Es, yp[pi]))
        model_change = -float(np.sum(mres * (lin.rt + 0.5 * mres)))
        if ((not ok) or not (model_change > 0.0)) and alt == "invalid_step_is_plain_reject":
            radius /= decrease; decrease *= 2.0
            summ.n_unsuccessful += 1
            summ.trace.append(dict(it=it, cost=cost, radius=radius, ok=False, invalid=True))
            if radius < opt.min_radius:
                return finish("CONVERGENCE: min trust region radius", cost)
            continue
        if (not ok) or not (model_change > 0.0):
            invalid += 1
            summ.trace.append(dict(it=it, cost=cost, radius=radius, ok=False, invalid=True))
            if invalid >= opt.max_consecutive_invalid_steps:
                return finish("FAILURE: too many invalid steps", cost)
            radius /= decrease; decrease *= 2.0
            summ.n_unsuccessful += 1
            continue
        invalid = 0
        dc = -yc * sc_c; dp = -yp * sc_p
        dc[~qvar, 0:3] = 0.0; dc[~tvar, 3:6] = 0.0; dp[~pvar] = 0.0
        q2 = q.copy(); q2[qvar] = quat_plus(q[qvar], dc[qvar, 0:3])
        t2 = t + dc[:, 3:6]
        P2 = P + dp
        intr2 = intr
        if is_wide:
            dc[~ivar, 6:9] = 0.0
            intr2 = intr.copy(); intr2[iv_rows, 0:3] = intr[iv_rows, 0:3] + dc[ivar, 6:9]
        cost2 = evaluate(problem, q2, t2, P2, a, want_jac=False, intr=intr2)
        step_norm = math.sqrt(float(((q2 - q)[qvar] ** 2).sum() + ((t2 - t)[tvar] ** 2).sum()
                                    + ((P2 - P)[pvar] ** 2).sum() + ((intr2 - intr) ** 2).sum()))
        cost_change = cost - cost2
        rel = cost_change / model_change

        def tolerance_exit(kind):
            summ.trace.append(dict(it=it, cost=cost2, radius=radius, ok=None, step_norm=step_norm))
            if alt == "accept_before_tolerance" and rel > min_rel_decrease:
                nonlocal q, t, P
                q, t, P = q2, t2, P2
                intr[:] = intr2
                summ.n_successful += 1
                return finish("CONVERGENCE: " + kind, cost2)
            return finish("CONVERGENCE: " + kind, cost)

        if step_norm <= opt.parameter_tolerance * (xn + opt.parameter_tolerance):
            return tolerance_exit("parameter tolerance")
        if abs(cost_change) <= opt.function_tolerance * (cost2 if alt == "function_tolerance_vs_candidate_cost" else cost):
            return tolerance_exit("function tolerance")
        if rel > min_rel_decrease:
            q, t, P = q2, t2, P2
            intr = intr2
            xn = x_norm(q, t, P)
            cost, rt, Fc, Ep = evaluate(problem, q, t, P, a, intr=intr)
            lin = linearize(rt, Fc, Ep)
            radius = min(opt.max_radius, radius / max(0.5 if alt == "radius_factor_capped_at_2" else 1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            decrease = 2.0
            summ.n_successful += 1
            summ.trace.append(dict(it=it, cost=cost, radius=radius, ok=True, step_norm=step_norm,
                                   rel=rel, model_change=model_change))
            if grad_max(lin, q, t, P) <= opt.gradient_tolerance:
                return finish("CONVERGENCE: gradient tolerance", cost)
        else:
            radius /= decrease; decrease *= 2.0
            if alt == "radius_halving_on_reject":
                decrease = 2.0
            summ.n_unsuccessful += 1
            summ.trace.append(dict(it=it, cost=cost, radius=radius, ok=False, step_norm=step_norm,
                                   rel=rel, model_change=model_change))
            if radius < opt.min_radius:
                return finish("CONVERGENCE: min trust region radius", cost)


# ----------------------------------------------------------------------------
# post-BA track filter
# ----------------------------------------------------------------------------
def filter_tracks(problem: Problem, max_re: float, min_angle: float):
    """Restates FilterPoints3d/FilterPoint3d/UpdateTrackAngle (/root/reference/src/geometry/track_processor.cc:253-332)
    and colmap::CalculateTriangulationAngle (geometry/colmap/base/triangulation.cc:124-147).  Pure-Python loops."""
    ci, pi = problem.obs_cam, problem.obs_pt
    M = rotation_from_quat(problem.cam_q)
    centre = -np.einsum("nji,nj->ni", M, problem.cam_t)
    Pc = np.einsum("nij,nj->ni", M[ci], problem.points[pi]) + problem.cam_t[ci]
    z = Pc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        xn = Pc[:, 0] / z; yn = Pc[:, 1] / z
    intr = problem.cam_intr[ci]; model = problem.intr_model[intr]; prm = problem.intr_params[intr]
    fx, fy, cx, cy = _intrinsics_per_obs(model, prm)
    r2 = xn * xn + yn * yn
    du = np.zeros_like(xn); dv = np.zeros_like(xn)
    m = (model == 0) | (model == 1); du[m] = xn[m]; dv[m] = yn[m]
    for mid, ik in ((2, 3), (3, 4)):
        m = model == mid; du[m] = xn[m] * (prm[m, ik] * r2[m]); dv[m] = yn[m] * (prm[m, ik] * r2[m])
    m = model == 4
    if m.any():
        k1, k2, p1, p2 = prm[m, 4], prm[m, 5], prm[m, 6], prm[m, 7]
        rad = k1 * r2[m] + k2 * r2[m] ** 2; xy = xn[m] * yn[m]
        du[m] = xn[m] * rad + 2 * p1 * xy + p2 * (r2[m] + 2 * xn[m] ** 2)
        dv[m] = yn[m] * rad + 2 * p2 * xy + p1 * (r2[m] + 2 * yn[m] ** 2)
    re = np.hypot(fx * (xn + du) + cx - problem.obs_uv[:, 0], fy * (yn + dv) + cy - problem.obs_uv[:, 1])
    delete = (re > max_re) | (z < 1e-3) | (z > 1e3)
    n_p = problem.points.shape[0]
    outlier = np.zeros(n_p, np.uint8); err = np.full(n_p, -1.0); ang = np.full(n_p, -1.0); cnt = [0, 0]
    order = np.lexsort((ci, pi)); ptr = np.searchsorted(pi[order], np.arange(n_p + 1))
    for j in range(n_p):
        ids = order[ptr[j]:ptr[j + 1]]
        n = len(ids)
        if n == 0:
            continue
        nd = int(delete[ids].sum())
        if nd >= n - 1:
            outlier[j] = 1; cnt[0] += n; continue
        cnt[0] += nd
        keep = ids[~delete[ids]]
        err[j] = re[keep].sum() / len(keep)
        best = 0.0; done = False
        P = problem.points[j]
        for a in range(len(keep)):
            if done:
                break
            for b in range(a + 1, len(keep)):
                c1, c2 = centre[ci[keep[a]]], centre[ci[keep[b]]]
                b2 = ((c1 - c2) ** 2).sum(); r1 = ((P - c1) ** 2).sum(); r2_ = ((P - c2) ** 2).sum()
                den = 2.0 * math.sqrt(r1 * r2_)
                t = 0.0 if den == 0.0 else abs(math.acos(max(-1.0, min(1.0, (r1 + r2_ - b2) / den))))
                t = min(t, math.pi - t)
                if t > best:
                    best = t
                    if best > min_angle:
                        done = True; break
        ang[j] = best
        if best < min_angle:
            outlier[j] = 2; cnt[1] += 1
    return dict(obs_delete=delete.astype(np.uint8), track_outlier=outlier, track_error=err, track_angle=ang, num_filtered=np.array(cnt))
