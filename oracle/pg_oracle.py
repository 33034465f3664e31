"""TEST INFRASTRUCTURE — CPU restatement of the scaled pose graph of BASolver::ScalePoseGraphUnorder.

Residuals restated from /root/reference/src/optimization/cost_factor_ceres.h:117-221 (PoseGraphCost, ScaleCost) and
lie_algebra.h:12-15 (logmap) in numpy; the minimiser is scipy.optimize.least_squares (trust-region reflective with the
reference's lower bound 0.2 on the scales), i.e. an implementation independent of xrsfm_amd/csrc/pose_graph.h.  Ceres is not
available here: PARITY UNPINNED against the real DOGLEG trajectory; what is pinned is the minimum.
Only tests/ may import this module.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


def _rot(q):                      # x,y,z,w -> matrix
    return Rotation.from_quat(q).as_matrix()


def residuals(rot_q, pos, scale, edges, weight_o, scale_costs):
    """All residual rows in Ceres' order per block: [rot(3), s1/s2-1, prior, pos(3)] per edge, then the scale costs."""
    out = []
    for e in range(len(edges["a"])):
        a, b, sa, sb = edges["a"][e], edges["b"][e], edges["sa"][e], edges["sb"][e]
        R1, R2 = _rot(rot_q[a]), _rot(rot_q[b])
        s1, s2 = scale[sa], scale[sb]
        R12 = R1.T @ R2
        Rm = _rot(edges["q_mea"][e])
        r_rot = Rotation.from_matrix(Rm @ R12.T).as_rotvec()
        p12 = R1.T @ (pos[b] - pos[a])
        prior = weight_o * (s1 - 1) if s1 < 1 else weight_o * (1.0 / s1 - 1)
        out.append(np.concatenate([r_rot, [s1 / s2 - 1.0, prior], p12 - s1 * np.asarray(edges["p_mea"][e])]))
    for sa, sb, s12 in scale_costs:
        out.append(np.array([10.0 * (scale[sa] / (s12 * scale[sb]) - 1.0)]))
    return np.concatenate(out) if out else np.zeros(0)


def solve(rot_q, pos, scale, edges, weight_o=0.0, scale_costs=(), pos_const=None, scale_const=None, scale_lower=None):
    rot_q = np.asarray(rot_q, float); pos = np.array(pos, float); scale = np.array(scale, float)
    n, m = pos.shape[0], scale.shape[0]
    pc = np.zeros(n, bool) if pos_const is None else np.asarray(pos_const, bool)
    scn = np.zeros(m, bool) if scale_const is None else np.asarray(scale_const, bool)
    used_p = np.zeros(n, bool); used_s = np.zeros(m, bool)
    used_p[list(edges["a"])] = True; used_p[list(edges["b"])] = True
    used_s[list(edges["sa"])] = True; used_s[list(edges["sb"])] = True
    for sa, sb, _ in scale_costs:
        used_s[sa] = used_s[sb] = True
    ip = np.nonzero(used_p & ~pc)[0]; isc = np.nonzero(used_s & ~scn)[0]

    def unpack(x):
        p2, s2 = pos.copy(), scale.copy()
        p2[ip] = x[:3 * len(ip)].reshape(-1, 3); s2[isc] = x[3 * len(ip):]
        return p2, s2

    def fun(x):
        p2, s2 = unpack(x)
        return residuals(rot_q, p2, s2, edges, weight_o, scale_costs)

    x0 = np.concatenate([pos[ip].ravel(), scale[isc]])
    lo = np.full(x0.shape, -np.inf)
    if scale_lower is not None:
        lo[3 * len(ip):] = np.asarray(scale_lower, float)[isc]
        x0 = np.maximum(x0, lo + 1e-12)
    res = least_squares(fun, x0, bounds=(lo, np.full(x0.shape, np.inf)), method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=2000)
    p2, s2 = unpack(res.x)
    return p2, s2, 0.5 * float(np.sum(res.fun ** 2)), 0.5 * float(np.sum(fun(x0) ** 2))
