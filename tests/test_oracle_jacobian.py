"""Pins the oracle's residual and closed-form Jacobians to the REFERENCE functor.

The residual below is a line-by-line torch restatement of ReProjectionCost::operator()
(/root/reference/src/optimization/cost_factor_ceres.h:19-40: qcw*pw+tcw, z<1e-2 clamp, hnormalized,
WorldToImage) with the five Distortion functions of /root/reference/src/base/camera_model.hpp:93-209 and
EigenQuaternionParameterization::Plus; torch.autograd (FP64) then plays the role of Ceres' autodiff Jets.
"""
import numpy as np
import pytest
import torch

from oracle import ba_oracle as bo
from tests import helpers as H


def _ref_residual(q, t, P, model, k, uv):
    """cost_factor_ceres.h:19-40 with Eigen's quaternion * vector (no normalisation), q = (x,y,z,w)."""
    u, w = q[:3], q[3]
    a2 = 2.0 * torch.linalg.cross(u, P)
    pc = P + w * a2 + torch.linalg.cross(u, a2) + t
    if pc[2] < 1e-2:
        return torch.tensor([12.0, 12.0], dtype=torch.float64) + 0.0 * pc[:2]
    xy = pc[:2] / pc[2]
    x, y = xy[0], xy[1]
    if model == 0:
        fx, fy, cx, cy = k[0], k[0], k[1], k[2]; du, dv = x, y
    elif model == 1:
        fx, fy, cx, cy = k[0], k[1], k[2], k[3]; du, dv = x, y
    elif model == 2:
        fx, fy, cx, cy = k[0], k[0], k[1], k[2]; rad = k[3] * (x * x + y * y); du, dv = x * rad, y * rad
    elif model == 3:
        fx, fy, cx, cy = k[0], k[1], k[2], k[3]; rad = k[4] * (x * x + y * y); du, dv = x * rad, y * rad
    elif model == 5:        # extension (not a reference model): BAL-style {f, k1, k2}, no principal point — bal9 mode
        fx, fy, cx, cy = k[0], k[0], 0.0, 0.0; r2 = x * x + y * y; rad = k[1] * r2 + k[2] * r2 * r2; du, dv = x * rad, y * rad
    else:
        fx, fy, cx, cy = k[0], k[1], k[2], k[3]; k1, k2, p1, p2 = k[4], k[5], k[6], k[7]
        x2, xy_, y2 = x * x, x * y, y * y; r2 = x2 + y2; rad = k1 * r2 + k2 * r2 * r2
        du = x * rad + 2 * p1 * xy_ + p2 * (r2 + 2 * x2); dv = y * rad + 2 * p2 * xy_ + p1 * (r2 + 2 * y2)
    return torch.stack([fx * (x + du) + cx - uv[0], fy * (y + dv) + cy - uv[1]])


def _plus(q, d):
    n = torch.sqrt((d * d).sum())
    s = torch.sin(n) / n
    a = torch.cat([s * d, torch.cos(n).reshape(1)])
    av, aw, bv, bw = a[:3], a[3], q[:3], q[3]
    return torch.cat([aw * bv + bw * av + torch.linalg.cross(av, bv), (aw * bw - (av * bv).sum()).reshape(1)])


@pytest.mark.parametrize("behind", [False, True])
def test_jacobians_match_autodiff_of_reference_functor(behind):
    arr = H.with_models(H.make(10, 40, 4, seed=110), seed=6)
    if behind:
        arr["points"][::5] += np.array([0.0, 0.0, -90.0])
    pr = H.to_oracle(arr)
    r, valid, Jr, Jt, JP = bo.project(pr)
    assert valid.all() != behind
    for i in range(0, pr.obs_cam.shape[0], 3):
        c, p = pr.obs_cam[i], pr.obs_pt[i]
        ii = pr.cam_intr[c]
        q = torch.tensor(pr.cam_q[c]); t = torch.tensor(pr.cam_t[c]); P = torch.tensor(pr.points[p])
        k = torch.tensor(pr.intr_params[ii]); uv = torch.tensor(pr.obs_uv[i]); model = int(pr.intr_model[ii])
        # Ceres: autodiff Jacobian w.r.t. the AMBIENT quaternion (2x4) times the parameterisation's analytic
        # 4x3 ComputeJacobian (EigenQuaternionParameterization, rows x,y,z,w)
        f = lambda qq, tt, PP: _ref_residual(qq, tt, PP, model, k, uv)
        Jq, Jtt, JPP = torch.autograd.functional.jacobian(f, (q, t, P))
        x, y, z, w = [float(v) for v in q]
        plusJ = torch.tensor([[w, z, -y], [-z, w, x], [y, -x, w], [-x, -y, -z]], dtype=torch.float64)
        Jd = Jq @ plusJ
        r_ref = _ref_residual(q, t, P, model, k, uv).numpy()
        assert np.abs(r[i] - r_ref).max() <= 1e-12 * max(1.0, np.abs(r_ref).max())
        scale = max(1.0, float(Jd.abs().max()), float(JPP.abs().max()))
        assert np.abs(Jr[i] - Jd.numpy()).max() <= 1e-12 * scale
        assert np.abs(Jt[i] - Jtt.numpy()).max() <= 1e-12 * scale
        assert np.abs(JP[i] - JPP.numpy()).max() <= 1e-12 * scale


def test_huber_and_clamp_constants():
    rho, rho1 = bo.huber(np.array([0.0, 35.0, 35.8801, 36.0, 288.0]))
    assert rho[0] == 0 and rho1[0] == 1 and rho1[2] == 1
    assert abs(rho[4] - (2 * 5.99 * np.sqrt(288.0) - 5.99 ** 2)) < 1e-12     # clamp residual (12,12): rho = 167.43
    assert abs(rho1[3] - 5.99 / 6.0) < 1e-15


def test_quaternion_plus_is_full_angle_left_multiplication():
    q = np.array([[0.1, -0.2, 0.3, 0.9]]); q /= np.linalg.norm(q)
    d = np.array([[0.02, -0.01, 0.03]])
    out = bo.quat_plus(q, d)[0]
    n = np.linalg.norm(d)
    dq = np.concatenate([np.sin(n) / n * d[0], [np.cos(n)]])      # angle n, not n/2
    w = dq[3] * q[0, 3] - dq[:3] @ q[0, :3]
    v = dq[3] * q[0, :3] + q[0, 3] * dq[:3] + np.cross(dq[:3], q[0, :3])
    assert np.abs(out - np.concatenate([v, [w]])).max() < 1e-16
    assert abs(np.linalg.norm(out) - 1) < 1e-15
    assert np.array_equal(bo.quat_plus(q, np.zeros((1, 3))), q)


def test_plus_jacobian_matrix_is_the_derivative_of_plus():
    """The 4x3 matrix used above equals d Plus(q, delta) / d delta at 0 (central differences)."""
    q = np.array([0.3, -0.1, 0.2, 0.9]); q /= np.linalg.norm(q)
    x, y, z, w = q
    plusJ = np.array([[w, z, -y], [-z, w, x], [y, -x, w], [-x, -y, -z]])
    num = np.zeros((4, 3))
    h = 1e-6
    for a in range(3):
        d = np.zeros((1, 3)); d[0, a] = h
        num[:, a] = (bo.quat_plus(q[None], d)[0] - bo.quat_plus(q[None], -d)[0]) / (2 * h)
    assert np.abs(num - plusJ).max() < 1e-9


def test_bal9_jacobians_match_autodiff():
    """Extension model 5 {f, k1, k2} with variable intrinsics (bal9 mode): residual and all four Jacobian blocks — rotation
    (through the quaternion parameterisation), translation, point, and d r / d (f, k1, k2) — against torch.autograd."""
    arr = H.make_bal9(8, 60, 4, seed=111)
    arr["intr_params"][:, 1] = np.linspace(-0.05, 0.08, 8); arr["intr_params"][:, 2] = np.linspace(0.03, -0.04, 8)
    pr = H.to_oracle(arr)
    r, valid, Jr, Jt, JP, Ji = bo.project(pr, want_intr_jac=True)
    assert valid.all()
    for i in range(0, pr.obs_cam.shape[0], 2):
        c, p = pr.obs_cam[i], pr.obs_pt[i]
        ii = pr.cam_intr[c]
        q = torch.tensor(pr.cam_q[c]); t = torch.tensor(pr.cam_t[c]); P = torch.tensor(pr.points[p])
        k = torch.tensor(pr.intr_params[ii]); uv = torch.tensor(pr.obs_uv[i])
        f = lambda qq, tt, PP, kk: _ref_residual(qq, tt, PP, 5, kk, uv)
        Jq, Jtt, JPP, Jk = torch.autograd.functional.jacobian(f, (q, t, P, k))
        x, y, z, w = [float(v) for v in q]
        plusJ = torch.tensor([[w, z, -y], [-z, w, x], [y, -x, w], [-x, -y, -z]], dtype=torch.float64)
        Jd = Jq @ plusJ
        scale = max(1.0, float(Jd.abs().max()), float(JPP.abs().max()), float(Jk.abs().max()))
        assert np.abs(r[i] - f(q, t, P, k).numpy()).max() <= 1e-12 * max(1.0, np.abs(r[i]).max())
        assert np.abs(Jr[i] - Jd.numpy()).max() <= 1e-12 * scale and np.abs(Jt[i] - Jtt.numpy()).max() <= 1e-12 * scale
        assert np.abs(JP[i] - JPP.numpy()).max() <= 1e-12 * scale
        assert np.abs(Ji[i] - Jk.numpy()[:, :3]).max() <= 1e-12 * scale and np.abs(Jk.numpy()[:, 3:]).max() == 0.0


def test_bal9_oracle_solve_converges_and_moves_intrinsics():
    arr = H.make_bal9(12, 600, 4, seed=5)
    pr = H.to_oracle(arr)
    s = bo.solve(pr, bo.Options())
    n = arr["obs_cam"].shape[0]
    assert s.num_effective_params == 3 * 12 + 3 * 10 + 3 * 600 + 3 * 12          # rotations, translations (two fixed), points, intrinsics
    assert s.final_cost < 0.05 * s.initial_cost and np.sqrt(s.final_cost / n) < 2.0
    assert np.abs(pr.intr_params[:, :3] - arr["intr_params"][:, :3]).max() > 1e-3     # the intrinsics took part
    assert np.array_equal(pr.intr_params[:, 3:], arr["intr_params"][:, 3:])
    # with bit 2 clear the same problem is the ordinary 6-wide one: intrinsics untouched
    arr6 = dict(arr); arr6["cam_const"] = (arr["cam_const"] & 3).astype(np.uint8)
    pr6 = H.to_oracle(arr6)
    s6 = bo.solve(pr6, bo.Options())
    assert np.array_equal(pr6.intr_params, arr["intr_params"]) and s6.num_effective_params == s.num_effective_params - 36
