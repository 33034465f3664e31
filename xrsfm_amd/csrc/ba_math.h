// Device math for the reprojection residual family of XRSfM's BA.
//
// Model (what Ceres autodiff evaluates for the reference functor):
//   residual        /root/reference/src/optimization/cost_factor_ceres.h:19-40
//   WorldToImage    /root/reference/src/base/camera_model.hpp:57-68
//   distortions     /root/reference/src/base/camera_model.hpp:93-209
//   Huber(5.99)     /root/reference/src/optimization/ba_solver.cc:343,374
//   quaternion plus ceres::EigenQuaternionParameterization (ba_solver.cc:353-354)
// Closed-form Jacobians: SURVEY.md Appendix A.2.  Everything is FP64.
#pragma once
#include <hip/hip_runtime.h>

namespace xba {

// One camera as laid out in HBM: a 128-byte record so a gather touches one line.
struct __attribute__((aligned(16))) CamRec {
    double q[4];     // x,y,z,w
    double t[3];
    double pad;
    double intr[8];  // camera.params_, zero padded
};
static_assert(sizeof(CamRec) == 128, "CamRec must be 128 bytes");

constexpr double kMinDepth = 1e-2;   // cost_factor_ceres.h:29
constexpr double kClampRes = 12.0;   // cost_factor_ceres.h:31

struct Proj {
    double r0, r1;        // residual (not robustified)
    double jp[6];         // d r / d Pc, 2x3 row-major (zero in the clamp case)
    double rp[3];         // M(q) * P  (= Pc - t)
    double ji[6];         // d r / d (f, k1, k2), 2x3 row-major: extension model 5 only (bal9 mode), project<true, true>
    bool clamped;
};

__device__ __forceinline__ void quat_to_mat(const double q[4], double M[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    M[0] = 1.0 - 2.0 * (y * y + z * z); M[1] = 2.0 * (x * y - w * z); M[2] = 2.0 * (x * z + w * y);
    M[3] = 2.0 * (x * y + w * z); M[4] = 1.0 - 2.0 * (x * x + z * z); M[5] = 2.0 * (y * z - w * x);
    M[6] = 2.0 * (x * z - w * y); M[7] = 2.0 * (y * z + w * x); M[8] = 1.0 - 2.0 * (x * x + y * y);
}

// Residual (and optionally d r/d Pc) for one observation.
// kIntr: also d r / d (f, k1, k2) of the extension model 5 (zero for the reference's models, whose intrinsics are constant)
template <bool kJac, bool kIntr = false>
__device__ __forceinline__ void project(const double M[9], const double t[3], const double* __restrict__ k,
                                        int model, const double P[3], double u_obs, double v_obs, Proj& o) {
    o.rp[0] = M[0] * P[0] + M[1] * P[1] + M[2] * P[2];
    o.rp[1] = M[3] * P[0] + M[4] * P[1] + M[5] * P[2];
    o.rp[2] = M[6] * P[0] + M[7] * P[1] + M[8] * P[2];
    const double X = o.rp[0] + t[0], Y = o.rp[1] + t[1], Z = o.rp[2] + t[2];
    if (Z < kMinDepth) {
        o.r0 = kClampRes; o.r1 = kClampRes; o.clamped = true;
        if (kJac) { for (int i = 0; i < 6; ++i) o.jp[i] = 0.0; }
        if (kIntr) { for (int i = 0; i < 6; ++i) o.ji[i] = 0.0; }
        return;
    }
    o.clamped = false;
    const double iz = 1.0 / Z;
    const double xn = X * iz, yn = Y * iz;
    double fx, fy, cx, cy, du, dv, D00 = 1.0, D01 = 0.0, D10 = 0.0, D11 = 1.0;
    const double r2 = xn * xn + yn * yn;
    switch (model) {
    case 0:  // SIMPLE_PINHOLE: duv = xy (reference quirk)
        fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; du = xn; dv = yn; D00 = 2.0; D11 = 2.0; break;
    case 1:  // PINHOLE: same quirk
        fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; du = xn; dv = yn; D00 = 2.0; D11 = 2.0; break;
    case 2: case 3: {  // SIMPLE_RADIAL {f,cx,cy,k} / RADIAL {fx,fy,cx,cy,k}
        double kk;
        if (model == 2) { fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; kk = k[3]; }
        else { fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; kk = k[4]; }
        const double rad = kk * r2;
        du = xn * rad; dv = yn * rad;
        if (kJac) {
            D00 = 1.0 + rad + 2.0 * kk * xn * xn; D11 = 1.0 + rad + 2.0 * kk * yn * yn;
            D01 = 2.0 * kk * xn * yn; D10 = D01;
        }
        break; }
    case 5: {   // extension, not a reference model: BAL-style {f, k1, k2}, no principal point (bal9 mode)
        fx = k[0]; fy = k[0]; cx = 0.0; cy = 0.0;
        const double k1 = k[1], k2 = k[2];
        const double rad = k1 * r2 + k2 * r2 * r2;
        du = xn * rad; dv = yn * rad;
        if (kJac) {
            const double rad_x = 2.0 * k1 * xn + 4.0 * k2 * r2 * xn;
            const double rad_y = 2.0 * k1 * yn + 4.0 * k2 * r2 * yn;
            D00 = 1.0 + rad + xn * rad_x; D01 = xn * rad_y;
            D10 = yn * rad_x; D11 = 1.0 + rad + yn * rad_y;
        }
        break; }
    default: {  // OPENCV
        fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
        const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7];
        const double xy = xn * yn, x2 = xn * xn, y2 = yn * yn;
        const double rad = k1 * r2 + k2 * r2 * r2;
        du = xn * rad + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2);
        dv = yn * rad + 2.0 * p2 * xy + p1 * (r2 + 2.0 * y2);
        if (kJac) {
            const double rad_x = 2.0 * k1 * xn + 4.0 * k2 * r2 * xn;
            const double rad_y = 2.0 * k1 * yn + 4.0 * k2 * r2 * yn;
            D00 = 1.0 + rad + xn * rad_x + 2.0 * p1 * yn + 6.0 * p2 * xn;
            D01 = xn * rad_y + 2.0 * p1 * xn + 2.0 * p2 * yn;
            D10 = yn * rad_x + 2.0 * p2 * yn + 2.0 * p1 * xn;
            D11 = 1.0 + rad + yn * rad_y + 2.0 * p2 * xn + 6.0 * p1 * yn;
        }
        break; }
    }
    o.r0 = fx * (xn + du) + cx - u_obs;
    o.r1 = fy * (yn + dv) + cy - v_obs;
    if (kIntr) {
        if (model == 5) {
            o.ji[0] = xn + du; o.ji[1] = fx * xn * r2; o.ji[2] = fx * xn * r2 * r2;
            o.ji[3] = yn + dv; o.ji[4] = fx * yn * r2; o.ji[5] = fx * yn * r2 * r2;
        } else {
            for (int i = 0; i < 6; ++i) o.ji[i] = 0.0;
        }
    }
    if (kJac) {
        const double A00 = fx * D00, A01 = fx * D01, A10 = fy * D10, A11 = fy * D11;
        o.jp[0] = A00 * iz; o.jp[1] = A01 * iz; o.jp[2] = -(A00 * xn + A01 * yn) * iz;
        o.jp[3] = A10 * iz; o.jp[4] = A11 * iz; o.jp[5] = -(A10 * xn + A11 * yn) * iz;
    }
}

// ceres::HuberLoss(a) on s = |r|^2: returns rho, sets rho1 = rho'.
__device__ __forceinline__ double huber(double s, double a, double& rho1) {
    const double b = a * a;
    if (s > b) {
        const double r = sqrt(s);
        rho1 = fmax(2.2250738585072014e-308, a / r);
        return 2.0 * a * r - b;
    }
    rho1 = 1.0;
    return s;
}

// EigenQuaternionParameterization::Plus (full angle, left multiplication), q = xyzw.
__device__ __forceinline__ void quat_plus(const double q[4], const double d[3], double out[4]) {
    // (computed unconditionally, selected at the end: with the two branches writing through `out` the compiler kept the
    //  caller's array in scratch memory)
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const bool nz = n > 0.0;
    const double s = sin(n) / (nz ? n : 1.0);
    const double ax = s * d[0], ay = s * d[1], az = s * d[2], aw = cos(n);
    const double bx = q[0], by = q[1], bz = q[2], bw = q[3];
    const double ow = aw * bw - (ax * bx + ay * by + az * bz);
    const double ox = aw * bx + bw * ax + (ay * bz - az * by);
    const double oy = aw * by + bw * ay + (az * bx - ax * bz);
    const double oz = aw * bz + bw * az + (ax * by - ay * bx);
    out[0] = nz ? ox : bx; out[1] = nz ? oy : by; out[2] = nz ? oz : bz; out[3] = nz ? ow : bw;
}

// Inverse of a symmetric positive definite 3x3 given as upper triangle {00,01,02,11,12,22}.
__device__ __forceinline__ void sym3_inverse(const double h[6], double inv[6]) {
    const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = (a * f - c * c) * id; inv[4] = (b * c - a * e) * id; inv[5] = (a * d - b * b) * id;
}

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }

__device__ __forceinline__ double fast_rcp(double u) {               // v_rcp_f64 + 2 Newton steps
    double r = __builtin_amdgcn_rcp(u);
    double e = fma(-u, r, 1.0); r = fma(r, e, r);
    e = fma(-u, r, 1.0); r = fma(r, e, r);
    return r;
}
__device__ __forceinline__ double fast_rsqrt(double u) {             // v_rsq_f64 + 2 Newton steps
    double y = __builtin_amdgcn_rsq(u);
    double h = 0.5 * u;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// LM damping of Ceres' LevenbergMarquardtStrategy: D^2 = clamp(diag(J^T J), min_lm_diagonal, max_lm_diagonal) / radius
constexpr double kLmDiagMin = 1e-6, kLmDiagMax = 1e32;

// Damped point block H = Hpp + D^2 (Hpp: upper triangle {00,01,02,11,12,22}) -> upper-triangular C = L^-T of its Cholesky
// factor H = L L^T, so that H^-1 = C C^T: out = {c00, c01, c02, c11, c12, c22}.  What k_point_prep used to store per point
// (48 + 48 bytes written, read again by the S assembly and the back-substitution) is ~45 instructions on the 48 bytes of Hpp
// the consumers load instead.  Three reciprocal square roots, no division.
__device__ __forceinline__ void point_factor(const double h[6], double radius, double c[6]) {
    const double ir = 1.0 / radius;
    const double h00 = fma(clampd(h[0], kLmDiagMin, kLmDiagMax), ir, h[0]);
    const double h11 = fma(clampd(h[3], kLmDiagMin, kLmDiagMax), ir, h[3]);
    const double h22 = fma(clampd(h[5], kLmDiagMin, kLmDiagMax), ir, h[5]);
    const double r0 = fast_rsqrt(h00);
    const double l10 = h[1] * r0, l20 = h[2] * r0;
    const double r1 = fast_rsqrt(fmax(fma(-l10, l10, h11), 1e-300));
    const double l21 = fma(-l20, l10, h[4]) * r1;
    const double r2 = fast_rsqrt(fmax(fma(-l21, l21, fma(-l20, l20, h22)), 1e-300));
    const double m10 = -l10 * r0 * r1;
    const double m21 = -l21 * r1 * r2;
    const double m20 = -fma(l20, r0, l21 * m10) * r2;
    c[0] = r0; c[1] = m10; c[2] = m20; c[3] = r1; c[4] = m21; c[5] = r2;
}

}  // namespace xba
