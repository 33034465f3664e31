// Metric-scale refinement against AprilTag corners: the two ceres::Solve calls of tag_refine
// (/root/reference/src/tag/tag_extract.hpp:193-265).  HOST code, no GPU: the unknowns are one scale, one
// 6-DoF pose per tag and -- in the second solve -- 3-D points whose cameras are all constant, so the normal matrix is
// block diagonal (3x3 per track point, one (6|18)x(6|18) block per tag) bordered by the single scale column.  What is
// restated here in our own words:
//   * TagCost        (cost_factor_ceres.h:223-260)  r = w (p - (R (s p_tag) + t)), analytic Jacobians as written there
//   * ProjectionCost (cost_factor_ceres.h:66-112)   normalised-plane residual with the functor's own sqrt(sigma/|r|) factor
//                                                   (applied to residual and Jacobian alike, its derivative ignored)
//   * QuatParam      (cost_factor_ceres.h:262-282)  q <- (q * exp(theta)).normalized(), local Jacobian [I;0]
//   * ceres::Solver::Options defaults with max_num_iterations = 500 (tag_extract.hpp:229-231): Levenberg-Marquardt, exact
//     linear solve, Jacobi scaling fixed at the first iterate, radius 1e4, rho > 1e-3, tolerances 1e-6 / 1e-8 / 1e-10
//   * the lower bound 0.2 on the scale (tag_extract.hpp:227) makes the problem "constrained": Plus() projects onto the box,
//     the gradient test uses |x - P(x - g)|, and every trust-region step goes through the projected Armijo search
//     (sufficient decrease 1e-4, cubic interpolation, contraction in [1e-3, 0.6], 20 trials) before it is evaluated; the
//     model decrease used for rho stays the one of the unshortened step, like in Ceres.
// Ceres itself is not in /root/reference: parity unpinned (the tests check against an independent scipy solve).
#ifndef XRSFM_AMD_TAG_REFINE_H
#define XRSFM_AMD_TAG_REFINE_H

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <vector>

#include <cstdint>

#include "../../include/xrsfm_ba.h"
#include "line_search.h"

namespace xtag {

inline void quat_to_rot(const double* q, double* R) {   // x,y,z,w -> row-major, Eigen's toRotationMatrix (no normalisation)
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// q <- (q * exp(theta)).normalized()
inline void quat_plus(const double* q, const double* th, double* o) {
    const double n = std::sqrt(th[0] * th[0] + th[1] * th[1] + th[2] * th[2]);
    double d[4] = {0, 0, 0, 1};
    if (n > 0.0) {
        const double s = std::sin(0.5 * n) / n;
        d[0] = s * th[0]; d[1] = s * th[1]; d[2] = s * th[2]; d[3] = std::cos(0.5 * n);
    }
    const double r[4] = {q[3] * d[0] + q[0] * d[3] + q[1] * d[2] - q[2] * d[1],
                         q[3] * d[1] - q[0] * d[2] + q[1] * d[3] + q[2] * d[0],
                         q[3] * d[2] + q[0] * d[1] - q[1] * d[0] + q[2] * d[3],
                         q[3] * d[3] - q[0] * d[0] - q[1] * d[1] - q[2] * d[2]};
    const double m = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int k = 0; k < 4; ++k) o[k] = r[k] / m;
}

// 3x3 symmetric positive definite solve (lower: a00 a10 a11 a20 a21 a22); false when not positive definite
inline bool solve3(const double* a, const double* b, double* x) {
    const double l00 = a[0] > 0 ? std::sqrt(a[0]) : 0.0;
    if (!(l00 > 0)) return false;
    const double l10 = a[1] / l00, l20 = a[3] / l00;
    const double d1 = a[2] - l10 * l10;
    if (!(d1 > 0)) return false;
    const double l11 = std::sqrt(d1), l21 = (a[4] - l20 * l10) / l11;
    const double d2 = a[5] - l20 * l20 - l21 * l21;
    if (!(d2 > 0)) return false;
    const double l22 = std::sqrt(d2);
    const double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11, y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    x[2] = y2 / l22; x[1] = (y1 - l21 * x[2]) / l11; x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
    return true;
}

// dense Cholesky of an m x m row-major matrix in place (lower), m <= 18
inline bool chol_small(double* A, int m) {
    for (int j = 0; j < m; ++j) {
        double d = A[j * m + j];
        for (int k = 0; k < j; ++k) d -= A[j * m + k] * A[j * m + k];
        if (!(d > 0)) return false;
        d = std::sqrt(d);
        A[j * m + j] = d;
        for (int i = j + 1; i < m; ++i) {
            double v = A[i * m + j];
            for (int k = 0; k < j; ++k) v -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = v / d;
        }
    }
    return true;
}
inline void chol_small_solve(const double* L, int m, double* b) {
    for (int i = 0; i < m; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= L[i * m + k] * b[k]; b[i] = v / L[i * m + i]; }
    for (int i = m - 1; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < m; ++k) v -= L[k * m + i] * b[k]; b[i] = v / L[i * m + i]; }
}

struct State {
    double scale;
    std::vector<double> tag_q, tag_t, corners, points;
};

// Tangent-space vector: [scale | per tag: theta(3) t(3) (corners 4x3) | per track point 3]
struct Solver {
    const xrsfm_tag_problem& p;
    const int stage;                 // 1: tag poses + scale; 2: also tag corners and track points
    const int T, P, m;               // tags, track points, unknowns per tag block
    const double sigma = 5.99 / 700.0;        // ProjectionCost default (cost_factor_ceres.h:68)
    std::vector<double> frame_R;     // [n_frames][9]
    std::vector<uint8_t> point_used; // track points with at least one observation are parameter blocks of the problem
    std::vector<double> tag_pt;      // get_tag(tag_length): 4 corners in the tag frame (tag_extract.hpp:123-131)
    int nt() const { return 1 + T * m + 3 * P; }

    // normal equations at the current linearisation point
    double Hs;                        // scale-scale
    std::vector<double> Hb, Hbs;      // per tag: m x m block, m border (coupling with the scale)
    std::vector<double> Hp;           // per point: 6 (lower 3x3)
    bool scale_held = false;          // the scale is on its lower bound and held for the current step (active set)
    std::vector<double> grad;         // tangent-space gradient J^T r

    Solver(const xrsfm_tag_problem& pr, int stage_)
        : p(pr), stage(stage_), T(pr.n_tags), P(stage_ == 2 ? pr.n_points : 0), m(stage_ == 2 ? 18 : 6) {
        frame_R.resize(9 * (size_t)p.n_frames);
        for (int i = 0; i < p.n_frames; ++i) quat_to_rot(p.frame_q + 4 * (size_t)i, frame_R.data() + 9 * (size_t)i);
        point_used.assign(P, 0);
        if (stage == 2) for (int i = 0; i < p.n_obs; ++i) point_used[p.obs_pt[i]] = 1;
        const double L = p.tag_length;
        tag_pt = {0, 0, 0, L, 0, 0, L, 0, L, 0, 0, L};
        Hb.resize((size_t)T * m * m); Hbs.resize((size_t)T * m); Hp.resize(6 * (size_t)P); grad.resize(nt());
    }

    // ProjectionCost: residual (2) and d r / d P_w (2x3) for frame f, normalised observation xy
    inline void projection(int f, const double* xy, const double* Pw, double* r, double* J) const {
        const double* R = frame_R.data() + 9 * (size_t)f;
        const double* t = p.frame_t + 3 * (size_t)f;
        const double x = R[0] * Pw[0] + R[1] * Pw[1] + R[2] * Pw[2] + t[0];
        const double y = R[3] * Pw[0] + R[4] * Pw[1] + R[5] * Pw[2] + t[1];
        const double z = R[6] * Pw[0] + R[7] * Pw[1] + R[8] * Pw[2] + t[2];
        r[0] = x / z - xy[0]; r[1] = y / z - xy[1];
        const double r2 = r[0] * r[0] + r[1] * r[1];
        const double hf = r2 > sigma * sigma ? std::sqrt(sigma / std::sqrt(r2)) : 1.0;
        r[0] *= hf; r[1] *= hf;
        if (J) {
            const double a = hf / z, bx = -hf * x / (z * z), by = -hf * y / (z * z);
            for (int c = 0; c < 3; ++c) { J[c] = a * R[c] + bx * R[6 + c]; J[3 + c] = a * R[3 + c] + by * R[6 + c]; }
        }
    }

    // cost = 1/2 sum r^2 of the residual blocks that have a variable parameter; with_normal also fills H*, grad
    double evaluate(const State& s, bool with_normal) {
        double cost2 = 0.0;
        if (with_normal) {
            Hs = 0.0;
            std::fill(Hb.begin(), Hb.end(), 0.0); std::fill(Hbs.begin(), Hbs.end(), 0.0);
            std::fill(Hp.begin(), Hp.end(), 0.0); std::fill(grad.begin(), grad.end(), 0.0);
        }
        for (int k = 0; k < T; ++k) {
            double R[9];
            quat_to_rot(s.tag_q.data() + 4 * (size_t)k, R);
            double* B = Hb.data() + (size_t)k * m * m;
            double* bs = Hbs.data() + (size_t)k * m;
            double* g = grad.data() + 1 + (size_t)k * m;
            for (int c = 0; c < 4; ++c) {
                const double* p0 = tag_pt.data() + 3 * c;
                const double* pw = s.corners.data() + 12 * (size_t)k + 3 * c;
                double Rp[3];
                for (int i = 0; i < 3; ++i) Rp[i] = R[3 * i] * p0[0] + R[3 * i + 1] * p0[1] + R[3 * i + 2] * p0[2];
                double r[3];
                for (int i = 0; i < 3; ++i) r[i] = pw[i] - (s.scale * Rp[i] + s.tag_t[3 * (size_t)k + i]);
                cost2 += r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                if (!with_normal) continue;
                // rows i = 0..2; columns: theta(3) = R [s p0]x, t(3) = -I, scale = -R p0, corner(3) = +I
                const double v[3] = {s.scale * p0[0], s.scale * p0[1], s.scale * p0[2]};
                const double K[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
                double J[3][10];
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) J[i][j] = R[3 * i] * K[j] + R[3 * i + 1] * K[3 + j] + R[3 * i + 2] * K[6 + j];
                    for (int j = 0; j < 3; ++j) J[i][3 + j] = (i == j) ? -1.0 : 0.0;
                    J[i][6] = -Rp[i];
                    for (int j = 0; j < 3; ++j) J[i][7 + j] = (i == j) ? 1.0 : 0.0;
                }
                int col[10];
                for (int j = 0; j < 6; ++j) col[j] = j;
                col[6] = -1;                                     // scale
                for (int j = 0; j < 3; ++j) col[7 + j] = stage == 2 ? 6 + 3 * c + j : -2;
                for (int a = 0; a < 10; ++a) {
                    if (col[a] == -2) continue;
                    double ga = 0.0;
                    for (int i = 0; i < 3; ++i) ga += J[i][a] * r[i];
                    if (col[a] == -1) grad[0] += ga; else g[col[a]] += ga;
                    for (int b = 0; b < 10; ++b) {
                        if (col[b] == -2) continue;
                        double h = 0.0;
                        for (int i = 0; i < 3; ++i) h += J[i][a] * J[i][b];
                        if (col[a] == -1 && col[b] == -1) Hs += h;
                        else if (col[b] == -1) bs[col[a]] += h;
                        else if (col[a] >= 0) B[col[a] * m + col[b]] += h;
                    }
                }
            }
        }
        if (stage == 2) {
            for (int o = 0; o < p.n_tag_obs; ++o) {          // ProjectionCost on the four corners of an observed tag
                const int k = p.tag_obs_tag[o], f = p.tag_obs_frame[o];
                for (int c = 0; c < 4; ++c) {
                    double r[2], J[6];
                    projection(f, p.tag_obs_xy + 8 * (size_t)o + 2 * c, s.corners.data() + 12 * (size_t)k + 3 * c, r, with_normal ? J : nullptr);
                    cost2 += r[0] * r[0] + r[1] * r[1];
                    if (!with_normal) continue;
                    double* B = Hb.data() + (size_t)k * m * m;
                    double* g = grad.data() + 1 + (size_t)k * m + 6 + 3 * c;
                    for (int a = 0; a < 3; ++a) {
                        g[a] += J[a] * r[0] + J[3 + a] * r[1];
                        for (int b = 0; b < 3; ++b) B[(6 + 3 * c + a) * m + 6 + 3 * c + b] += J[a] * J[b] + J[3 + a] * J[3 + b];
                    }
                }
            }
            for (int o = 0; o < p.n_obs; ++o) {              // ProjectionCost on the map's tracks, every camera constant
                const int j = p.obs_pt[o];
                double r[2], J[6];
                projection(p.obs_frame[o], p.obs_xy + 2 * (size_t)o, s.points.data() + 3 * (size_t)j, r, with_normal ? J : nullptr);
                cost2 += r[0] * r[0] + r[1] * r[1];
                if (!with_normal) continue;
                double* h = Hp.data() + 6 * (size_t)j;
                double* g = grad.data() + 1 + (size_t)T * m + 3 * (size_t)j;
                for (int a = 0; a < 3; ++a) g[a] += J[a] * r[0] + J[3 + a] * r[1];
                h[0] += J[0] * J[0] + J[3] * J[3];
                h[1] += J[1] * J[0] + J[4] * J[3]; h[2] += J[1] * J[1] + J[4] * J[4];
                h[3] += J[2] * J[0] + J[5] * J[3]; h[4] += J[2] * J[1] + J[5] * J[4]; h[5] += J[2] * J[2] + J[5] * J[5];
            }
        }
        return 0.5 * cost2;
    }

    double diag(int i) const {          // diagonal of J^T J in tangent coordinates
        if (i == 0) return Hs;
        i -= 1;
        if (i < T * m) { const int k = i / m, a = i % m; return Hb[(size_t)k * m * m + a * m + a]; }
        i -= T * m;
        static const int dd[3] = {0, 2, 5};
        return Hp[6 * (size_t)(i / 3) + dd[i % 3]];
    }

    // y = H x (H = J^T J, unscaled)
    void multiply(const double* x, double* y) const {
        double ys = Hs * x[0];
        for (int k = 0; k < T; ++k) {
            const double* B = Hb.data() + (size_t)k * m * m;
            const double* bs = Hbs.data() + (size_t)k * m;
            const double* xk = x + 1 + (size_t)k * m;
            double* yk = y + 1 + (size_t)k * m;
            for (int a = 0; a < m; ++a) {
                double v = bs[a] * x[0];
                for (int b = 0; b < m; ++b) v += B[a * m + b] * xk[b];
                yk[a] = v;
                ys += bs[a] * xk[a];
            }
        }
        y[0] = ys;
        for (int j = 0; j < P; ++j) {
            const double* h = Hp.data() + 6 * (size_t)j;
            const double* xj = x + 1 + (size_t)T * m + 3 * (size_t)j;
            double* yj = y + 1 + (size_t)T * m + 3 * (size_t)j;
            yj[0] = h[0] * xj[0] + h[1] * xj[1] + h[3] * xj[2];
            yj[1] = h[1] * xj[0] + h[2] * xj[1] + h[4] * xj[2];
            yj[2] = h[3] * xj[0] + h[4] * xj[1] + h[5] * xj[2];
        }
    }

    // (S H S + D^2) ds = -S g with D^2 = d2 (already divided by the radius); returns the step S ds in tangent coordinates
    bool lm_step(const std::vector<double>& S, const std::vector<double>& d2, std::vector<double>& step) const {
        const int n = nt();
        step.assign(n, 0.0);
        // points: independent 3x3 systems
        for (int j = 0; j < P; ++j) {
            if (!point_used[j]) continue;
            const int o = 1 + T * m + 3 * j;
            const double* h = Hp.data() + 6 * (size_t)j;
            const double a[6] = {h[0] * S[o] * S[o] + d2[o], h[1] * S[o + 1] * S[o], h[2] * S[o + 1] * S[o + 1] + d2[o + 1],
                                 h[3] * S[o + 2] * S[o], h[4] * S[o + 2] * S[o + 1], h[5] * S[o + 2] * S[o + 2] + d2[o + 2]};
            const double b[3] = {-S[o] * grad[o], -S[o + 1] * grad[o + 1], -S[o + 2] * grad[o + 2]};
            double x[3];
            if (!solve3(a, b, x)) return false;
            for (int c = 0; c < 3; ++c) step[o + c] = S[o + c] * x[c];
        }
        // tags: bordered block diagonal, eliminated onto the scale
        double ss = Hs * S[0] * S[0] + d2[0], rs = -S[0] * grad[0];
        std::vector<double> L((size_t)T * m * m), w((size_t)T * m), u((size_t)T * m);
        for (int k = 0; k < T; ++k) {
            const int o = 1 + k * m;
            double* Lk = L.data() + (size_t)k * m * m;
            const double* B = Hb.data() + (size_t)k * m * m;
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b) Lk[a * m + b] = B[a * m + b] * S[o + a] * S[o + b] + (a == b ? d2[o + a] : 0.0);
            if (!chol_small(Lk, m)) return false;
            double* wk = w.data() + (size_t)k * m;      // B^-1 border
            double* uk = u.data() + (size_t)k * m;      // B^-1 rhs
            for (int a = 0; a < m; ++a) { wk[a] = Hbs[(size_t)k * m + a] * S[o + a] * S[0]; uk[a] = -S[o + a] * grad[o + a]; }
            std::vector<double> border(wk, wk + m);
            chol_small_solve(Lk, m, wk); chol_small_solve(Lk, m, uk);
            for (int a = 0; a < m; ++a) { ss -= border[a] * wk[a]; rs -= border[a] * uk[a]; }
        }
        if (!scale_held && !(ss > 0)) return false;
        const double xs = scale_held ? 0.0 : rs / ss;          // (active set, see run())
        step[0] = S[0] * xs;
        for (int k = 0; k < T; ++k)
            for (int a = 0; a < m; ++a) step[1 + k * m + a] = S[1 + k * m + a] * (u[(size_t)k * m + a] - w[(size_t)k * m + a] * xs);
        for (double v : step) if (!std::isfinite(v)) return false;
        return true;
    }

    // x (+) delta, projected onto scale >= lower bound
    void plus(const State& x, const double* d, State& o) const {
        o = x;
        o.scale = std::max(x.scale + d[0], p.scale_lower);
        for (int k = 0; k < T; ++k) {
            const double* dk = d + 1 + (size_t)k * m;
            quat_plus(x.tag_q.data() + 4 * (size_t)k, dk, o.tag_q.data() + 4 * (size_t)k);
            for (int c = 0; c < 3; ++c) o.tag_t[3 * (size_t)k + c] = x.tag_t[3 * (size_t)k + c] + dk[3 + c];
            if (stage == 2) for (int c = 0; c < 12; ++c) o.corners[12 * (size_t)k + c] = x.corners[12 * (size_t)k + c] + dk[6 + c];
        }
        for (int j = 0; j < P; ++j) {
            if (!point_used[j]) continue;
            const double* dj = d + 1 + (size_t)T * m + 3 * (size_t)j;
            for (int c = 0; c < 3; ++c) o.points[3 * (size_t)j + c] = x.points[3 * (size_t)j + c] + dj[c];
        }
    }

    // ambient-space norms over the variable parameter blocks
    double diff_norm2(const State& a, const State* b, bool inf_norm) const {
        double acc = 0.0;
        auto add = [&](double va, double vb) { const double d = va - vb; if (inf_norm) acc = std::max(acc, std::fabs(d)); else acc += d * d; };
        add(a.scale, b ? b->scale : 0.0);
        for (size_t i = 0; i < a.tag_q.size(); ++i) add(a.tag_q[i], b ? b->tag_q[i] : 0.0);
        for (size_t i = 0; i < a.tag_t.size(); ++i) add(a.tag_t[i], b ? b->tag_t[i] : 0.0);
        if (stage == 2) {
            for (size_t i = 0; i < a.corners.size(); ++i) add(a.corners[i], b ? b->corners[i] : 0.0);
            for (int j = 0; j < P; ++j) if (point_used[j]) for (int c = 0; c < 3; ++c) add(a.points[3 * (size_t)j + c], b ? b->points[3 * (size_t)j + c] : 0.0);
        }
        return acc;
    }

    double projected_gradient_max(const State& x, State& tmp) const {
        std::vector<double> ng(grad.size());
        for (size_t i = 0; i < grad.size(); ++i) ng[i] = -grad[i];
        plus(x, ng.data(), tmp);
        return diff_norm2(x, &tmp, true);
    }

    // Projected Armijo search along delta (TrustRegionMinimizer::DoLineSearch): returns the factor to apply to delta.
    // Leaves grad/H* overwritten by trial evaluations; the caller restores them (keep/restore below).
    double line_search(const State& x, double cost, const std::vector<double>& g0, const std::vector<double>& delta, State& tmp) {
        double slope0 = 0.0, dmax = 0.0;
        for (size_t i = 0; i < delta.size(); ++i) { slope0 += g0[i] * delta[i]; dmax = std::max(dmax, std::fabs(delta[i])); }
        std::vector<double> d(delta.size());
        auto eval = [&](double a, xls::Sample& s) {
            for (size_t i = 0; i < d.size(); ++i) d[i] = a * delta[i];
            plus(x, d.data(), tmp);
            s.x = a; s.f = evaluate(tmp, true); s.has_g = true;
            s.g = 0.0;
            for (size_t i = 0; i < d.size(); ++i) s.g += grad[i] * delta[i];
            return std::isfinite(s.f);
        };
        return xls::armijo_search(eval, cost, slope0, dmax);
    }

    int run(const xrsfm_pg_options& o, State& x, xrsfm_pg_summary* sum) {
        const int n = nt();
        State cand = x, tmp = x;
        x.scale = std::max(x.scale, p.scale_lower);
        double cost = evaluate(x, true);
        sum->initial_cost = cost; sum->iterations = 0; sum->n_successful = 0; sum->n_unsuccessful = 0;
        auto finish = [&](int term, double c) { sum->final_cost = c; sum->termination = term; return XRSFM_BA_OK; };
        if (!std::isfinite(cost)) return finish(6, cost);        // non-finite input: nothing to minimise (Ceres: FAILURE)
        std::vector<double> S(n);
        for (int i = 0; i < n; ++i) S[i] = 1.0 / (1.0 + std::sqrt(diag(i)));
        double gmax = projected_gradient_max(x, tmp);
        double xnorm = std::sqrt(diff_norm2(x, nullptr, false));
        double radius = o.initial_radius, decrease = 2.0;
        if (o.verbose) printf("tag refine stage %d iter %3d  cost %.6e  |g| %.3e  radius %.3e\n", stage, 0, cost, gmax, radius);
        if (gmax <= o.gradient_tolerance) return finish(1, cost);
        std::vector<double> d2(n), step, Hstep(n), keep_grad;
        int it = 0, invalid = 0;
        bool reuse_diagonal = false;
        std::vector<double> dg(n);
        while (true) {
            if (it >= o.max_iterations) return finish(5, cost);
            ++it;
            sum->iterations = it;
            // Default (options.bounds_active_set = 0): the bound scale >= scale_lower (tag_extract.hpp:227) is handled as Ceres
            // handles it — projected Plus + projected Armijo search only; with the bound active its model keeps promising the
            // infeasible decrease, rho stays small, the radius collapses and the loop stops above the constrained minimum, as
            // upstream does.  bounds_active_set = 1 (deliberate DEVIATION): while the scale sits on its bound and
            // the gradient pushes it further down it is held, so that the tags take the step of the problem restricted to
            // scale = bound (tests compare that mode with scipy's bounded least squares).
            scale_held = o.bounds_active_set && x.scale <= p.scale_lower && grad[0] > 0.0;
            if (!reuse_diagonal) for (int i = 0; i < n; ++i) dg[i] = std::min(std::max(diag(i) * S[i] * S[i], 1e-6), 1e32);
            for (int i = 0; i < n; ++i) d2[i] = dg[i] / radius;
            double model = -1.0;
            if (lm_step(S, d2, step)) {
                multiply(step.data(), Hstep.data());
                double gd = 0.0, dHd = 0.0;
                for (int i = 0; i < n; ++i) { gd += grad[i] * step[i]; dHd += step[i] * Hstep[i]; }
                model = -(gd + 0.5 * dHd);
            }
            if (!(model > 0.0) || !std::isfinite(model)) {
                sum->n_unsuccessful++;
                if (++invalid >= 5) return finish(6, cost);
                radius /= decrease; decrease *= 2.0; reuse_diagonal = true;
                continue;
            }
            invalid = 0;
            {   // the problem is bounds-constrained: projected line search on every step
                keep_grad = grad;
                const double Hs0 = Hs;
                std::vector<double> kb = Hb, kbs = Hbs, kp = Hp;
                const double a = line_search(x, cost, keep_grad, step, tmp);
                grad.swap(keep_grad); Hs = Hs0; Hb.swap(kb); Hbs.swap(kbs); Hp.swap(kp);
                if (a != 1.0) for (double& v : step) v *= a;
            }
            plus(x, step.data(), cand);
            double cost_c = evaluate(cand, false);
            if (!std::isfinite(cost_c)) cost_c = DBL_MAX;
            const double step_norm = std::sqrt(diff_norm2(x, &cand, false));
            if (step_norm <= o.parameter_tolerance * (xnorm + o.parameter_tolerance)) return finish(2, cost);
            const double change = cost - cost_c;
            if (std::fabs(change) <= o.function_tolerance * cost) return finish(3, cost);
            const double rho = change / model;
            if (rho > 1e-3) {
                std::swap(x, cand);
                cost = evaluate(x, true);
                xnorm = std::sqrt(diff_norm2(x, nullptr, false));
                gmax = projected_gradient_max(x, tmp);
                radius = std::fmin(1e16, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3)));
                decrease = 2.0; reuse_diagonal = false;
                sum->n_successful++;
                if (o.verbose) printf("tag refine stage %d iter %3d  cost %.6e  |g| %.3e  radius %.3e  rho %.3e\n", stage, it, cost, gmax, radius, rho);
                if (gmax <= o.gradient_tolerance) return finish(1, cost);
            } else {
                radius /= decrease; decrease *= 2.0; reuse_diagonal = true;
                sum->n_unsuccessful++;
                if (o.verbose) printf("tag refine stage %d iter %3d  rejected (rho %.3e)  radius %.3e\n", stage, it, rho, radius);
                if (radius < 1e-32) return finish(4, cost);
            }
        }
    }
};

inline int refine(const xrsfm_pg_options& o, xrsfm_tag_problem& p, int stages, xrsfm_pg_summary* sums) {
    State x;
    x.scale = p.scale;
    x.tag_q.assign(p.tag_q, p.tag_q + 4 * (size_t)p.n_tags);
    x.tag_t.assign(p.tag_t, p.tag_t + 3 * (size_t)p.n_tags);
    x.corners.assign(p.tag_corners, p.tag_corners + 12 * (size_t)p.n_tags);
    if (p.n_points > 0) x.points.assign(p.points, p.points + 3 * (size_t)p.n_points);
    for (int st = 1; st <= stages; ++st) {
        Solver s(p, st);
        const int e = s.run(o, x, &sums[st - 1]);
        if (e != XRSFM_BA_OK) return e;
    }
    p.scale = x.scale;
    std::copy(x.tag_q.begin(), x.tag_q.end(), p.tag_q);
    std::copy(x.tag_t.begin(), x.tag_t.end(), p.tag_t);
    if (stages >= 2) {
        std::copy(x.corners.begin(), x.corners.end(), p.tag_corners);
        if (p.n_points > 0) std::copy(x.points.begin(), x.points.end(), p.points);
    }
    return XRSFM_BA_OK;
}

}  // namespace xtag
#endif
