// Scaled pose graph of BASolver::ScalePoseGraphUnorder — HOST code, no GPU: O(frames) unknowns, the
// survey rates a GPU version as not justified.  It exists so that the drop-in BASolver needs no Ceres at all.
//
// Restated from /root/reference/src/optimization (no code shared):
//   residual + Jacobians   PoseGraphCost  cost_factor_ceres.h:117-198   (8 rows: rotation 3, scale ratio, scale prior, position 3)
//                          ScaleCost      cost_factor_ceres.h:200-221
//   logmap                 lie_algebra.h:12-15 (Eigen::AngleAxisd(q): angle * axis)
//   what is variable       ba_solver.cc:233-256: rotations constant (:248-249 "may bug"), positions and per-frame scales free,
//                          lower bound 0.2 on the scales (:245-247, :251-252), gauge = position and scale of the two init frames
//   solver                 ba_solver.cc:258-266: DOGLEG, initial radius 1e16, max 100 iterations, Ceres default tolerances
// Ceres itself is not available: the trust-region logic below follows the published traditional-dogleg strategy
// (Gauss-Newton step with a 1e-8 relative regulariser, Cauchy point, radius update 0.5x / max(r, 3|step|)) with Ceres'
// handling of bounds (projection inside Plus + the projected Armijo search of line_search.h on every step).  PARITY UNPINNED against real Ceres;
// tests compare the minimum with an independent bounded least-squares solver (scipy) on the same residuals.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <limits>
#include <vector>

#include "../../include/xrsfm_ba.h"
#include "line_search.h"

namespace xpg {

inline void quat_mul(const double* a, const double* b, double* o) {   // x,y,z,w
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
inline void quat_conj(const double* a, double* o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
inline void quat_to_rot(const double* q, double* R) {   // row-major, Eigen's toRotationMatrix (no normalisation)
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
inline void quat_log(const double* q, double* w) {      // angle * axis of Eigen::AngleAxisd(q)
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n == 0.0) { w[0] = w[1] = w[2] = 0.0; return; }
    const double angle = 2.0 * std::atan2(n, std::fabs(q[3]));
    if (q[3] < 0) n = -n;
    for (int k = 0; k < 3; ++k) w[k] = angle * q[k] / n;
}

struct Graph {
    const xrsfm_pg_problem& p;
    std::vector<double> R;            // [n_edges][9]  R1^T of the edge's first pose
    std::vector<double> rot_cost;     // sum of squares of the (constant) rotation rows
    std::vector<int> vp, vs;          // variable index of positions (first of 3) / scales, -1 if constant
    int nv = 0;
    double const_cost = 0.0;

    explicit Graph(const xrsfm_pg_problem& pr) : p(pr) {
        vp.assign(p.n_frames, -1); vs.assign(p.n_scales, -1);
        std::vector<char> used_p(p.n_frames, 0), used_s(p.n_scales, 0);
        for (int e = 0; e < p.n_edges; ++e) { used_p[p.edge_a[e]] = used_p[p.edge_b[e]] = 1; used_s[p.edge_sa[e]] = used_s[p.edge_sb[e]] = 1; }
        for (int e = 0; e < p.n_scale_costs; ++e) { used_s[p.sc_a[e]] = used_s[p.sc_b[e]] = 1; }
        // frame-major order (position, then the frame's own scale): neighbours in the graph stay close in the envelope
        for (int i = 0; i < std::max(p.n_frames, p.n_scales); ++i) {
            if (i < p.n_frames && used_p[i] && !(p.pos_const && p.pos_const[i])) { vp[i] = nv; nv += 3; }
            if (i < p.n_scales && used_s[i] && !(p.scale_const && p.scale_const[i])) { vs[i] = nv; nv += 1; }
        }
        R.resize(9 * (size_t)p.n_edges);
        for (int e = 0; e < p.n_edges; ++e) {
            double Rm[9];
            quat_to_rot(p.rot_q + 4 * (size_t)p.edge_a[e], Rm);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R[9 * (size_t)e + 3 * r + c] = Rm[3 * c + r];   // transpose
            // rotation rows: logmap(q_mea * (q1^-1 q2)^-1), constant because the rotations are
            double q1i[4], q12[4], q12i[4], d[4], w[3];
            quat_conj(p.rot_q + 4 * (size_t)p.edge_a[e], q1i);
            const double n1 = q1i[0] * q1i[0] + q1i[1] * q1i[1] + q1i[2] * q1i[2] + q1i[3] * q1i[3];
            for (int k = 0; k < 4; ++k) q1i[k] /= n1;                                    // Eigen inverse = conjugate / squaredNorm
            quat_mul(q1i, p.rot_q + 4 * (size_t)p.edge_b[e], q12);
            quat_conj(q12, q12i);
            const double n2 = q12[0] * q12[0] + q12[1] * q12[1] + q12[2] * q12[2] + q12[3] * q12[3];
            for (int k = 0; k < 4; ++k) q12i[k] /= n2;
            quat_mul(p.edge_q_mea + 4 * (size_t)e, q12i, d);
            quat_log(d, w);
            const_cost += w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
        }
    }

    // residual rows that depend on the variables: per edge 5 (scale ratio, scale prior, position 3), per scale cost 1
    int n_rows() const { return 5 * p.n_edges + p.n_scale_costs; }

    // cost = 1/2 (constant rotation part + sum r^2).  If J != nullptr: row-compressed Jacobian, 8 entries per edge row
    // block / 2 per scale-cost row, laid out by `fill` below.
    double eval(const double* pos, const double* sc, double* r) const {
        double c = const_cost;
        for (int e = 0; e < p.n_edges; ++e) {
            const int a = p.edge_a[e], b = p.edge_b[e];
            const double s1 = sc[p.edge_sa[e]], s2 = sc[p.edge_sb[e]];
            const double* Rt = R.data() + 9 * (size_t)e;
            const double d[3] = {pos[3 * b] - pos[3 * a], pos[3 * b + 1] - pos[3 * a + 1], pos[3 * b + 2] - pos[3 * a + 2]};
            double* re = r + 5 * (size_t)e;
            re[0] = s1 / s2 - 1.0;
            re[1] = s1 < 1 ? p.weight_o * (s1 - 1.0) : p.weight_o * (1.0 / s1 - 1.0);
            for (int k = 0; k < 3; ++k)
                re[2 + k] = Rt[3 * k] * d[0] + Rt[3 * k + 1] * d[1] + Rt[3 * k + 2] * d[2] - s1 * p.edge_p_mea[3 * (size_t)e + k];
            for (int k = 0; k < 5; ++k) c += re[k] * re[k];
        }
        for (int e = 0; e < p.n_scale_costs; ++e) {
            const double s1 = sc[p.sc_a[e]], s2 = sc[p.sc_b[e]];
            const double v = 10.0 * (s1 / (p.sc_s12[e] * s2) - 1.0);
            r[5 * (size_t)p.n_edges + e] = v;
            c += v * v;
        }
        return 0.5 * c;
    }
};

// Symmetric positive definite system in envelope (skyline) storage: row i holds columns first[i]..i.
struct Envelope {
    int n = 0;
    std::vector<int> first;
    std::vector<size_t> off;
    std::vector<double> a;
    void pattern_begin(int n_) { n = n_; first.resize(n); for (int i = 0; i < n; ++i) first[i] = i; }
    void touch(int i, int j) { if (i < j) std::swap(i, j); first[i] = std::min(first[i], j); }
    void pattern_end() {
        off.resize(n + 1); off[0] = 0;
        for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (size_t)(i - first[i] + 1);
        a.assign(off[n], 0.0);
    }
    double& at(int i, int j) { if (i < j) std::swap(i, j); return a[off[i] + (j - first[i])]; }
    void zero() { std::fill(a.begin(), a.end(), 0.0); }
    // in-place L L^T; false if a pivot is not positive
    bool factor() {
        for (int i = 0; i < n; ++i) {
            double* ri = a.data() + off[i];
            const int fi = first[i];
            for (int j = fi; j <= i; ++j) {
                const double* rj = a.data() + off[j];
                const int fj = first[j];
                double s = ri[j - fi];
                for (int k = std::max(fi, fj); k < j; ++k) s -= ri[k - fi] * rj[k - fj];
                if (j < i) ri[j - fi] = s / rj[j - fj];
                else { if (!(s > 0.0) || !std::isfinite(s)) return false; ri[j - fi] = std::sqrt(s); }
            }
        }
        return true;
    }
    void solve(double* x) const {      // x <- (L L^T)^-1 x
        for (int i = 0; i < n; ++i) {
            const double* ri = a.data() + off[i];
            double s = x[i];
            for (int k = first[i]; k < i; ++k) s -= ri[k - first[i]] * x[k];
            x[i] = s / ri[i - first[i]];
        }
        for (int i = n - 1; i >= 0; --i) {
            const double* ri = a.data() + off[i];
            x[i] /= ri[i - first[i]];
            for (int k = first[i]; k < i; ++k) x[k] -= ri[k - first[i]] * x[i];
        }
    }
};

struct Solver {
    const xrsfm_pg_problem& p;
    Graph g;
    Envelope H;
    std::vector<double> grad, r;
    explicit Solver(const xrsfm_pg_problem& pr) : p(pr), g(pr) {
        H.pattern_begin(g.nv);
        auto link = [&](const int* v, const int* len, int m) {
            for (int x = 0; x < m; ++x) for (int y = 0; y < m; ++y)
                if (v[x] >= 0 && v[y] >= 0) for (int i = 0; i < len[x]; ++i) for (int j = 0; j < len[y]; ++j) H.touch(v[x] + i, v[y] + j);
        };
        for (int e = 0; e < p.n_edges; ++e) {
            const int v[4] = {g.vp[p.edge_a[e]], g.vp[p.edge_b[e]], g.vs[p.edge_sa[e]], g.vs[p.edge_sb[e]]};
            const int len[4] = {3, 3, 1, 1};
            link(v, len, 4);
        }
        for (int e = 0; e < p.n_scale_costs; ++e) {
            const int v[2] = {g.vs[p.sc_a[e]], g.vs[p.sc_b[e]]};
            const int len[2] = {1, 1};
            link(v, len, 2);
        }
        H.pattern_end();
        grad.resize(g.nv); r.resize(g.n_rows());
    }

    // per residual row: up to 8 (variable, value) entries
    struct Row { int idx[8]; double val[8]; int n; };
    template <typename F> void for_rows(const double* pos, const double* sc, F&& f) const {
        for (int e = 0; e < p.n_edges; ++e) {
            const int va = g.vp[p.edge_a[e]], vb = g.vp[p.edge_b[e]], v1 = g.vs[p.edge_sa[e]], v2 = g.vs[p.edge_sb[e]];
            const double s1 = sc[p.edge_sa[e]], s2 = sc[p.edge_sb[e]];
            const double* Rt = g.R.data() + 9 * (size_t)e;
            Row row;
            row.n = 0;                                                   // scale ratio s1/s2 - 1
            if (v1 >= 0) { row.idx[row.n] = v1; row.val[row.n++] = 1.0 / s2; }
            if (v2 >= 0) { row.idx[row.n] = v2; row.val[row.n++] = -s1 / (s2 * s2); }
            f(5 * e + 0, row);
            row.n = 0;                                                   // scale prior
            if (v1 >= 0) { row.idx[row.n] = v1; row.val[row.n++] = s1 < 1 ? p.weight_o : -p.weight_o / (s1 * s1); }
            f(5 * e + 1, row);
            for (int k = 0; k < 3; ++k) {                                // position rows: R1^T (p2 - p1) - s1 p_mea
                row.n = 0;
                if (va >= 0) for (int c = 0; c < 3; ++c) { row.idx[row.n] = va + c; row.val[row.n++] = -Rt[3 * k + c]; }
                if (vb >= 0) for (int c = 0; c < 3; ++c) { row.idx[row.n] = vb + c; row.val[row.n++] = Rt[3 * k + c]; }
                if (v1 >= 0) { row.idx[row.n] = v1; row.val[row.n++] = -p.edge_p_mea[3 * (size_t)e + k]; }
                f(5 * e + 2 + k, row);
            }
        }
        for (int e = 0; e < p.n_scale_costs; ++e) {
            const int v1 = g.vs[p.sc_a[e]], v2 = g.vs[p.sc_b[e]];
            const double s1 = sc[p.sc_a[e]], s2 = sc[p.sc_b[e]], s12 = p.sc_s12[e];
            Row row;
            row.n = 0;
            if (v1 >= 0) { row.idx[row.n] = v1; row.val[row.n++] = 10.0 / (s12 * s2); }
            if (v2 >= 0) { row.idx[row.n] = v2; row.val[row.n++] = -10.0 * s1 / (s12 * s2 * s2); }
            f(5 * p.n_edges + e, row);
        }
    }

    void plus(const double* pos, const double* sc, const double* delta, double* pos_o, double* sc_o) const {
        for (int i = 0; i < p.n_frames; ++i)
            for (int k = 0; k < 3; ++k) pos_o[3 * i + k] = pos[3 * i + k] + (g.vp[i] >= 0 ? delta[g.vp[i] + k] : 0.0);
        for (int i = 0; i < p.n_scales; ++i) {
            double v = sc[i] + (g.vs[i] >= 0 ? delta[g.vs[i]] : 0.0);
            if (g.vs[i] >= 0 && p.scale_lower) v = std::max(v, p.scale_lower[i]);      // ParameterBlock::Plus projects onto the box
            sc_o[i] = v;
        }
    }

    int run(const xrsfm_pg_options& o, xrsfm_pg_summary* sum) {
        const int nv = g.nv;
        std::vector<double> pos(p.pos, p.pos + 3 * (size_t)p.n_frames), sc(p.scale, p.scale + p.n_scales);
        std::vector<double> pos_c(pos.size()), sc_c(sc.size()), r_c(r.size());
        std::vector<double> diag(nv), gs(nv), gn(nv), step(nv), dl(nv), Jd(r.size());
        std::vector<char> active(nv, 0);         // variables held on their bound for the current linearisation
        const bool constrained = p.scale_lower != nullptr;
        if (constrained) { std::vector<double> z(nv, 0.0); plus(pos.data(), sc.data(), z.data(), pos.data(), sc.data()); }   // feasible start
        double cost = g.eval(pos.data(), sc.data(), r.data());
        sum->initial_cost = cost; sum->iterations = 0; sum->n_successful = 0; sum->n_unsuccessful = 0;
        double radius = o.initial_radius, mu = 1e-8;
        const double min_mu = 1e-8, max_mu = 1.0, mu_up = 10.0;
        bool reuse = false;
        double alpha = 0.0, gs_norm = 0.0, gn_norm = 0.0;
        auto finish = [&](int term, double c) {
            for (size_t i = 0; i < pos.size(); ++i) p.pos[i] = pos[i];
            for (size_t i = 0; i < sc.size(); ++i) p.scale[i] = sc[i];
            sum->final_cost = c; sum->termination = term;
            return XRSFM_BA_OK;
        };
        if (!std::isfinite(cost)) return finish(6, cost);        // non-finite input: nothing to minimise (Ceres: FAILURE)
        if (nv == 0) return finish(1, cost);
        for (int it = 0;; ++it) {
            if (!reuse) {
                // gradient first: it decides the active set of this linearisation
                std::fill(grad.begin(), grad.end(), 0.0);
                for_rows(pos.data(), sc.data(), [&](int ri, const Row& row) { for (int x = 0; x < row.n; ++x) grad[row.idx[x]] += row.val[x] * r[ri]; });
                // Default (options.bounds_active_set = 0) = what Ceres does with bounds: Plus projects onto the box and every step
                // goes through the projected Armijo search, nothing else; such a loop can stop above the constrained minimum,
                // and so does upstream (ba_solver.cc:245-266 -> ceres::Solve).
                // bounds_active_set = 1 (a deliberate DEVIATION): projected-Newton active set — a scale that sits
                // on its lower bound while the gradient pushes it further down is held for this step, so that the step of the
                // OTHER variables is the one of the problem restricted to the feasible face; reaches the constrained minimum of an
                // independent bounded least-squares solver (tests/test_pose_graph_cpu.py).
                std::fill(active.begin(), active.end(), 0);
                if (constrained && o.bounds_active_set)
                    for (int i = 0; i < p.n_scales; ++i)
                        if (g.vs[i] >= 0 && sc[i] <= p.scale_lower[i] && grad[g.vs[i]] > 0.0) active[g.vs[i]] = 1;
                // normal equations over the free variables (identity rows for the held ones)
                H.zero();
                for_rows(pos.data(), sc.data(), [&](int ri, const Row& row) {
                    for (int x = 0; x < row.n; ++x) {
                        if (active[row.idx[x]]) continue;
                        for (int y = 0; y <= x; ++y) {
                            if (active[row.idx[y]]) continue;
                            H.at(row.idx[x], row.idx[y]) += row.val[x] * row.val[y] * ((row.idx[x] == row.idx[y] && x != y) ? 2.0 : 1.0);
                        }
                    }
                });
                for (int i = 0; i < nv; ++i) if (active[i]) { H.at(i, i) = 1.0; grad[i] = 0.0; }
                // gradient tolerance on the projected gradient |x - P(x - g)|_inf
                double gmax = 0.0;
                {
                    std::vector<double> mg(nv);
                    for (int i = 0; i < nv; ++i) mg[i] = -grad[i];
                    plus(pos.data(), sc.data(), mg.data(), pos_c.data(), sc_c.data());
                    for (size_t i = 0; i < pos.size(); ++i) gmax = std::max(gmax, std::fabs(pos[i] - pos_c[i]));
                    for (size_t i = 0; i < sc.size(); ++i) gmax = std::max(gmax, std::fabs(sc[i] - sc_c[i]));
                }
                if (o.verbose) printf("pose graph iter %3d  cost %.6e  |g| %.3e  radius %.3e\n", it, cost, gmax, radius);
                if (gmax <= o.gradient_tolerance) return finish(1, cost);
                if (it >= o.max_iterations) return finish(5, cost);
                for (int i = 0; i < nv; ++i) diag[i] = std::sqrt(std::min(std::max(H.at(i, i), 1e-6), 1e32));
                for (int i = 0; i < nv; ++i) gs[i] = grad[i] / diag[i];                      // gradient in the scaled space
                gs_norm = 0.0;
                for (int i = 0; i < nv; ++i) gs_norm += gs[i] * gs[i];
                gs_norm = std::sqrt(gs_norm);
                {   // Cauchy step length alpha = |g|^2 / |J D^-1 g|^2
                    std::vector<double> v(nv);
                    for (int i = 0; i < nv; ++i) v[i] = gs[i] / diag[i];
                    std::fill(Jd.begin(), Jd.end(), 0.0);
                    for_rows(pos.data(), sc.data(), [&](int ri, const Row& row) { for (int x = 0; x < row.n; ++x) Jd[ri] += row.val[x] * v[row.idx[x]]; });   // (v is 0 on held variables)
                    double q = 0.0;
                    for (double x : Jd) q += x * x;
                    alpha = q > 0.0 ? gs_norm * gs_norm / q : 0.0;
                }
                // Gauss-Newton step with the relative regulariser mu (raised until the factorisation succeeds)
                std::vector<double> keep = H.a;
                bool ok = false;
                while (mu <= max_mu) {
                    for (int i = 0; i < nv; ++i) H.at(i, i) += mu * diag[i] * diag[i];
                    if (H.factor()) { ok = true; break; }
                    H.a = keep;
                    mu *= mu_up;
                }
                if (!ok) { mu = max_mu; return finish(6, cost); }
                mu = std::max(min_mu, 2.0 * mu / mu_up);          // relaxed after a successful step in Ceres (DoglegStrategy::StepAccepted: mu = max(min_mu, 2 mu / mu_increase_factor));
                                                                  // done here after the factorisation instead — equivalent, because this loop only re-factorises after an accepted step
                for (int i = 0; i < nv; ++i) gn[i] = -grad[i];
                H.solve(gn.data());
                gn_norm = 0.0;
                for (int i = 0; i < nv; ++i) { gn[i] *= diag[i]; gn_norm += gn[i] * gn[i]; }   // scaled space
                gn_norm = std::sqrt(gn_norm);
            } else if (it >= o.max_iterations) {
                return finish(5, cost);
            }
            // traditional dogleg in the scaled space
            if (gn_norm <= radius) dl = gn;
            else if (gs_norm * alpha >= radius) for (int i = 0; i < nv; ++i) dl[i] = -(radius / gs_norm) * gs[i];
            else {
                double b_dot_a = 0.0, a2 = 0.0, b2 = gn_norm * gn_norm;
                for (int i = 0; i < nv; ++i) { b_dot_a += -alpha * gs[i] * gn[i]; a2 += alpha * alpha * gs[i] * gs[i]; }
                const double bma2 = a2 + b2 - 2 * b_dot_a, c2 = b_dot_a - a2;
                const double dd = std::sqrt(c2 * c2 + bma2 * (radius * radius - a2));
                const double beta = (c2 <= 0) ? (dd - c2) / bma2 : (radius * radius - a2) / (dd + c2);
                for (int i = 0; i < nv; ++i) dl[i] = -alpha * (1 - beta) * gs[i] + beta * gn[i];
            }
            double dl_norm = 0.0;
            for (int i = 0; i < nv; ++i) { dl_norm += dl[i] * dl[i]; step[i] = dl[i] / diag[i]; }
            dl_norm = std::sqrt(dl_norm);
            sum->iterations = it + 1;
            // model cost change -(J d)^T (r + J d / 2) of the trust-region step (Ceres keeps it when the line search shortens d)
            std::fill(Jd.begin(), Jd.end(), 0.0);
            for_rows(pos.data(), sc.data(), [&](int ri, const Row& row) { for (int x = 0; x < row.n; ++x) Jd[ri] += row.val[x] * step[row.idx[x]]; });
            double model = 0.0;
            for (size_t i = 0; i < Jd.size(); ++i) model -= Jd[i] * (r[i] + 0.5 * Jd[i]);
            if (!(model > 0.0) || !std::isfinite(model)) {           // invalid step
                sum->n_unsuccessful++;
                radius *= 0.5; reuse = true;
                if (radius < 1e-32) return finish(4, cost);
                continue;
            }
            // bounds: projected Armijo search along the step before it is evaluated (TrustRegionMinimizer::DoLineSearch)
            double cost_c = 0.0;
            if (constrained) {
                double slope0 = 0.0, dmax = 0.0;
                for (int i = 0; i < nv; ++i) { slope0 += grad[i] * step[i]; dmax = std::max(dmax, std::fabs(step[i])); }
                std::vector<double> st(nv), gt(nv);
                auto eval = [&](double a, xls::Sample& sm) {
                    for (int i = 0; i < nv; ++i) st[i] = a * step[i];
                    plus(pos.data(), sc.data(), st.data(), pos_c.data(), sc_c.data());
                    sm.x = a; sm.f = g.eval(pos_c.data(), sc_c.data(), r_c.data()); sm.has_g = true;
                    std::fill(gt.begin(), gt.end(), 0.0);
                    for_rows(pos_c.data(), sc_c.data(), [&](int ri, const Row& row) { for (int x = 0; x < row.n; ++x) gt[row.idx[x]] += row.val[x] * r_c[ri]; });
                    sm.g = 0.0;
                    for (int i = 0; i < nv; ++i) sm.g += gt[i] * step[i];
                    return std::isfinite(sm.f);
                };
                const double t = xls::armijo_search(eval, cost, slope0, dmax);
                if (t != 1.0) for (int i = 0; i < nv; ++i) step[i] *= t;
            }
            plus(pos.data(), sc.data(), step.data(), pos_c.data(), sc_c.data());
            cost_c = g.eval(pos_c.data(), sc_c.data(), r_c.data());
            double xnorm = 0.0, snorm = 0.0;
            for (int i = 0; i < p.n_frames; ++i) if (g.vp[i] >= 0) for (int k = 0; k < 3; ++k) { xnorm += pos[3 * i + k] * pos[3 * i + k]; const double d = pos_c[3 * i + k] - pos[3 * i + k]; snorm += d * d; }
            for (int i = 0; i < p.n_scales; ++i) if (g.vs[i] >= 0) { xnorm += sc[i] * sc[i]; const double d = sc_c[i] - sc[i]; snorm += d * d; }
            xnorm = std::sqrt(xnorm); snorm = std::sqrt(snorm);
            if (snorm <= o.parameter_tolerance * (xnorm + o.parameter_tolerance)) return finish(2, cost);
            const double change = cost - cost_c;
            if (std::fabs(change) <= o.function_tolerance * cost) return finish(3, cost);
            const double rho = change / model;
            if (rho > 1e-3) {
                pos.swap(pos_c); sc.swap(sc_c); r.swap(r_c); cost = cost_c;
                sum->n_successful++;
                if (rho < 0.25) radius *= 0.5;
                if (rho > 0.75) radius = std::max(radius, 3.0 * dl_norm);
                reuse = false;
            } else {
                sum->n_unsuccessful++;
                radius *= 0.5; reuse = true;
                if (radius < 1e-32) return finish(4, cost);
            }
        }
    }
};

}  // namespace xpg
