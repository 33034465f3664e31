// Projected Armijo line search of Ceres' bounds-constrained trust-region path (TrustRegionMinimizer::DoLineSearch ->
// ArmijoLineSearch with the solver defaults: sufficient decrease 1e-4, CUBIC interpolation through values and slopes,
// contraction of the trial step within [1e-3, 0.6], at most 20 trials, minimum step 1e-9 in the max-norm), restated for the
// two host solvers that carry the reference's lower bound on a scale (pose_graph.h, tag_refine.h).  Ceres is not in
// /root/reference; this follows its published behaviour.
#ifndef XRSFM_AMD_LINE_SEARCH_H
#define XRSFM_AMD_LINE_SEARCH_H

#include <algorithm>
#include <cmath>

namespace xls {

// Minimum over [lo, hi] of the polynomial that interpolates the given values / slopes (Ceres: MinimizeInterpolatingPolynomial).
struct Sample { double x, f, g; bool has_g; };
inline double minimize_interpolant(const Sample* s, int n, double lo, double hi) {
    int m = 0;
    for (int i = 0; i < n; ++i) m += s[i].has_g ? 2 : 1;
    double A[6][7];
    int row = 0;
    for (int i = 0; i < n; ++i) {          // coefficient k multiplies x^k
        double pw = 1.0;
        for (int k = 0; k < m; ++k) { A[row][k] = pw; pw *= s[i].x; }
        A[row][m] = s[i].f; ++row;
        if (s[i].has_g) {
            pw = 1.0;
            A[row][0] = 0.0;
            for (int k = 1; k < m; ++k) { A[row][k] = k * pw; pw *= s[i].x; }
            A[row][m] = s[i].g; ++row;
        }
    }
    for (int c = 0; c < m; ++c) {          // Gauss-Jordan with partial pivoting
        int piv = c;
        for (int r = c + 1; r < m; ++r) if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0) return 0.5 * (lo + hi);
        if (piv != c) for (int k = 0; k <= m; ++k) std::swap(A[piv][k], A[c][k]);
        for (int r = 0; r < m; ++r) if (r != c) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k <= m; ++k) A[r][k] -= f * A[c][k];
        }
    }
    double coef[6];
    for (int k = 0; k < m; ++k) coef[k] = A[k][m] / A[k][k];
    auto val = [&](double x) { double v = 0.0; for (int k = m - 1; k >= 0; --k) v = v * x + coef[k]; return v; };
    auto der = [&](double x) { double v = 0.0; for (int k = m - 1; k >= 1; --k) v = v * x + k * coef[k]; return v; };
    double best_x = lo, best = val(lo);
    if (val(hi) < best) { best = val(hi); best_x = hi; }
    const int kCells = 256;                // stationary points inside the bracket: sign changes of the derivative, bisected
    double xa = lo, da = der(lo);
    for (int i = 1; i <= kCells; ++i) {
        const double xb = lo + (hi - lo) * i / kCells, db = der(xb);
        if ((da < 0.0 && db >= 0.0) || (da > 0.0 && db <= 0.0)) {
            double l = xa, r = xb, dl = da;
            for (int it = 0; it < 80; ++it) {
                const double mid = 0.5 * (l + r), dm = der(mid);
                if ((dl < 0.0) == (dm < 0.0)) { l = mid; dl = dm; } else r = mid;
            }
            const double x = 0.5 * (l + r), v = val(x);
            if (v < best) { best = v; best_x = x; }
        }
        xa = xb; da = db;
    }
    return best_x;
}

// eval(a, sample): cost and directional derivative at x (+) a * delta (projected); returns false when the cost is not finite.
// Returns the factor to apply to delta; 1.0 when the search gives up (Ceres then keeps the step as it is).
template <typename Eval>
inline double armijo_search(Eval&& eval, double cost, double slope0, double delta_max_norm) {
    const Sample start = {0.0, cost, slope0, true};
    Sample prev = {0, 0, 0, false}, cur;
    bool prev_valid = false;
    bool cur_valid = eval(1.0, cur);
    int iters = 0;
    while (!cur_valid || cur.f > cost + 1e-4 * slope0 * cur.x) {
        if (++iters >= 20) return 1.0;
        double a;
        const double lo = 1e-3 * cur.x, hi = 0.6 * cur.x;
        if (!cur_valid) a = std::min(std::max(0.5 * cur.x, lo), hi);
        else {
            const Sample ss[3] = {start, cur, prev};
            a = minimize_interpolant(ss, prev_valid ? 3 : 2, lo, hi);
        }
        if (a * delta_max_norm < 1e-9) return 1.0;
        prev = cur; prev_valid = cur_valid;
        cur_valid = eval(a, cur);
    }
    return cur.x;
}

}  // namespace xls
#endif
