// Pose-only refinement of one frame against fixed 3-D points (RegisterImage, /root/reference/src/geometry/pnp.cc:38-71) as ONE
// persistent workgroup per frame: the whole Levenberg-Marquardt loop runs on the device.  A one-camera problem through the
// general engine is ~10 launches and a host hand-off per iteration for 6 unknowns; here an iteration is one pass of the
// workgroup over the correspondences (residual, Huber, 2x6 block, normal equations by wave shuffles + one LDS exchange), the
// 6x6 damped solve and the trust-region bookkeeping on one lane, and the candidate is linearised in the same pass that
// yields its cost.  Same arithmetic as the engine (ba_math.h: project, huber, quat_plus; the camera block of linearize_item)
// and the same restated Ceres loop as xrsfm_ba_run (Jacobi scaling from the first linearisation, D^2 = clamp(diag)/radius,
// rho > 1e-3, radius update, tolerance exits that keep the current point).  grid.x = frames (batched candidates).
#pragma once

#include "ba_kernels.h"

namespace xba {

struct RefineJob {
    int n, model;
    const double* P;      // [n][3] world points (constant)
    const double* uv;     // [n][2]
    double intr[8];
    double q[4], t[3];    // initial pose
    double pad;
};
struct RefineOpt { int max_it; double ftol, ptol, gtol, radius0, huber_a; };
struct RefineResult {
    double q[4], t[3];
    double initial_cost, final_cost;
    int n_successful, n_unsuccessful, termination, reason, attempted, pad;
};

constexpr int kRefVals = 28;   // sum rho | H upper 6x6 row-major (21) | g (6)

// Totals over the correspondences at pose (q, t); every thread returns with tot[] valid.
__device__ __forceinline__ void refine_eval(const RefineJob& job, const double q[4], const double t[3], double huber_a,
                                            double* __restrict__ lds /* [waves][kRefVals] */, double* __restrict__ tot /* shared [kRefVals] */) {
    double acc[kRefVals];
#pragma unroll
    for (int k = 0; k < kRefVals; ++k) acc[k] = 0.0;
    double M[9];
    quat_to_mat(q, M);
    for (int i = threadIdx.x; i < job.n; i += blockDim.x) {
        const double Pw[3] = {job.P[3 * (size_t)i], job.P[3 * (size_t)i + 1], job.P[3 * (size_t)i + 2]};
        Proj pr;
        project<true>(M, t, job.intr, job.model, Pw, job.uv[2 * (size_t)i], job.uv[2 * (size_t)i + 1], pr);
        double rho1;
        acc[0] += huber(pr.r0 * pr.r0 + pr.r1 * pr.r1, huber_a, rho1);
        const double sw = sqrt(rho1);
        const double r[2] = {pr.r0 * sw, pr.r1 * sw};
        double F[12];
#pragma unroll
        for (int row = 0; row < 2; ++row) {
            const double* j = pr.jp + 3 * row;
            F[6 * row + 0] = -2.0 * (j[1] * pr.rp[2] - j[2] * pr.rp[1]) * sw;
            F[6 * row + 1] = -2.0 * (j[2] * pr.rp[0] - j[0] * pr.rp[2]) * sw;
            F[6 * row + 2] = -2.0 * (j[0] * pr.rp[1] - j[1] * pr.rp[0]) * sw;
            F[6 * row + 3] = j[0] * sw;
            F[6 * row + 4] = j[1] * sw;
            F[6 * row + 5] = j[2] * sw;
        }
        int k = 1;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[k++] += F[a] * F[b] + F[6 + a] * F[6 + b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[22 + a] += F[a] * r[0] + F[6 + a] * r[1];
    }
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < kRefVals; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if (lane == 0) lds[wave * kRefVals + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < kRefVals) {
        double v = 0.0;
        for (int w = 0; w < nw; ++w) v += lds[w * kRefVals + threadIdx.x];     // fixed order
        tot[threadIdx.x] = v;
    }
    __syncthreads();
}

__device__ __forceinline__ int ref_idx(int a, int b) {     // position of (a, b), a <= b, in the row-major upper triangle
    return a * 6 - a * (a - 1) / 2 + (b - a);
}

// max-norm of x - Plus(x, -g): Ceres' gradient test for a block with a local parameterisation
__device__ __forceinline__ double refine_gradmax(const double q[4], const double* g) {
    const double dl[3] = {-g[0], -g[1], -g[2]};
    double qn[4];
    quat_plus(q, dl, qn);
    double m = 0.0;
    for (int k = 0; k < 4; ++k) m = fmax(m, fabs(q[k] - qn[k]));
    for (int k = 0; k < 3; ++k) m = fmax(m, fabs(g[3 + k]));
    return m;
}

__global__ __launch_bounds__(kBlock) void k_refine_pose(const RefineJob* __restrict__ jobs, RefineOpt opt, RefineResult* __restrict__ results) {
    __shared__ double lds[(kBlock / kWave) * kRefVals];
    __shared__ double tot[kRefVals];
    __shared__ double sq[4], st[3];          // pose to evaluate next
    __shared__ int sh_go;                    // 1: evaluate (sq, st) and continue; 0: finished
    const RefineJob& job = jobs[blockIdx.x];
    // lane 0 state.  The arrays the LM step indexes with loop variables live in LDS, not in private memory (round 6): as private
    // arrays they made this the library's only kernel with a scratch segment (304 bytes per lane), and the runtime re-allocates a
    // queue's scratch when a dispatch needs it after a large launch of another kernel with scratch has released it (the device-side
    // packing's radix sorts): 20-28 ms on the first pose refinement after every KGBA of a map of 140+ frames (mapper replay,
    // tools/runs/r06_call12.sh: device idle, kernel 0.1 ms, 24 ms between the launch and the end of hipStreamSynchronize).
    __shared__ double H[21], g[6], S[6], delta[6], A[36], b[6];
    double q[4], t[3], cost = 0.0, gmax = 0.0, radius = opt.radius0, decrease = 2.0;
    double qc[4], tc[3], model = 0.0;
    int it = 0, invalid = 0, n_succ = 0, n_unsucc = 0, attempted = 0, term = 0, reason = 0;
    if (threadIdx.x == 0) {
        for (int k = 0; k < 4; ++k) { q[k] = job.q[k]; sq[k] = q[k]; }
        for (int k = 0; k < 3; ++k) { t[k] = job.t[k]; st[k] = t[k]; }
    }
    __syncthreads();
    {
        const double q0[4] = {sq[0], sq[1], sq[2], sq[3]}, t0[3] = {st[0], st[1], st[2]};
        refine_eval(job, q0, t0, opt.huber_a, lds, tot);
    }
    double initial_cost = 0.0;
    bool have_step = false;
    if (threadIdx.x == 0) {
        cost = 0.5 * tot[0]; initial_cost = cost;
        for (int k = 0; k < 21; ++k) H[k] = tot[1 + k];
        for (int k = 0; k < 6; ++k) g[k] = tot[22 + k];
        for (int k = 0; k < 6; ++k) S[k] = 1.0 / (1.0 + sqrt(H[ref_idx(k, k)]));      // Jacobi scaling, fixed from here on
        gmax = refine_gradmax(q, g);
        if (gmax <= opt.gtol) { term = XRSFM_BA_CONVERGENCE; reason = 1; sh_go = 0; }
        else sh_go = -1;      // enter the loop
    }
    __syncthreads();
    while (sh_go != 0) {
        __syncthreads();                     // (everybody has read sh_go)
        if (threadIdx.x == 0) {
            // ---- propose steps until one is valid (or the iteration / invalid-step limits end the solve)
            have_step = false;
            while (!have_step) {
                if (it >= opt.max_it) { term = XRSFM_BA_NO_CONVERGENCE; reason = 5; break; }
                ++it; ++attempted;
                // (S H S + D^2) y = -S g,  D^2 = clamp(diag(S H S), 1e-6, 1e32) / radius
                for (int a = 0; a < 6; ++a) {
                    for (int c2 = a; c2 < 6; ++c2) { const double v = H[ref_idx(a, c2)] * S[a] * S[c2]; A[a * 6 + c2] = v; A[c2 * 6 + a] = v; }
                    b[a] = -S[a] * g[a];
                }
                for (int a = 0; a < 6; ++a) A[a * 6 + a] += fmin(fmax(A[a * 6 + a], 1e-6), 1e32) / radius;
                bool ok = true;
                for (int j = 0; j < 6 && ok; ++j) {          // Cholesky, lower triangle in place
                    double d = A[j * 6 + j];
                    for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
                    if (!(d > 0.0)) { ok = false; break; }
                    d = sqrt(d);
                    A[j * 6 + j] = d;
                    for (int i = j + 1; i < 6; ++i) {
                        double v = A[i * 6 + j];
                        for (int k = 0; k < j; ++k) v -= A[i * 6 + k] * A[j * 6 + k];
                        A[i * 6 + j] = v / d;
                    }
                }
                model = -1.0;
                if (ok) {
                    for (int i = 0; i < 6; ++i) { double v = b[i]; for (int k = 0; k < i; ++k) v -= A[i * 6 + k] * b[k]; b[i] = v / A[i * 6 + i]; }
                    for (int i = 5; i >= 0; --i) { double v = b[i]; for (int k = i + 1; k < 6; ++k) v -= A[k * 6 + i] * b[k]; b[i] = v / A[i * 6 + i]; }
                    double gd = 0.0, dHd = 0.0;
                    for (int a = 0; a < 6; ++a) delta[a] = S[a] * b[a];
                    for (int a = 0; a < 6; ++a) {
                        gd += g[a] * delta[a];
                        double hv = 0.0;
                        for (int c2 = 0; c2 < 6; ++c2) hv += H[a <= c2 ? ref_idx(a, c2) : ref_idx(c2, a)] * delta[c2];
                        dHd += delta[a] * hv;
                    }
                    model = -(gd + 0.5 * dHd);
                }
                if (!(model > 0.0) || !isfinite(model)) {
                    ++invalid; ++n_unsucc;
                    if (invalid >= 5) { term = XRSFM_BA_FAILURE; reason = 6; break; }
                    radius /= decrease; decrease *= 2.0;
                    continue;
                }
                invalid = 0;
                quat_plus(q, delta, qc);
                for (int k = 0; k < 3; ++k) tc[k] = t[k] + delta[3 + k];
                have_step = true;
            }
            if (have_step) {
                for (int k = 0; k < 4; ++k) sq[k] = qc[k];
                for (int k = 0; k < 3; ++k) st[k] = tc[k];
                sh_go = 1;
            } else sh_go = 0;
        }
        __syncthreads();
        if (sh_go == 0) break;
        {
            const double q1[4] = {sq[0], sq[1], sq[2], sq[3]}, t1[3] = {st[0], st[1], st[2]};
            refine_eval(job, q1, t1, opt.huber_a, lds, tot);      // cost AND linearisation at the candidate
        }
        if (threadIdx.x == 0) {
            const double cost_c = 0.5 * tot[0];
            double xn2 = 0.0, st2 = 0.0;
            for (int k = 0; k < 4; ++k) { xn2 += q[k] * q[k]; const double d = qc[k] - q[k]; st2 += d * d; }
            for (int k = 0; k < 3; ++k) { xn2 += t[k] * t[k]; const double d = tc[k] - t[k]; st2 += d * d; }
            const double step_norm = sqrt(st2), xnorm = sqrt(xn2);
            const double change = cost - cost_c;
            if (step_norm <= opt.ptol * (xnorm + opt.ptol)) { term = XRSFM_BA_CONVERGENCE; reason = 2; sh_go = 0; }
            else if (fabs(change) <= opt.ftol * cost) { term = XRSFM_BA_CONVERGENCE; reason = 3; sh_go = 0; }
            else {
                const double rel = change / model;
                if (rel > 1e-3) {
                    for (int k = 0; k < 4; ++k) q[k] = qc[k];
                    for (int k = 0; k < 3; ++k) t[k] = tc[k];
                    cost = cost_c;
                    for (int k = 0; k < 21; ++k) H[k] = tot[1 + k];
                    for (int k = 0; k < 6; ++k) g[k] = tot[22 + k];
                    gmax = refine_gradmax(q, g);
                    radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
                    decrease = 2.0;
                    ++n_succ;
                    if (gmax <= opt.gtol) { term = XRSFM_BA_CONVERGENCE; reason = 1; sh_go = 0; }
                } else {
                    radius /= decrease; decrease *= 2.0;
                    ++n_unsucc;
                    if (radius < 1e-32) { term = XRSFM_BA_CONVERGENCE; reason = 4; sh_go = 0; }
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        RefineResult r;
        for (int k = 0; k < 4; ++k) r.q[k] = q[k];
        for (int k = 0; k < 3; ++k) r.t[k] = t[k];
        r.initial_cost = initial_cost; r.final_cost = cost;
        r.n_successful = n_succ; r.n_unsuccessful = n_unsucc; r.termination = term; r.reason = reason; r.attempted = attempted; r.pad = 0;
        results[blockIdx.x] = r;          // (device memory or — the library's own call — pinned host memory: no copy-engine round trip)
        __threadfence_system();
    }
}

}  // namespace xba
