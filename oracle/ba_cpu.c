/*
 * ba_cpu.c — plain C (C99 + OpenMP) restatement of XRSfM's BA path: the CPU
 * baseline timed by bench.py and the at-scale checker of the tests.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under xrsfm_amd/ links or calls this file.
 *
 * PARITY UNPINNED against real Ceres (Ceres is not vendored under
 * /root/reference and not installed here; the reference ships no golden
 * vectors — SURVEY.md 8c).  This file is pinned to oracle/ba_oracle.py by
 * tests/test_oracle_golden.py; ba_oracle.py's Jacobians are pinned to autodiff of
 * the reference functor.
 *
 * What it restates (same structure as Ceres' SPARSE_SCHUR path the reference
 * selects at /root/reference/src/optimization/ba_solver.cc:74-75):
 *   residual / camera models   cost_factor_ceres.h:19-40, camera_model.hpp:57-209
 *   Huber(5.99), quaternion +  ba_solver.cc:343,353-354
 *   LM loop                    SURVEY.md A.5/A.6 (TrustRegionMinimizer, LevenbergMarquardtStrategy)
 *   linear solve               exact Schur complement (A.7): 3x3 point blocks eliminated, reduced
 *                              camera matrix in block-envelope storage, Cholesky, back-substitution
 *   bal9 mode (round 3)        camera blocks of width CW = 9 {rotation 3, translation 3, f, k1, k2} as soon as one camera
 *                              keeps its intrinsics variable (cam_const bit 2, extension camera model 5; the reference
 *                              never frees intrinsics, ba_solver.cc:602-606): the same code with CW instead of 6; pinned to
 *                              oracle/ba_oracle.py's bal9 branch by tests/test_oracle_golden.py
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/xrsfm_ba.h"

typedef struct {
    int Nc, Np, No;
    int CW;                     /* unknowns per camera block: 6, or 9 in bal9 mode */
    const xrsfm_ba_problem* p;
    double *K, *K2;             /* intrinsics entries [n_intr][8]: current and candidate (only {f,k1,k2} of variable cameras move) */
    int* pt_ptr;   /* CSR by point: obs ids sorted by (point, camera) */
    int* pt_obs;
    int* cam_ptr;  /* CSR by camera */
    int* cam_obs;
    int* rank_in_track; /* position of an obs inside its track list */
    double *q, *t, *P;          /* current state */
    double *q2, *t2, *P2;       /* candidate */
    double *rt, *F, *E;         /* per obs: 2, 2*CW, 6 (scaled, robustified) */
    double *W, *WH;             /* per obs: 3*CW each */
    double *Hpp, *gp, *Hinv;    /* per point: 9 (full), 3, 9 */
    double *Hcc, *gc;           /* per cam: CW*CW, CW */
    double *sc_c, *sc_p;
    double *yc, *yp;
    int* first;                 /* block envelope: first column block of each block row */
    size_t* rowoff;             /* scalar row offsets into env */
    double* env; size_t env_n;
    double* b;
    double huber_a;
} Ctx;

static void quat_to_mat(const double* q, double* M) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    M[0] = 1 - 2 * (y * y + z * z); M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
    M[3] = 2 * (x * y + w * z); M[4] = 1 - 2 * (x * x + z * z); M[5] = 2 * (y * z - w * x);
    M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = 1 - 2 * (x * x + y * y);
}

/* residual r[2], d r / d Pc (jp 2x3) and rp = M P; returns 1 if clamped (z < 1e-2) */
static int project(const double* M, const double* t, const double* k, int model, const double* P, const double* uv,
                   double* r, double* jp, double* rp, int want_jac, double* ji) {
    rp[0] = M[0] * P[0] + M[1] * P[1] + M[2] * P[2];
    rp[1] = M[3] * P[0] + M[4] * P[1] + M[5] * P[2];
    rp[2] = M[6] * P[0] + M[7] * P[1] + M[8] * P[2];
    const double X = rp[0] + t[0], Y = rp[1] + t[1], Z = rp[2] + t[2];
    if (Z < 1e-2) {
        r[0] = 12.0; r[1] = 12.0;
        if (want_jac) memset(jp, 0, 6 * sizeof(double));
        if (ji) memset(ji, 0, 6 * sizeof(double));
        return 1;
    }
    const double iz = 1.0 / Z, xn = X * iz, yn = Y * iz, r2 = xn * xn + yn * yn;
    double fx, fy, cx, cy, du, dv, D00 = 1, D01 = 0, D10 = 0, D11 = 1;
    if (model == 0) { fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; du = xn; dv = yn; D00 = 2; D11 = 2; }
    else if (model == 1) { fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; du = xn; dv = yn; D00 = 2; D11 = 2; }
    else if (model == 2 || model == 3) {
        double kk;
        if (model == 2) { fx = k[0]; fy = k[0]; cx = k[1]; cy = k[2]; kk = k[3]; }
        else { fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; kk = k[4]; }
        const double rad = kk * r2;
        du = xn * rad; dv = yn * rad;
        D00 = 1 + rad + 2 * kk * xn * xn; D11 = 1 + rad + 2 * kk * yn * yn; D01 = 2 * kk * xn * yn; D10 = D01;
    } else if (model == 5) {     /* extension: f (1 + k1 r2 + k2 r2^2) (x, y), no principal point */
        fx = k[0]; fy = k[0]; cx = 0.0; cy = 0.0;
        const double k1 = k[1], k2 = k[2], rad = k1 * r2 + k2 * r2 * r2;
        du = xn * rad; dv = yn * rad;
        const double rad_x = 2 * k1 * xn + 4 * k2 * r2 * xn, rad_y = 2 * k1 * yn + 4 * k2 * r2 * yn;
        D00 = 1 + rad + xn * rad_x; D01 = xn * rad_y; D10 = yn * rad_x; D11 = 1 + rad + yn * rad_y;
    } else {
        fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
        const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7];
        const double xy = xn * yn, x2 = xn * xn, y2 = yn * yn, rad = k1 * r2 + k2 * r2 * r2;
        du = xn * rad + 2 * p1 * xy + p2 * (r2 + 2 * x2);
        dv = yn * rad + 2 * p2 * xy + p1 * (r2 + 2 * y2);
        const double rad_x = 2 * k1 * xn + 4 * k2 * r2 * xn, rad_y = 2 * k1 * yn + 4 * k2 * r2 * yn;
        D00 = 1 + rad + xn * rad_x + 2 * p1 * yn + 6 * p2 * xn;
        D01 = xn * rad_y + 2 * p1 * xn + 2 * p2 * yn;
        D10 = yn * rad_x + 2 * p2 * yn + 2 * p1 * xn;
        D11 = 1 + rad + yn * rad_y + 2 * p2 * xn + 6 * p1 * yn;
    }
    r[0] = fx * (xn + du) + cx - uv[0];
    r[1] = fy * (yn + dv) + cy - uv[1];
    if (want_jac) {
        const double A00 = fx * D00, A01 = fx * D01, A10 = fy * D10, A11 = fy * D11;
        jp[0] = A00 * iz; jp[1] = A01 * iz; jp[2] = -(A00 * xn + A01 * yn) * iz;
        jp[3] = A10 * iz; jp[4] = A11 * iz; jp[5] = -(A10 * xn + A11 * yn) * iz;
    }
    if (ji) {                    /* d r / d (f, k1, k2): model 5 only */
        if (model == 5) {
            ji[0] = xn + du; ji[1] = fx * xn * r2; ji[2] = fx * xn * r2 * r2;
            ji[3] = yn + dv; ji[4] = fy * yn * r2; ji[5] = fy * yn * r2 * r2;
        } else memset(ji, 0, 6 * sizeof(double));
    }
    return 0;
}

static double huber(double s, double a, double* rho1) {
    const double b = a * a;
    if (s > b) { const double r = sqrt(s); *rho1 = fmax(2.2250738585072014e-308, a / r); return 2 * a * r - b; }
    *rho1 = 1.0;
    return s;
}

static void quat_plus(const double* q, const double* d, double* out) {
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    if (n > 0) {
        const double s = sin(n) / n, ax = s * d[0], ay = s * d[1], az = s * d[2], aw = cos(n);
        out[3] = aw * q[3] - (ax * q[0] + ay * q[1] + az * q[2]);
        out[0] = aw * q[0] + q[3] * ax + (ay * q[2] - az * q[1]);
        out[1] = aw * q[1] + q[3] * ay + (az * q[0] - ax * q[2]);
        out[2] = aw * q[2] + q[3] * az + (ax * q[1] - ay * q[0]);
    } else memcpy(out, q, 4 * sizeof(double));
}

/* cost (and, if want_jac, rt/F/E with the current scaling) at state (q,t,P) */
static double evaluate(Ctx* c, const double* q, const double* t, const double* P, const double* K, int want_jac) {
    const xrsfm_ba_problem* p = c->p;
    const int CW = c->CW;
    double cost = 0.0;
#pragma omp parallel for reduction(+ : cost) schedule(static)
    for (int i = 0; i < c->No; ++i) {
        const int cam = p->obs_cam[i], pt = p->obs_pt[i];
        const int ii = p->cam_intr[cam];
        double M[9], r[2], jp[6], rp[3], ji[6];
        quat_to_mat(q + 4 * (size_t)cam, M);
        project(M, t + 3 * (size_t)cam, K + 8 * (size_t)ii, p->intr_model[ii], P + 3 * (size_t)pt,
                p->obs_uv + 2 * (size_t)i, r, jp, rp, want_jac, (want_jac && CW == 9) ? ji : NULL);
        double rho1;
        cost += huber(r[0] * r[0] + r[1] * r[1], c->huber_a, &rho1);
        if (!want_jac) continue;
        const double sw = sqrt(rho1);
        const unsigned cc = p->cam_const ? p->cam_const[cam] : 0u;
        const double mq = (cc & 1u) ? 0.0 : sw, mt = (cc & 2u) ? 0.0 : sw, mi = (cc & 4u) ? sw : 0.0;
        const double mp = (p->point_const && p->point_const[pt]) ? 0.0 : sw;
        const double* sc = c->sc_c + CW * (size_t)cam;
        const double* sp = c->sc_p + 3 * (size_t)pt;
        double* F = c->F + 2 * CW * (size_t)i;
        double* E = c->E + 6 * (size_t)i;
        c->rt[2 * (size_t)i] = r[0] * sw; c->rt[2 * (size_t)i + 1] = r[1] * sw;
        for (int row = 0; row < 2; ++row) {
            const double* j = jp + 3 * row;
            F[CW * row + 0] = -2.0 * (j[1] * rp[2] - j[2] * rp[1]) * mq * sc[0];
            F[CW * row + 1] = -2.0 * (j[2] * rp[0] - j[0] * rp[2]) * mq * sc[1];
            F[CW * row + 2] = -2.0 * (j[0] * rp[1] - j[1] * rp[0]) * mq * sc[2];
            F[CW * row + 3] = j[0] * mt * sc[3]; F[CW * row + 4] = j[1] * mt * sc[4]; F[CW * row + 5] = j[2] * mt * sc[5];
            if (CW == 9) for (int a = 0; a < 3; ++a) F[CW * row + 6 + a] = ji[3 * row + a] * mi * sc[6 + a];
            E[3 * row + 0] = (j[0] * M[0] + j[1] * M[3] + j[2] * M[6]) * mp * sp[0];
            E[3 * row + 1] = (j[0] * M[1] + j[1] * M[4] + j[2] * M[7]) * mp * sp[1];
            E[3 * row + 2] = (j[0] * M[2] + j[1] * M[5] + j[2] * M[8]) * mp * sp[2];
        }
    }
    return 0.5 * cost;
}

/* normal-equation blocks from rt/F/E: Hpp, gp (per point), Hcc, gc (per camera) */
static void build_blocks(Ctx* c) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < c->Np; ++j) {
        double H[9] = {0}, g[3] = {0};
        for (int k = c->pt_ptr[j]; k < c->pt_ptr[j + 1]; ++k) {
            const int i = c->pt_obs[k];
            const double* E = c->E + 6 * (size_t)i;
            const double* r = c->rt + 2 * (size_t)i;
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) H[3 * a + b] += E[a] * E[b] + E[3 + a] * E[3 + b];
                g[a] += E[a] * r[0] + E[3 + a] * r[1];
            }
        }
        memcpy(c->Hpp + 9 * (size_t)j, H, sizeof H);
        memcpy(c->gp + 3 * (size_t)j, g, sizeof g);
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int cam = 0; cam < c->Nc; ++cam) {
        const int CW = c->CW;
        double H[81] = {0}, g[9] = {0};
        for (int k = c->cam_ptr[cam]; k < c->cam_ptr[cam + 1]; ++k) {
            const int i = c->cam_obs[k];
            const double* F = c->F + 2 * CW * (size_t)i;
            const double* r = c->rt + 2 * (size_t)i;
            for (int a = 0; a < CW; ++a) {
                for (int b = 0; b < CW; ++b) H[CW * a + b] += F[a] * F[b] + F[CW + a] * F[CW + b];
                g[a] += F[a] * r[0] + F[CW + a] * r[1];
            }
        }
        memcpy(c->Hcc + CW * CW * (size_t)cam, H, CW * CW * sizeof(double));
        memcpy(c->gc + CW * (size_t)cam, g, CW * sizeof(double));
    }
}

static void inv3(const double* h, double* inv) {
    const double a = h[0], b = h[1], cc = h[2], d = h[4], e = h[5], f = h[8];
    const double c00 = d * f - e * e, c01 = cc * e - b * f, c02 = b * e - cc * d;
    const double id = 1.0 / (a * c00 + b * c01 + cc * c02);
    inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
    inv[3] = inv[1]; inv[4] = (a * f - cc * cc) * id; inv[5] = (b * cc - a * e) * id;
    inv[6] = inv[2]; inv[7] = inv[5]; inv[8] = (a * d - b * b) * id;
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Solve (Js^T Js + D^2) y = Js^T r by exact Schur elimination; returns 0 on success */
static int solve_step(Ctx* c, double radius) {
    const int Nc = c->Nc, Np = c->Np, CW = c->CW;
#pragma omp parallel for schedule(static)
    for (int j = 0; j < Np; ++j) {
        double H[9];
        memcpy(H, c->Hpp + 9 * (size_t)j, sizeof H);
        for (int a = 0; a < 3; ++a) H[4 * a] += clampd(H[4 * a], 1e-6, 1e32) / radius;
        inv3(H, c->Hinv + 9 * (size_t)j);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < c->No; ++i) {
        const double* F = c->F + 2 * CW * (size_t)i;
        const double* E = c->E + 6 * (size_t)i;
        const double* Hi = c->Hinv + 9 * (size_t)c->p->obs_pt[i];
        double* W = c->W + 3 * CW * (size_t)i;
        double* WH = c->WH + 3 * CW * (size_t)i;
        for (int a = 0; a < CW; ++a) {
            for (int b = 0; b < 3; ++b) W[3 * a + b] = F[a] * E[b] + F[CW + a] * E[3 + b];
            for (int b = 0; b < 3; ++b) WH[3 * a + b] = W[3 * a] * Hi[b] + W[3 * a + 1] * Hi[3 + b] + W[3 * a + 2] * Hi[6 + b];
        }
    }
    memset(c->env, 0, c->env_n * sizeof(double));
    const int n = CW * Nc;
    /* rows are owned by their camera: no write conflicts */
#pragma omp parallel for schedule(dynamic, 4)
    for (int cam = 0; cam < Nc; ++cam) {
        double bl[9];
        for (int a = 0; a < CW; ++a) bl[a] = c->gc[CW * (size_t)cam + a];
        const int fc = CW * c->first[cam];
        /* diagonal block: Hcc + D^2 */
        for (int a = 0; a < CW; ++a) {
            double* row = c->env + c->rowoff[CW * cam + a] - fc;
            for (int b = 0; b <= a; ++b) row[CW * cam + b] += c->Hcc[CW * CW * (size_t)cam + CW * a + b];
            row[CW * cam + a] += clampd(c->Hcc[CW * CW * (size_t)cam + (CW + 1) * a], 1e-6, 1e32) / radius;
        }
        for (int k = c->cam_ptr[cam]; k < c->cam_ptr[cam + 1]; ++k) {
            const int i = c->cam_obs[k];
            const int j = c->p->obs_pt[i];
            const double* WHi = c->WH + 3 * CW * (size_t)i;
            const double* g = c->gp + 3 * (size_t)j;
            for (int a = 0; a < CW; ++a) bl[a] -= WHi[3 * a] * g[0] + WHi[3 * a + 1] * g[1] + WHi[3 * a + 2] * g[2];
            for (int kk = c->pt_ptr[j]; kk < c->pt_ptr[j + 1]; ++kk) {
                const int i2 = c->pt_obs[kk];
                const int cam2 = c->p->obs_cam[i2];
                if (cam2 > cam) continue;
                const double* W2 = c->W + 3 * CW * (size_t)i2;
                for (int a = 0; a < CW; ++a) {
                    double* row = c->env + c->rowoff[CW * cam + a] - fc;
                    const int bmax = (cam2 == cam) ? a : CW - 1;
                    for (int b = 0; b <= bmax; ++b)
                        row[CW * cam2 + b] -= WHi[3 * a] * W2[3 * b] + WHi[3 * a + 1] * W2[3 * b + 1] + WHi[3 * a + 2] * W2[3 * b + 2];
                }
            }
        }
        for (int a = 0; a < CW; ++a) c->b[CW * (size_t)cam + a] = bl[a];
    }
    /* envelope Cholesky, row by row: L[r][col] for col in [fc_r, r] */
    for (int r = 0; r < n; ++r) {
        const int fr = CW * c->first[r / CW];
        double* Lr = c->env + c->rowoff[r] - fr;
        for (int col = fr; col <= r; ++col) {
            const int fcol = CW * c->first[col / CW];
            const double* Lc = c->env + c->rowoff[col] - fcol;
            const int k0 = fr > fcol ? fr : fcol;
            double s = Lr[col];
            for (int k = k0; k < col; ++k) s -= Lr[k] * Lc[k];
            if (col < r) Lr[col] = s / Lc[col];
            else { if (!(s > 0.0)) return 1; Lr[r] = sqrt(s); }
        }
    }
    double* y = c->yc;
    for (int r = 0; r < n; ++r) {       /* forward: L z = b */
        const int fr = CW * c->first[r / CW];
        const double* Lr = c->env + c->rowoff[r] - fr;
        double s = c->b[r];
        for (int k = fr; k < r; ++k) s -= Lr[k] * y[k];
        y[r] = s / Lr[r];
    }
    for (int r = n - 1; r >= 0; --r) {  /* backward: L^T y = z (column sweep) */
        const int fr = CW * c->first[r / CW];
        const double* Lr = c->env + c->rowoff[r] - fr;
        y[r] /= Lr[r];
        const double v = y[r];
        for (int k = fr; k < r; ++k) y[k] -= Lr[k] * v;
    }
    /* back-substitute the points: y_p = Hinv (g_p - sum W^T y_c) */
#pragma omp parallel for schedule(static)
    for (int j = 0; j < Np; ++j) {
        double a[3] = {c->gp[3 * (size_t)j], c->gp[3 * (size_t)j + 1], c->gp[3 * (size_t)j + 2]};
        for (int k = c->pt_ptr[j]; k < c->pt_ptr[j + 1]; ++k) {
            const int i = c->pt_obs[k];
            const double* W = c->W + 3 * CW * (size_t)i;
            const double* yc = c->yc + CW * (size_t)c->p->obs_cam[i];
            for (int m = 0; m < CW; ++m) { a[0] -= W[3 * m] * yc[m]; a[1] -= W[3 * m + 1] * yc[m]; a[2] -= W[3 * m + 2] * yc[m]; }
        }
        const double* Hi = c->Hinv + 9 * (size_t)j;
        for (int m = 0; m < 3; ++m) c->yp[3 * (size_t)j + m] = Hi[3 * m] * a[0] + Hi[3 * m + 1] * a[1] + Hi[3 * m + 2] * a[2];
    }
    return 0;
}

typedef struct {
    double initial_cost, final_cost;
    int n_successful, n_unsuccessful, termination, reason, num_effective_params;
    double linearize_s, solve_s, total_s;
} CpuSummary;

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return (double)clock() / CLOCKS_PER_SEC;
#endif
}

static int cmp_int_pair(const void* a, const void* b) {
    const int* x = (const int*)a; const int* y = (const int*)b;
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    return (x[1] > y[1]) - (x[1] < y[1]);
}

int ba_cpu_solve(const xrsfm_ba_options* opt, xrsfm_ba_problem* p, CpuSummary* sum, int threads) {
    if (!opt || !p || !sum) return -1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    const double t_begin = now_s();
    memset(sum, 0, sizeof *sum);
    Ctx c; memset(&c, 0, sizeof c);
    c.p = p; c.Nc = p->n_cams; c.Np = p->n_points; c.No = p->n_obs; c.huber_a = opt->huber_a;
    const int Nc = c.Nc, Np = c.Np, No = c.No;
    for (int i = 0; i < No; ++i)
        if (p->obs_cam[i] < 0 || p->obs_cam[i] >= Nc || p->obs_pt[i] < 0 || p->obs_pt[i] >= Np) return -1;
    /* bal9 mode: some camera keeps {f, k1, k2} variable (bit 2): model 5 with an intrinsics entry of its own */
    c.CW = 6;
    for (int k = 0; k < Nc; ++k)
        if (p->cam_const && (p->cam_const[k] & 4u)) {
            if (p->intr_model[p->cam_intr[k]] != 5) return -1;
            for (int k2 = 0; k2 < Nc; ++k2) if (k2 != k && p->cam_intr[k2] == p->cam_intr[k]) return -1;
            c.CW = 9;
        }
    const int CW = c.CW;
#define ALLOC(ptr, n) do { (ptr) = calloc((n) > 0 ? (n) : 1, sizeof *(ptr)); if (!(ptr)) return -3; } while (0)
    ALLOC(c.pt_ptr, (size_t)Np + 1); ALLOC(c.pt_obs, No); ALLOC(c.cam_ptr, (size_t)Nc + 1); ALLOC(c.cam_obs, No);
    {   /* CSR by point with obs sorted by camera, CSR by camera */
        int* pairs; ALLOC(pairs, 2 * (size_t)No);
        for (int i = 0; i < No; ++i) c.pt_ptr[p->obs_pt[i] + 1]++;
        for (int j = 0; j < Np; ++j) c.pt_ptr[j + 1] += c.pt_ptr[j];
        int* fill; ALLOC(fill, (size_t)Np + 1);
        memcpy(fill, c.pt_ptr, ((size_t)Np + 1) * sizeof(int));
        for (int i = 0; i < No; ++i) { const int k = fill[p->obs_pt[i]]++; pairs[2 * k] = p->obs_cam[i]; pairs[2 * k + 1] = i; }
        for (int j = 0; j < Np; ++j) qsort(pairs + 2 * (size_t)c.pt_ptr[j], c.pt_ptr[j + 1] - c.pt_ptr[j], 2 * sizeof(int), cmp_int_pair);
        for (int k = 0; k < No; ++k) c.pt_obs[k] = pairs[2 * k + 1];
        free(pairs); free(fill);
        for (int i = 0; i < No; ++i) c.cam_ptr[p->obs_cam[i] + 1]++;
        for (int k = 0; k < Nc; ++k) c.cam_ptr[k + 1] += c.cam_ptr[k];
        int* cf; ALLOC(cf, (size_t)Nc + 1);
        memcpy(cf, c.cam_ptr, ((size_t)Nc + 1) * sizeof(int));
        for (int k = 0; k < No; ++k) { const int i = c.pt_obs[k]; c.cam_obs[cf[p->obs_cam[i]]++] = i; }
        free(cf);
    }
    /* block envelope of the reduced camera matrix (natural ordering) */
    ALLOC(c.first, Nc);
    for (int k = 0; k < Nc; ++k) c.first[k] = k;
    for (int j = 0; j < Np; ++j) {
        if (c.pt_ptr[j + 1] == c.pt_ptr[j]) continue;
        const int cmin = p->obs_cam[c.pt_obs[c.pt_ptr[j]]];
        for (int k = c.pt_ptr[j]; k < c.pt_ptr[j + 1]; ++k) {
            const int cam = p->obs_cam[c.pt_obs[k]];
            if (cmin < c.first[cam]) c.first[cam] = cmin;
        }
    }
    ALLOC(c.rowoff, (size_t)CW * Nc + 1);
    size_t off = 0;
    for (int r = 0; r < CW * Nc; ++r) { c.rowoff[r] = off; off += (size_t)(r - CW * c.first[r / CW] + 1); }
    c.env_n = off;
    ALLOC(c.env, c.env_n);
    ALLOC(c.q, (size_t)4 * Nc); ALLOC(c.t, (size_t)3 * Nc); ALLOC(c.P, (size_t)3 * Np);
    ALLOC(c.q2, (size_t)4 * Nc); ALLOC(c.t2, (size_t)3 * Nc); ALLOC(c.P2, (size_t)3 * Np);
    memcpy(c.q, p->cam_q, (size_t)4 * Nc * sizeof(double)); memcpy(c.t, p->cam_t, (size_t)3 * Nc * sizeof(double));
    memcpy(c.P, p->points, (size_t)3 * Np * sizeof(double));
    ALLOC(c.K, (size_t)8 * p->n_intr); ALLOC(c.K2, (size_t)8 * p->n_intr);
    memcpy(c.K, p->intr_params, (size_t)8 * p->n_intr * sizeof(double)); memcpy(c.K2, c.K, (size_t)8 * p->n_intr * sizeof(double));
    ALLOC(c.rt, (size_t)2 * No); ALLOC(c.F, (size_t)2 * CW * No); ALLOC(c.E, (size_t)6 * No);
    ALLOC(c.W, (size_t)3 * CW * No); ALLOC(c.WH, (size_t)3 * CW * No);
    ALLOC(c.Hpp, (size_t)9 * Np); ALLOC(c.gp, (size_t)3 * Np); ALLOC(c.Hinv, (size_t)9 * Np);
    ALLOC(c.Hcc, (size_t)CW * CW * Nc); ALLOC(c.gc, (size_t)CW * Nc);
    ALLOC(c.sc_c, (size_t)CW * Nc); ALLOC(c.sc_p, (size_t)3 * Np);
    ALLOC(c.yc, (size_t)CW * Nc); ALLOC(c.yp, (size_t)3 * Np); ALLOC(c.b, (size_t)CW * Nc);
    /* which blocks are variable */
    unsigned char *qvar, *tvar, *pvar, *ivar;
    ALLOC(qvar, Nc); ALLOC(tvar, Nc); ALLOC(pvar, Np); ALLOC(ivar, Nc);
    for (int k = 0; k < Nc; ++k) {
        const int act = c.cam_ptr[k + 1] > c.cam_ptr[k];
        const unsigned cc = p->cam_const ? p->cam_const[k] : 0u;
        qvar[k] = act && !(cc & 1u); tvar[k] = act && !(cc & 2u); ivar[k] = act && (cc & 4u);
        sum->num_effective_params += 3 * qvar[k] + 3 * tvar[k] + 3 * ivar[k];
    }
    for (int j = 0; j < Np; ++j) {
        pvar[j] = (c.pt_ptr[j + 1] > c.pt_ptr[j]) && !(p->point_const && p->point_const[j]);
        sum->num_effective_params += 3 * pvar[j];
    }
    double t0 = now_s();
    for (int k = 0; k < CW * Nc; ++k) c.sc_c[k] = 1.0;
    for (int k = 0; k < 3 * Np; ++k) c.sc_p[k] = 1.0;
    double cost = evaluate(&c, c.q, c.t, c.P, c.K, 1);
    build_blocks(&c);
    for (int k = 0; k < Nc; ++k) for (int a = 0; a < CW; ++a) c.sc_c[CW * (size_t)k + a] = 1.0 / (1.0 + sqrt(c.Hcc[CW * CW * (size_t)k + (CW + 1) * a]));
    for (int j = 0; j < Np; ++j) for (int a = 0; a < 3; ++a) c.sc_p[3 * (size_t)j + a] = 1.0 / (1.0 + sqrt(c.Hpp[9 * (size_t)j + 4 * a]));
    cost = evaluate(&c, c.q, c.t, c.P, c.K, 1);
    build_blocks(&c);
    sum->linearize_s += now_s() - t0;
    sum->initial_cost = cost;
#define XNORM(qq, tt, PP, KK, out) do { double s_ = 0; \
        for (int k = 0; k < Nc; ++k) { if (qvar[k]) for (int a = 0; a < 4; ++a) s_ += (qq)[4 * (size_t)k + a] * (qq)[4 * (size_t)k + a]; \
                                        if (tvar[k]) for (int a = 0; a < 3; ++a) s_ += (tt)[3 * (size_t)k + a] * (tt)[3 * (size_t)k + a]; \
                                        if (ivar[k]) for (int a = 0; a < 3; ++a) s_ += (KK)[8 * (size_t)p->cam_intr[k] + a] * (KK)[8 * (size_t)p->cam_intr[k] + a]; } \
        for (int j = 0; j < Np; ++j) if (pvar[j]) for (int a = 0; a < 3; ++a) s_ += (PP)[3 * (size_t)j + a] * (PP)[3 * (size_t)j + a]; \
        (out) = sqrt(s_); } while (0)
#define GRADMAX(out) do { double m_ = 0; \
        for (int k = 0; k < Nc; ++k) { const double* g = c.gc + CW * (size_t)k; const double* s = c.sc_c + CW * (size_t)k; \
            if (qvar[k]) { double d[3] = {-g[0] / s[0], -g[1] / s[1], -g[2] / s[2]}, qn[4]; quat_plus(c.q + 4 * (size_t)k, d, qn); \
                for (int a = 0; a < 4; ++a) m_ = fmax(m_, fabs(c.q[4 * (size_t)k + a] - qn[a])); } \
            if (tvar[k]) for (int a = 0; a < 3; ++a) m_ = fmax(m_, fabs(g[3 + a] / s[3 + a])); \
            if (ivar[k]) for (int a = 0; a < 3; ++a) m_ = fmax(m_, fabs(g[6 + a] / s[6 + a])); } \
        for (int j = 0; j < Np; ++j) if (pvar[j]) for (int a = 0; a < 3; ++a) m_ = fmax(m_, fabs(c.gp[3 * (size_t)j + a] / c.sc_p[3 * (size_t)j + a])); \
        (out) = m_; } while (0)
    double gmax; GRADMAX(gmax);
    int term = XRSFM_BA_NO_CONVERGENCE, reason = 5;
    double radius = opt->initial_radius, decrease = 2.0, xnorm;
    XNORM(c.q, c.t, c.P, c.K, xnorm);
    int it = 0, invalid = 0;
    if (gmax <= opt->gradient_tolerance) { term = XRSFM_BA_CONVERGENCE; reason = 1; goto done; }
    while (1) {
        if (it >= opt->max_iterations) { term = XRSFM_BA_NO_CONVERGENCE; reason = 5; break; }
        ++it;
        t0 = now_s();
        const int fail = solve_step(&c, radius);
        sum->solve_s += now_s() - t0;
        double model = 0.0;
        if (!fail) {
#pragma omp parallel for reduction(+ : model) schedule(static)
            for (int i = 0; i < No; ++i) {
                const double* F = c.F + 2 * CW * (size_t)i; const double* E = c.E + 6 * (size_t)i;
                const double* yc = c.yc + CW * (size_t)p->obs_cam[i]; const double* yp = c.yp + 3 * (size_t)p->obs_pt[i];
                double m0 = 0, m1 = 0;
                for (int a = 0; a < CW; ++a) { m0 += F[a] * yc[a]; m1 += F[CW + a] * yc[a]; }
                for (int a = 0; a < 3; ++a) { m0 += E[a] * yp[a]; m1 += E[3 + a] * yp[a]; }
                model += m0 * (c.rt[2 * (size_t)i] - 0.5 * m0) + m1 * (c.rt[2 * (size_t)i + 1] - 0.5 * m1);
            }
        }
        if (fail || !(model > 0.0) || !isfinite(model)) {
            ++invalid; sum->n_unsuccessful++;
            if (invalid >= 5) { term = XRSFM_BA_FAILURE; reason = 6; break; }
            radius /= decrease; decrease *= 2.0;
            continue;
        }
        invalid = 0;
        double step2 = 0.0;
        if (CW == 9) memcpy(c.K2, c.K, (size_t)8 * p->n_intr * sizeof(double));
        for (int k = 0; k < Nc; ++k) {
            const double* y = c.yc + CW * (size_t)k; const double* s = c.sc_c + CW * (size_t)k;
            memcpy(c.q2 + 4 * (size_t)k, c.q + 4 * (size_t)k, 4 * sizeof(double));
            memcpy(c.t2 + 3 * (size_t)k, c.t + 3 * (size_t)k, 3 * sizeof(double));
            if (ivar[k]) for (int a = 0; a < 3; ++a) {
                const size_t m = 8 * (size_t)p->cam_intr[k] + a;
                c.K2[m] = c.K[m] + (-y[6 + a] * s[6 + a]);
                const double df = c.K2[m] - c.K[m]; step2 += df * df;
            }
            if (qvar[k]) {
                double d[3] = {-y[0] * s[0], -y[1] * s[1], -y[2] * s[2]};
                quat_plus(c.q + 4 * (size_t)k, d, c.q2 + 4 * (size_t)k);
                for (int a = 0; a < 4; ++a) { const double df = c.q2[4 * (size_t)k + a] - c.q[4 * (size_t)k + a]; step2 += df * df; }
            }
            if (tvar[k]) for (int a = 0; a < 3; ++a) {
                c.t2[3 * (size_t)k + a] = c.t[3 * (size_t)k + a] + (-y[3 + a] * s[3 + a]);
                const double df = c.t2[3 * (size_t)k + a] - c.t[3 * (size_t)k + a]; step2 += df * df;
            }
        }
        for (int j = 0; j < Np; ++j) for (int a = 0; a < 3; ++a) {
            const size_t m = 3 * (size_t)j + a;
            c.P2[m] = pvar[j] ? c.P[m] + (-c.yp[m] * c.sc_p[m]) : c.P[m];
            const double df = c.P2[m] - c.P[m]; step2 += df * df;
        }
        t0 = now_s();
        const double cost2 = evaluate(&c, c.q2, c.t2, c.P2, c.K2, 0);
        sum->linearize_s += now_s() - t0;
        const double step_norm = sqrt(step2);
        if (step_norm <= opt->parameter_tolerance * (xnorm + opt->parameter_tolerance)) { term = XRSFM_BA_CONVERGENCE; reason = 2; break; }
        const double cost_change = cost - cost2;
        if (fabs(cost_change) <= opt->function_tolerance * cost) { term = XRSFM_BA_CONVERGENCE; reason = 3; break; }
        const double rel = cost_change / model;
        if (rel > 1e-3) {
            double* tmp;
            tmp = c.q; c.q = c.q2; c.q2 = tmp; tmp = c.t; c.t = c.t2; c.t2 = tmp; tmp = c.P; c.P = c.P2; c.P2 = tmp;
            tmp = c.K; c.K = c.K2; c.K2 = tmp;
            XNORM(c.q, c.t, c.P, c.K, xnorm);
            t0 = now_s();
            cost = evaluate(&c, c.q, c.t, c.P, c.K, 1);
            build_blocks(&c);
            sum->linearize_s += now_s() - t0;
            radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0;
            sum->n_successful++;
            GRADMAX(gmax);
            if (gmax <= opt->gradient_tolerance) { term = XRSFM_BA_CONVERGENCE; reason = 1; break; }
        } else {
            radius /= decrease; decrease *= 2.0;
            sum->n_unsuccessful++;
            if (radius < 1e-32) { term = XRSFM_BA_CONVERGENCE; reason = 4; break; }
        }
    }
done:
    sum->termination = term; sum->reason = reason; sum->final_cost = cost;
    memcpy(p->cam_q, c.q, (size_t)4 * Nc * sizeof(double)); memcpy(p->cam_t, c.t, (size_t)3 * Nc * sizeof(double));
    memcpy(p->points, c.P, (size_t)3 * Np * sizeof(double));
    if (CW == 9) memcpy(p->intr_params, c.K, (size_t)8 * p->n_intr * sizeof(double));
    free(c.K); free(c.K2); free(ivar);
    free(c.pt_ptr); free(c.pt_obs); free(c.cam_ptr); free(c.cam_obs); free(c.first); free(c.rowoff); free(c.env);
    free(c.q); free(c.t); free(c.P); free(c.q2); free(c.t2); free(c.P2); free(c.rt); free(c.F); free(c.E); free(c.W); free(c.WH);
    free(c.Hpp); free(c.gp); free(c.Hinv); free(c.Hcc); free(c.gc); free(c.sc_c); free(c.sc_p); free(c.yc); free(c.yp); free(c.b);
    free(qvar); free(tvar); free(pvar);
    sum->total_s = now_s() - t_begin;
    return 0;
}
