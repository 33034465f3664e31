// bal9 mode: 9-wide camera blocks {rotation 3, translation 3, f, k1, k2} (BASELINE.json north_star:
// "LDS-staged 2x9 / 2x3 blocks").  The reference never frees intrinsics (ba_solver.cc:602-606, 655-659, 389), so nothing in
// src/mapper reaches this path; it exists for BAL-style problems where every camera carries its own {f, k1, k2} (extension
// camera model 5, ba_math.h) and is selected per camera by cam_const bit 2 (include/xrsfm_ba.h).
//
// Kernels of this file work per OBSERVATION and per observation PAIR on stored 2x9 / 2x3 Jacobian blocks ("J-stored": the
// accounting of SURVEY 8(d) with 160 -> 208, 144 -> 192, 216 -> 432 bytes) — wave = tile as in ba_kernels.h, segmented wave
// reductions for the point blocks, camera-major scatter + fixed-order segmented sums for the camera side (bit-reproducible,
// no FP atomics) — without the Gram-product / compressed-Jacobian machinery of the 6-wide hot path: correct and
// bandwidth-shaped, not tuned.  The reduced system is solved by the same tile Cholesky (7 cameras x 9 rows per 64-row tile).
#pragma once
#include "ba_chol.h"

namespace xba {

constexpr int kW = 9;                       // unknowns per camera
constexpr int kWS = 56;                     // per-observation diagonal-block record: 45 (upper triangle) + 9 (rhs) + 2 pad
constexpr int kWB = 81;                     // off-diagonal block
#ifndef XBA_K9LIN_W
#define XBA_K9LIN_W 3
#endif
#ifndef XBA_K9G_W
#define XBA_K9G_W 3
#endif

struct DevW {
    double* Fw;        // [18][n_slots] sqrt(rho') d r / d cam (2x9), Jacobi-scaled, constant blocks zero
    double* Ew;        // [6][n_slots]  ... d r / d point (2x3)
    double* scale_c;   // [Nc][9]
    double* camlin;    // [Nc][18]: diag(Hcc) 9, gc 9
    double* camS;      // [Nc][56]
    double* px;        // [Nc][9] (+ one tile of slack): solution of the reduced system, camera order
    double* scat;      // camera-major scatter buffer [n_obs][56]
};

__device__ __forceinline__ void load_FE9(const Dev& d, const DevW& w, int slot, double (&F)[18], double (&E)[6]) {
    const size_t ns = (size_t)d.n_slots;
#pragma unroll
    for (int k = 0; k < 18; ++k) F[k] = w.Fw[k * ns + slot];
#pragma unroll
    for (int k = 0; k < 6; ++k) E[k] = w.Ew[k * ns + slot];
}

// Per-camera sums over the lanes of a Gram tile through the wave's LDS (red: [64][kRedLd]): NV <= 14 values per lane go in —
// deposited SORTED BY CAMERA (tile_camera_runs) — and one lane per (distinct camera c of the tile, value k) adds the contiguous
// run of camera c, i.e. the entries of the lanes whose observation is in camera c in lane order, and stores the sum at
// out[stride * entry(c) + off + k], entry(c) = the camera-major entry held by the camera's first lane (cp).
// Called by all 64 lanes; ends with the wave's LDS reads complete (the buffer may be reused).
template <int NV>
__device__ __forceinline__ void tile_camera_sums(double* red, const double (&v)[NV], int lane, int C, int mypos,
                                                 const int (&run_pk)[2], int cp, double* __restrict__ out, int stride, int off) {
    static_assert(NV <= 14 && NV * kGramMaxCamsWide <= 2 * kWave, "two lane rounds cover every (camera, value) of a tile");
#pragma unroll
    for (int k = 0; k < NV; ++k) red[mypos * kRedLd + k] = v[k];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int nq = NV * C;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        if (rd * kWave < nq) {                             // (uniform)
            const int q = lane + kWave * rd;
            const bool on = q < nq;
            const int pk = rd == 0 ? run_pk[0] : run_pk[1];
            const int cpr = __shfl(cp, on ? (pk >> 16) : 0, kWave);
            if (on) {
                const int k = q % NV, n = (pk >> 8) & 255;
                const double* src = red + (pk & 255) * kRedLd + k;
                double sum = 0.0;
                int j = 0;
                for (; j + 4 <= n; j += 4) {
                    const double a0 = src[j * kRedLd], a1 = src[(j + 1) * kRedLd], a2 = src[(j + 2) * kRedLd], a3 = src[(j + 3) * kRedLd];
                    sum += a0; sum += a1; sum += a2; sum += a3;
                }
                for (; j < n; ++j) sum += src[j * kRedLd];
                out[(size_t)stride * cpr + off + k] = sum;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}
// What tile_camera_sums needs: the lane's position in the camera-sorted order (lanes of earlier cameras + earlier lanes of the
// own camera; a lane without an observation parks in row 63, which is in no run) and, for the (camera, value) slots lane and
// lane + 64, the camera's run: start | length << 8 | first lane << 16 — from the per-context tables of k_gram_runs (ba_kernels.h;
// round 6: until then a ballot loop over the tile's cameras in every launch).
template <int NV>
__device__ __forceinline__ void tile_camera_runs(const Dev& d, int tile, int slot, int lane, int C, int& mypos, int (&run_pk)[2]) {
    const int* trun = d.tile_run + (size_t)tile * kTileRunLd;
    const int c0 = lane / NV, c1 = (lane + kWave) / NV;
    mypos = (int)d.slot_gpos[slot];
    run_pk[0] = c0 < C ? trun[c0] : 0;
    run_pk[1] = c1 < C ? trun[c1] : 0;
}

// ---------------------------------------------------------------- linearise
// (round 4) Gram tiles (ba_pack.h: tile_ncam > 0) sum the 18 camera-side values of their observations per distinct camera of
// the tile through LDS and write ONE entry per camera (slot_campos_g / cam_ptr_g, the entries of the S assembly) instead of one
// per observation: 144 bytes per observation less to write, and the fixed-order sum that follows reads a sixteenth.
// LONG = the item is one track of more than 64 observations spread over several tiles: the sums that carry over its tiles live in
// the wave's LDS slice (a long item is never a Gram tile) and exist only in that instantiation, as in linearize_item (ba_kernels.h).
template <bool LONG>
__device__ __forceinline__ void linearize9_item(const Dev& d, const DevW& w, const Item& it, int item, int lane, double huber_a, double* red) {
    constexpr bool is_long = LONG;
    const size_t ns = (size_t)d.n_slots;
    double* acc = red;                                // long item: sums over its tiles (lane 0)
    if (is_long && lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    }
    double cost = 0.0, xn2 = 0.0, gm = 0.0;
    int long_pt = -1;
    for (int tl = 0; tl < (LONG ? it.n_tiles : 1); ++tl) {
        const SlotCtx s = load_slot(d, it.first_tile + tl, lane);
        const int maxlen = d.tile_maxlen[it.first_tile + tl];
        double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double spk[3] = {1.0, 1.0, 1.0};
        double cs0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, cs1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (s.valid) {
            const CamRec& c = d.cam[s.cam];
            const double q[4] = {c.q[0], c.q[1], c.q[2], c.q[3]};
            const double t[3] = {c.t[0], c.t[1], c.t[2]};
            const double* P = d.P + 3 * (size_t)s.pt;
            const double Pw[3] = {P[0], P[1], P[2]};
            double M[9];
            quat_to_mat(q, M);
            Proj pr;
            project<true, true>(M, t, c.intr, d.cam_model[s.cam], Pw, d.slot_u[s.slot], d.slot_v[s.slot], pr);
            double rho1;
            const double rho = huber(pr.r0 * pr.r0 + pr.r1 * pr.r1, huber_a, rho1);
            cost += rho;
            const double sw = sqrt(rho1);
            const double r0 = pr.r0 * sw, r1 = pr.r1 * sw;
            const unsigned cc = d.cam_const[s.cam];
            const double mq = (cc & 1u) ? 0.0 : sw, mt = (cc & 2u) ? 0.0 : sw, mi = (cc & 4u) ? sw : 0.0;   // bit 2: intrinsics VARIABLE
            const bool pt_fixed = d.pt_const[s.pt] != 0;
            const double mp = pt_fixed ? 0.0 : sw;
            const double* sc = w.scale_c + 9 * (size_t)s.cam;
            const double* sp = d.scale_p + 3 * (size_t)s.pt;
            spk[0] = sp[0]; spk[1] = sp[1]; spk[2] = sp[2];
            double F[18], E[6];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const double* j = pr.jp + 3 * row;
                F[9 * row + 0] = -2.0 * (j[1] * pr.rp[2] - j[2] * pr.rp[1]) * mq * sc[0];
                F[9 * row + 1] = -2.0 * (j[2] * pr.rp[0] - j[0] * pr.rp[2]) * mq * sc[1];
                F[9 * row + 2] = -2.0 * (j[0] * pr.rp[1] - j[1] * pr.rp[0]) * mq * sc[2];
                F[9 * row + 3] = j[0] * mt * sc[3];
                F[9 * row + 4] = j[1] * mt * sc[4];
                F[9 * row + 5] = j[2] * mt * sc[5];
                F[9 * row + 6] = pr.ji[3 * row + 0] * mi * sc[6];
                F[9 * row + 7] = pr.ji[3 * row + 1] * mi * sc[7];
                F[9 * row + 8] = pr.ji[3 * row + 2] * mi * sc[8];
                E[3 * row + 0] = (j[0] * M[0] + j[1] * M[3] + j[2] * M[6]) * mp * spk[0];
                E[3 * row + 1] = (j[0] * M[1] + j[1] * M[4] + j[2] * M[7]) * mp * spk[1];
                E[3 * row + 2] = (j[0] * M[2] + j[1] * M[5] + j[2] * M[8]) * mp * spk[2];
            }
            d.rt[s.slot] = r0; d.rt[ns + s.slot] = r1;
#pragma unroll
            for (int k = 0; k < 18; ++k) w.Fw[k * ns + s.slot] = F[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) w.Ew[k * ns + s.slot] = E[k];
            v[0] = E[0] * E[0] + E[3] * E[3]; v[1] = E[0] * E[1] + E[3] * E[4]; v[2] = E[0] * E[2] + E[3] * E[5];
            v[3] = E[1] * E[1] + E[4] * E[4]; v[4] = E[1] * E[2] + E[4] * E[5]; v[5] = E[2] * E[2] + E[5] * E[5];
            v[6] = E[0] * r0 + E[3] * r1; v[7] = E[1] * r0 + E[4] * r1; v[8] = E[2] * r0 + E[5] * r1;
            // camera side: diag(F^T F) and F^T r of this observation -> the camera-major scatter buffer (below)
#pragma unroll
            for (int k = 0; k < 9; ++k) { cs0[k] = F[k] * F[k] + F[9 + k] * F[9 + k]; cs1[k] = F[k] * r0 + F[9 + k] * r1; }
            if (s.head && (!is_long || tl == 0) && !pt_fixed) xn2 += Pw[0] * Pw[0] + Pw[1] * Pw[1] + Pw[2] * Pw[2];
            if (is_long && lane == 0 && tl == 0) long_pt = s.pt;
        }
        {
            const int Cg = is_long ? 0 : d.tile_ncam[it.first_tile + tl];
            const int cpg = d.slot_campos_g[s.slot];
            if (Cg > 0) {
                int mypos, run_pk[2];
                tile_camera_runs<9>(d, it.first_tile + tl, s.slot, lane, Cg, mypos, run_pk);
                tile_camera_sums<9>(red, cs0, lane, Cg, mypos, run_pk, cpg, w.scat, 18, 0);
                tile_camera_sums<9>(red, cs1, lane, Cg, mypos, run_pk, cpg, w.scat, 18, 9);
            } else if (s.valid) {
                double* out = w.scat + 18 * (size_t)cpg;
#pragma unroll
                for (int k = 0; k < 9; ++k) { out[k] = cs0[k]; out[9 + k] = cs1[k]; }
            }
        }
        seg_reduce<9>(v, s.pt, lane, is_long ? kWave : maxlen);
        if (!is_long) {
            if (s.head) {
                double* H = d.Hpp + 6 * (size_t)s.pt;
#pragma unroll
                for (int k = 0; k < 6; ++k) H[k] = v[k];
                double* g = d.gp + 3 * (size_t)s.pt;
                g[0] = v[6]; g[1] = v[7]; g[2] = v[8];
                gm = fmax(gm, fmax(fabs(v[6] / spk[0]), fmax(fabs(v[7] / spk[1]), fabs(v[8] / spk[2]))));
            }
        } else if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += v[k];
        }
    }
    if (is_long && lane == 0 && long_pt >= 0) {
        double* H = d.Hpp + 6 * (size_t)long_pt;
#pragma unroll
        for (int k = 0; k < 6; ++k) H[k] = acc[k];
        double* g = d.gp + 3 * (size_t)long_pt;
        g[0] = acc[6]; g[1] = acc[7]; g[2] = acc[8];
        const double* sp = d.scale_p + 3 * (size_t)long_pt;
        gm = fmax(fabs(acc[6] / sp[0]), fmax(fabs(acc[7] / sp[1]), fabs(acc[8] / sp[2])));
    }
    cost = wave_sum(cost);
    xn2 = wave_sum(xn2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_xor(gm, off, kWave));
    if (lane == 0) { d.part[item] = cost; d.part[d.n_items + item] = xn2; d.part[2 * (size_t)d.n_items + item] = gm; }
}

// 168 VGPRs: 3 waves per SIMD (the 24 values of F and E are live across the stores and the products)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(XBA_K9LIN_W, XBA_K9LIN_W))) void k9_linearize(Dev d, DevW w, double huber_a) {
    __shared__ double red_all[kWavesPerBlock][kWave * kRedLd];
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    double* red = red_all[threadIdx.x >> 6];
    if (it.n_tiles > 1) linearize9_item<true>(d, w, it, item, lane, huber_a, red);
    else linearize9_item<false>(d, w, it, item, lane, huber_a, red);
}

__global__ void k9_scale_from_norms(Dev d, DevW w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_cams * kW) w.scale_c[i] = 1.0 / (1.0 + sqrt(w.camlin[18 * (size_t)(i / kW) + i % kW]));
    if (i < d.n_pts * 3) {
        const int p = i / 3, k = i % 3;
        const int di = (k == 0) ? 0 : (k == 1 ? 3 : 5);
        d.scale_p[i] = 1.0 / (1.0 + sqrt(d.Hpp[6 * (size_t)p + di]));
    }
}

// Ceres' gradient max-norm |x - Plus(x, -g)|_inf over the camera blocks (intrinsics: plain |g| of the unscaled gradient)
__global__ __launch_bounds__(kPcgThreads) void k9_gradmax_cams(Dev d, DevW w, double* __restrict__ out) {
    __shared__ double lds[kPcgThreads / kWave];
    double m = 0.0;
    for (int c = threadIdx.x; c < d.n_cams; c += kPcgThreads) {
        const double* g = w.camlin + 18 * (size_t)c + 9;
        const double* sc = w.scale_c + 9 * (size_t)c;
        const unsigned cc = d.cam_const[c];
        const bool active = d.cam_act[c] > 0.0;
        if (active && !(cc & 1u)) {
            const CamRec& cur = d.cam[c];
            const double q[4] = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
            const double dl[3] = {-g[0] / sc[0], -g[1] / sc[1], -g[2] / sc[2]};
            double qn[4];
            quat_plus(q, dl, qn);
            for (int k = 0; k < 4; ++k) m = fmax(m, fabs(q[k] - qn[k]));
        }
        if (active && !(cc & 2u)) for (int k = 3; k < 6; ++k) m = fmax(m, fabs(g[k] / sc[k]));
        if (active && (cc & 4u)) for (int k = 6; k < 9; ++k) m = fmax(m, fabs(g[k] / sc[k]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, kWave));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < kPcgThreads / kWave; ++i) r = fmax(r, lds[i]);
        *out = r;
    }
}

// ---------------------------------------------------------------- S assembly: diagonal-block / rhs terms per observation, 9x9 blocks per pair
__device__ __forceinline__ void pairs9_V(const double* F, const double* E, const double* cf, double* V) {      // V = (F^T E) C, C upper (point_factor)
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        const double w0 = F[a] * E[0] + F[9 + a] * E[3], w1 = F[a] * E[1] + F[9 + a] * E[4], w2 = F[a] * E[2] + F[9 + a] * E[5];
        V[3 * a + 0] = w0 * cf[0];
        V[3 * a + 1] = w0 * cf[1] + w1 * cf[3];
        V[3 * a + 2] = w0 * cf[2] + w1 * cf[4] + w2 * cf[5];
    }
}
__device__ __forceinline__ void pairs9_diag(const double* F, const double* V, const double* cf, const double* g, double* __restrict__ out) {
    int idx = 0;
    for (int a = 0; a < 9; ++a)
        for (int c2 = a; c2 < 9; ++c2)
            out[idx++] = F[a] * F[c2] + F[9 + a] * F[9 + c2] - (V[3 * a] * V[3 * c2] + V[3 * a + 1] * V[3 * c2 + 1] + V[3 * a + 2] * V[3 * c2 + 2]);
    const double u0 = cf[0] * g[0], u1 = cf[1] * g[0] + cf[3] * g[1], u2 = cf[2] * g[0] + cf[4] * g[1] + cf[5] * g[2];      // C^T g
    for (int a = 0; a < 9; ++a) out[45 + a] = -(V[3 * a] * u0 + V[3 * a + 1] * u1 + V[3 * a + 2] * u2);
    out[54] = 0.0; out[55] = 0.0;
}

// (round 4: the items that are NOT Gram tiles — tiles with more than kGramMaxCamsWide cameras, long tracks; item_list as for
//  k_schur_pairs<false, ...>.  Their per-observation diagonal terms go to the S assembly's camera-major entries, slot_campos_g.)
__global__ __launch_bounds__(kBlock) void k9_pairs(Dev d, DevW w, const int* __restrict__ item_list, int n_list, const int* __restrict__ slot_pair_ptr,
                                                  const int* __restrict__ pair_dst, double* __restrict__ scat2, double radius) {
    const int lane = threadIdx.x & (kWave - 1);
    const int li_ = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (li_ >= n_list) return;
    const Item it = d.items[item_list[li_]];
    if (it.n_tiles == 1) {
        const SlotCtx s = load_slot(d, it.first_tile, lane);
        double V[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) V[k] = 0.0;
        int npair = 0, pbase = 0;
        if (s.valid) {
            double F[18], E[6], cf[6];
            load_FE9(d, w, s.slot, F, E);
            const double* hp = d.Hpp + 6 * (size_t)s.pt;
            const double h[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
            point_factor(h, radius, cf);
            pairs9_V(F, E, cf, V);
            const double* gp = d.gp + 3 * (size_t)s.pt;
            const double g[3] = {gp[0], gp[1], gp[2]};
            pairs9_diag(F, V, cf, g, w.scat + kWS * (size_t)d.slot_campos_g[s.slot]);
            pbase = slot_pair_ptr[s.slot];
            npair = slot_pair_ptr[s.slot + 1] - pbase;
        }
        int maxp = npair;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxp = max(maxp, __shfl_xor(maxp, off, kWave));
        for (int dd = 1; dd <= maxp; ++dd) {
            double Vb[27];
#pragma unroll
            for (int k = 0; k < 27; ++k) Vb[k] = __shfl_down(V[k], dd, kWave);
            if (dd <= npair) {
                double* out = scat2 + kWB * (size_t)pair_dst[pbase + dd - 1];
                for (int rb = 0; rb < 9; ++rb)
                    for (int ca = 0; ca < 9; ++ca)
                        out[9 * rb + ca] = Vb[3 * rb] * V[3 * ca] + Vb[3 * rb + 1] * V[3 * ca + 1] + Vb[3 * rb + 2] * V[3 * ca + 2];
            }
        }
        return;
    }
    // long track: lane handles observation a, loops over all later observations b of the track
    const int s_begin = it.first_tile * kWave, s_end = s_begin + it.n_tiles * kWave;
    for (int sa = s_begin + lane; sa < s_end; sa += kWave) {
        if (d.slot_cam[sa] < 0) continue;
        const int pt = d.slot_pt[sa];
        const double* hp = d.Hpp + 6 * (size_t)pt;
        const double h[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
        double cf[6];
        point_factor(h, radius, cf);
        double Fa[18], Ea[6], Va[27];
        load_FE9(d, w, sa, Fa, Ea);
        pairs9_V(Fa, Ea, cf, Va);
        const double g[3] = {d.gp[3 * (size_t)pt], d.gp[3 * (size_t)pt + 1], d.gp[3 * (size_t)pt + 2]};
        pairs9_diag(Fa, Va, cf, g, w.scat + kWS * (size_t)d.slot_campos_g[sa]);
        const int pbase = slot_pair_ptr[sa];
        const int npair = slot_pair_ptr[sa + 1] - pbase;
        for (int dd = 1; dd <= npair; ++dd) {
            const int sb = sa + dd;
            double Fb[18], Eb[6], Vb[27];
            load_FE9(d, w, sb, Fb, Eb);
            pairs9_V(Fb, Eb, cf, Vb);
            double* out = scat2 + kWB * (size_t)pair_dst[pbase + dd - 1];
            for (int rb = 0; rb < 9; ++rb)
                for (int ca = 0; ca < 9; ++ca)
                    out[9 * rb + ca] = Vb[3 * rb] * Va[3 * ca] + Vb[3 * rb + 1] * Va[3 * ca + 1] + Vb[3 * rb + 2] * Va[3 * ca + 2];
        }
    }
}

// Value IDX of the 56-record of pairs9_diag (45 upper-triangle entries of F^T F - V V^T, 9 of -V C^T g, 2 pad) with compile-time
// indices, and a run of 14 of them stored to dst[0..13]: the Gram kernel forms the record in four rounds of 14 live values
// (the whole record in registers cost 112 VGPRs: one wave per SIMD at NI = 4).
constexpr int tri9_row(int idx) { int a = 0; while (idx >= 9 - a) { idx -= 9 - a; ++a; } return a; }
constexpr int tri9_col(int idx) { int a = 0; while (idx >= 9 - a) { idx -= 9 - a; ++a; } return a + idx; }
template <int IDX>
__device__ __forceinline__ double diag9_value(const double (&F)[18], const double (&V)[27], double u0, double u1, double u2) {
    if constexpr (IDX < 45) {
        constexpr int a = tri9_row(IDX), c2 = tri9_col(IDX);
        return F[a] * F[c2] + F[9 + a] * F[9 + c2] - (V[3 * a] * V[3 * c2] + V[3 * a + 1] * V[3 * c2 + 1] + V[3 * a + 2] * V[3 * c2 + 2]);
    } else if constexpr (IDX < 54) {
        constexpr int a = IDX - 45;
        return -(V[3 * a] * u0 + V[3 * a + 1] * u1 + V[3 * a + 2] * u2);
    } else return 0.0;
}
template <int IDX0, int K>
__device__ __forceinline__ void diag9_store(double* dst, const double (&F)[18], const double (&V)[27], double u0, double u1, double u2) {
    if constexpr (K < 14) {
        dst[K] = diag9_value<IDX0 + K>(F, V, u0, u1, u2);
        diag9_store<IDX0, K + 1>(dst, F, V, u0, u1, u2);
    }
}

// Gram tiles in bal9 mode (round 4).  The per-pair kernel above writes one 648-byte block per
// observation pair and 448 bytes of diagonal terms per observation, which the segmented sums read back: 2.8 GB per LM iteration
// at config 4's size, 0.046 of the HBM roofline.  Here a tile of T tracks over C <= 7 distinct cameras stages V = W chol(Hinv)
// as a [9C x 3T] operand in LDS and forms G = V V^T on the FP64 matrix cores (gram_tile<NI, 9> of ba_chol.h: every camera-pair
// block of the tile already summed over its tracks, written once), and the 56 diagonal-block / rhs values are summed per
// distinct camera of the tile through LDS before they are written (four rounds of 14 values): one wave = one tile, one
// instantiation per operand height NI = ceil(9 C / 16).
template <int NI>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(NI < 4 ? XBA_K9G_W : 2, NI < 4 ? XBA_K9G_W : 2))) void k9_pairs_gram(Dev d, DevW w, const int* __restrict__ tile_list, const int* __restrict__ pair_dst,
                                                       int n_obs_pairs, double* __restrict__ scat2, double radius) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const int tile = tile_list[blockIdx.x];
    const int C = d.tile_ncam[tile];
    int dt0 = -1, dt1 = -1;                  // entries lane and lane + 64 of the kGramTabLd x kGramTabLd destination table
    {
        const int* src = pair_dst + n_obs_pairs + d.tile_gt_off[tile];
        const int a0 = lane / kGramTabLd, b0 = lane - kGramTabLd * a0;
        const int a1 = (lane + kWave) / kGramTabLd, b1 = lane + kWave - kGramTabLd * a1;
        if (b0 > a0 && b0 < C) dt0 = src[a0 * C + b0];
        if (b1 > a1 && b1 < C && a1 < kGramTabLd) dt1 = src[a1 * C + b1];
    }
    const SlotCtx s = load_slot(d, tile, lane);
    const int cp = d.slot_campos_g[s.slot];
    const int cidx_raw = (int)d.slot_cidx[s.slot];
    double V[27], F[18], u0, u1, u2;         // (lanes without an observation: never staged, their sums parked in a row no camera reads)
    if (s.valid) {
        double E[6], cf[6];
        load_FE9(d, w, s.slot, F, E);
        const double* hp = d.Hpp + 6 * (size_t)s.pt;
        const double h[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
        const double* gp = d.gp + 3 * (size_t)s.pt;
        const double g[3] = {gp[0], gp[1], gp[2]};
        point_factor(h, radius, cf);
        pairs9_V(F, E, cf, V);
        u0 = cf[0] * g[0]; u1 = cf[1] * g[0] + cf[3] * g[1]; u2 = cf[2] * g[0] + cf[4] * g[1] + cf[5] * g[2];      // C^T g
    } else { dead_values(V); dead_values(F); XBA_DEAD_VALUE(u0); XBA_DEAD_VALUE(u1); XBA_DEAD_VALUE(u2); }

    const unsigned long long headmask = __ballot(s.head);
    const int T = __popcll(headmask);
    const int nvalid = __popcll(__ballot(s.valid));
    const bool dense = nvalid == T * C;
    {   // the 56 diagonal-block / rhs values, summed per distinct camera of the tile (tile_camera_sums), four rounds of 14
        int mypos, run_pk[2];
        tile_camera_runs<14>(d, tile, s.slot, lane, C, mypos, run_pk);
        double* red = smem;                                         // [64][kRedLd]
        auto round = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            double v14[14];
            diag9_store<14 * h, 0>(v14, F, V, u0, u1, u2);          // formed where they are needed: 14 live values, not 56
            tile_camera_sums<14>(red, v14, lane, C, mypos, run_pk, cp, w.scat, kWS, 14 * h);
        };
        round(std::integral_constant<int, 0>{}); round(std::integral_constant<int, 1>{});
        round(std::integral_constant<int, 2>{}); round(std::integral_constant<int, 3>{});
    }
    const int t = __popcll(headmask & ((2ull << lane) - 1ull)) - 1;        // rank of the lane's track in the tile
    const int cidx = s.valid ? cidx_raw : 0;
    int passes = 1;
    (void)gram_lds_need(C, T, &passes, kW);
    const int Th = (T + passes - 1) / passes;
    const int R = kW * C, Cp = ((3 * Th + 3) & ~3) + kGramPad;
    double* Vst = smem;
    int* dtab = reinterpret_cast<int*>(smem + R * Cp);
    dtab[lane] = dt0;
    if (lane + kWave < kGramTabLd * kGramTabLd) dtab[lane + kWave] = dt1;
    gram_tile<NI, kW>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, Th, passes, dense);
}

// workgroups [0, n_cams): the 56 diagonal-block / rhs values of a camera; [n_cams, n_cams + n_blocks): the 81 values of a block
__global__ __launch_bounds__(kBlock) void k9_chol_segsum(const double* __restrict__ scat, const int* __restrict__ cam_ptr, double* __restrict__ camS,
                                                        int n_cams, const double* __restrict__ scat2, const int* __restrict__ blk_ptr,
                                                        double* __restrict__ Sblk) {
    if ((int)blockIdx.x < n_cams) segsum_body<kWS>(scat, cam_ptr, camS, blockIdx.x);
    else segsum_body<kWB>(scat2, blk_ptr, Sblk, blockIdx.x - n_cams);
}

// tile fill with 9x9 blocks (compose_tile of ba_chol.h restated for the wide records)
__global__ __launch_bounds__(256) void k9_tile_fill(CholDev c, Dev d, DevW w, const int* __restrict__ tiles, const int* __restrict__ tptr,
                                                    const int* __restrict__ tent, const double* __restrict__ Sblk,
                                                    const int* __restrict__ blk_rc, double radius) {
    __shared__ double A[kNB * (kNB + 1)];
    __shared__ double rl[kNB];
    constexpr int LD = kNB + 1;
    const int q = blockIdx.x;
    const int ti = tiles[2 * q], tj = tiles[2 * q + 1];
    const int t = threadIdx.x;
    const int nrows = c.tile_rows[ti];
    for (int e = t; e < kNB * kNB; e += 256) {
        const int r = e >> 6, col = e & 63;
        A[r * LD + col] = (ti == tj && r == col && r >= nrows) ? 1.0 : 0.0;
    }
    if (t < kNB) rl[t] = 0.0;
    __syncthreads();
    const int q0 = tptr[q], q1 = tptr[q + 1];
    for (int x = t; x < (q1 - q0) * kWB; x += 256) {
        const int ent = tent[q0 + x / kWB], e = x % kWB, r = e / kW, col = e % kW;
        if (ent >= 0) {
            const int orow = c.cam_off[blk_rc[2 * ent]], ocol = c.cam_off[blk_rc[2 * ent + 1]];
            const double v = -Sblk[kWB * (size_t)ent + e];
            if (orow > ocol) A[((orow & 63) + r) * LD + (ocol & 63) + col] = v;
            else A[((ocol & 63) + col) * LD + (orow & 63) + r] = v;
        } else {
            const int cam = -ent - 1, o = c.cam_off[cam] & 63;
            const double* S = w.camS + kWS * (size_t)cam;
            if (col >= r) {
                double v = S[kW * r - r * (r - 1) / 2 + (col - r)];      // packed upper triangle (r, col)
                if (r == col) v += clampd(w.camlin[18 * (size_t)cam + r], kLmDiagMin, kLmDiagMax) / radius;
                A[(o + col) * LD + o + r] = v;
            }
            if (e < kW) rl[o + e] = w.camlin[18 * (size_t)cam + 9 + e] + S[45 + e];
        }
    }
    __syncthreads();
    double* base = tile_ptr(c, ti, tj);
    for (int e = t; e < kNB * kNB; e += 256) {
        const int r = e >> 6, col = e & 63;
        base[(size_t)r * c.ld + col] = A[r * LD + col];
    }
    if (ti == tj && t < kNB) c.rhs[ti * kNB + t] = rl[t];
}

// ---------------------------------------------------------------- back-substitution, candidate state
__global__ __launch_bounds__(kBlock) void k9_backsub(Dev d, DevW w, int n_item_blocks, double radius) {
    if ((int)blockIdx.x >= n_item_blocks) {       // candidate cameras: Plus(x, -y * scale) on {q, t, (f, k1, k2)}
        const int c = (blockIdx.x - n_item_blocks) * kBlock + threadIdx.x;
        if (c >= d.n_cams) return;
        const CamRec& cur = d.cam[c];
        const double* y = w.px + 9 * (size_t)c;
        const double* sc = w.scale_c + 9 * (size_t)c;
        const double q[4] = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
        const double t[3] = {cur.t[0], cur.t[1], cur.t[2]};
        double qn[4] = {q[0], q[1], q[2], q[3]}, tn[3] = {t[0], t[1], t[2]}, in[3] = {cur.intr[0], cur.intr[1], cur.intr[2]};
        const unsigned cc = d.cam_const[c];
        const bool active = d.cam_act[c] > 0.0;
        double step2 = 0.0, xn2 = 0.0;
        if (active && !(cc & 1u)) {
            const double dl[3] = {-y[0] * sc[0], -y[1] * sc[1], -y[2] * sc[2]};
            quat_plus(q, dl, qn);
            for (int k = 0; k < 4; ++k) { const double df = qn[k] - q[k]; step2 += df * df; xn2 += q[k] * q[k]; }
        }
        if (active && !(cc & 2u))
            for (int k = 0; k < 3; ++k) { tn[k] = t[k] + (-y[3 + k] * sc[3 + k]); const double df = tn[k] - t[k]; step2 += df * df; xn2 += t[k] * t[k]; }
        if (active && (cc & 4u))
            for (int k = 0; k < 3; ++k) { const double v0 = cur.intr[k]; in[k] = v0 + (-y[6 + k] * sc[6 + k]); const double df = in[k] - v0; step2 += df * df; xn2 += v0 * v0; }
        CamRec& out = d.cam_cand[c];
        for (int k = 0; k < 4; ++k) out.q[k] = qn[k];
        for (int k = 0; k < 3; ++k) out.t[k] = tn[k];
        out.pad = cur.pad;
        for (int k = 0; k < 8; ++k) out.intr[k] = k < 3 ? in[k] : cur.intr[k];
        d.campart[c] = step2;
        d.campart[d.n_cams + c] = xn2;
        return;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    const size_t ns = (size_t)d.n_slots;
    double model = 0.0, step2 = 0.0;
    // two passes over the item's tiles (one tile: both in registers): w = sum E^T F y per track, then the model residual
    double wsum[3] = {0, 0, 0};
    const bool is_long = it.n_tiles > 1;
    double u[3] = {0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        for (int tl = 0; tl < it.n_tiles; ++tl) {
            const SlotCtx s = load_slot(d, it.first_tile + tl, lane);
            double E[6] = {0, 0, 0, 0, 0, 0}, v0 = 0.0, v1 = 0.0, r0 = 0.0, r1 = 0.0;
            if (s.valid) {
                double F[18];
                load_FE9(d, w, s.slot, F, E);
                const double* y = w.px + 9 * (size_t)s.cam;
#pragma unroll
                for (int k = 0; k < 9; ++k) { v0 += F[k] * y[k]; v1 += F[9 + k] * y[k]; }
                r0 = d.rt[s.slot]; r1 = d.rt[ns + s.slot];
            }
            if (pass == 0) {
                double wv[3] = {E[0] * v0 + E[3] * v1, E[1] * v0 + E[4] * v1, E[2] * v0 + E[5] * v1};
                if (!is_long) {
                    seg_reduce<3>(wv, s.pt, lane, d.tile_maxlen[it.first_tile]);
                    if (s.head) {
                        const double* hp = d.Hpp + 6 * (size_t)s.pt;
                        const double h[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
                        double cf[6];
                        point_factor(h, radius, cf);
                        const double* g = d.gp + 3 * (size_t)s.pt;
                        const double a0 = g[0] - wv[0], a1 = g[1] - wv[1], a2 = g[2] - wv[2];
                        const double s0 = cf[0] * a0, s1 = cf[1] * a0 + cf[3] * a1, s2 = cf[2] * a0 + cf[4] * a1 + cf[5] * a2;
                        u[0] = cf[0] * s0 + cf[1] * s1 + cf[2] * s2; u[1] = cf[3] * s1 + cf[4] * s2; u[2] = cf[5] * s2;
                        const double* sp = d.scale_p + 3 * (size_t)s.pt;
                        const double* P = d.P + 3 * (size_t)s.pt;
                        double* Pc = d.P_cand + 3 * (size_t)s.pt;
                        double* yo = d.yp + 3 * (size_t)s.pt;
                        const bool var = !d.pt_const[s.pt];
                        for (int k = 0; k < 3; ++k) {
                            const double pn = P[k] + (var ? -u[k] * sp[k] : 0.0);
                            Pc[k] = pn; yo[k] = u[k];
                            const double df = pn - P[k];
                            step2 += df * df;
                        }
                    }
                    const int hl = seg_head_lane(s.head || !s.valid, lane);
                    u[0] = __shfl(u[0], hl, kWave); u[1] = __shfl(u[1], hl, kWave); u[2] = __shfl(u[2], hl, kWave);
                    if (s.valid) {
                        const double m0 = v0 + E[0] * u[0] + E[1] * u[1] + E[2] * u[2];
                        const double m1 = v1 + E[3] * u[0] + E[4] * u[1] + E[5] * u[2];
                        model += m0 * (r0 - 0.5 * m0) + m1 * (r1 - 0.5 * m1);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k) wsum[k] += wv[k];
                }
            } else if (is_long && s.valid) {
                const double m0 = v0 + E[0] * u[0] + E[1] * u[1] + E[2] * u[2];
                const double m1 = v1 + E[3] * u[0] + E[4] * u[1] + E[5] * u[2];
                model += m0 * (r0 - 0.5 * m0) + m1 * (r1 - 0.5 * m1);
            }
        }
        if (!is_long) break;
        if (pass == 0) {       // the long track's point step (every lane computes it: same values)
#pragma unroll
            for (int k = 0; k < 3; ++k) { wsum[k] = wave_sum(wsum[k]); wsum[k] = __shfl(wsum[k], 0, kWave); }
            int pt0 = d.slot_pt[it.first_tile * kWave];
            const double* hp = d.Hpp + 6 * (size_t)pt0;
            const double h[6] = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5]};
            double cf[6];
            point_factor(h, radius, cf);
            const double* g = d.gp + 3 * (size_t)pt0;
            const double a0 = g[0] - wsum[0], a1 = g[1] - wsum[1], a2 = g[2] - wsum[2];
            const double s0 = cf[0] * a0, s1 = cf[1] * a0 + cf[3] * a1, s2 = cf[2] * a0 + cf[4] * a1 + cf[5] * a2;
            u[0] = cf[0] * s0 + cf[1] * s1 + cf[2] * s2; u[1] = cf[3] * s1 + cf[4] * s2; u[2] = cf[5] * s2;
            if (lane == 0) {
                const double* sp = d.scale_p + 3 * (size_t)pt0;
                const double* P = d.P + 3 * (size_t)pt0;
                double* Pc = d.P_cand + 3 * (size_t)pt0;
                double* yo = d.yp + 3 * (size_t)pt0;
                const bool var = !d.pt_const[pt0];
                for (int k = 0; k < 3; ++k) {
                    const double pn = P[k] + (var ? -u[k] * sp[k] : 0.0);
                    Pc[k] = pn; yo[k] = u[k];
                    const double df = pn - P[k];
                    step2 += df * df;
                }
            }
        }
    }
    model = wave_sum(model);
    step2 = wave_sum(step2);
    if (lane == 0) { d.part[2 * d.n_items + item] = model; d.part[3 * d.n_items + item] = step2; }
}

}  // namespace xba
