// HIP kernels (gfx950 / CDNA4, wave64) of the Schur-complement LM iteration.
//
// Data layout: observations are sorted by track and packed
// into 64-slot *tiles*, one wavefront per tile, whole tracks per tile, so that
// every per-track sum (3x3 point blocks, E^T v) is a segmented wave reduction
// with shuffles and every per-observation array is a lane-contiguous SoA stream
// (512 B per wave-load).  Tracks longer than 64 observations own a run of whole
// tiles handled by one wave ("long item").  Camera-side sums go through a
// camera-major scatter buffer + a fixed-order segmented sum, so results are
// bit-reproducible for a given tiling (no floating-point atomics anywhere).
#pragma once
#include "ba_math.h"

namespace xba {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // 4 waves per workgroup, waves are independent
constexpr int kWavesPerBlock = kBlock / kWave;

struct Item { int first_tile; int n_tiles; };

// status words of the PCG loop, device resident
struct PcgStatus {
    double rz;      // r^T M^-1 r
    double rr;      // |r|^2
    double bb;      // |b|^2
    double pq;
    int it;
    int done;
    unsigned arrived;   // workgroups of k_pcg_p that have read the status (the last one writes the new one)
};

// scalar slots (device buffer `scal`)
enum Scal {
    S_COST = 0, S_XNORM2_PTS, S_MODEL, S_STEP2_PTS, S_COST_CAND, S_STEP2_CAMS, S_XNORM2_CAMS,     // [0,4) and [2,5) each travel in one all-reduce
    S_GRADMAX_PTS, S_GRADMAX_CAMS,
    S_BWD_ERR,          // != 0: a hand-off of the one-launch backward substitution timed out (k_lv_bwd_all) — travels to the host with every scalar hand-over
    S_COUNT = 16
};

// Per-camera data the consumers need to rebuild the Jacobian blocks of an observation (128 bytes, one line).
struct __attribute__((aligned(16))) CamLin {
    double M[9];     // M(q) at the linearisation point
    double sq[3];    // Jacobi scale of the rotation columns, 0 if q is constant
    double st[3];    // ... of the translation columns, 0 if t is constant
    double pad;
};
static_assert(sizeof(CamLin) == 128, "CamLin must be 128 bytes");

struct Dev {
    int n_cams, n_pts, n_tiles, n_slots, n_items;
    // per slot
    const int* slot_cam; const int* slot_pt; const int* slot_campos;
    const double* slot_u; const double* slot_v;
    const Item* items;
    const int* tile_stride;   // [n_tiles] L > 0 for regular tiles (all tracks share one tuple of L cameras)
    const int* tile_maxlen;   // [n_tiles] longest track of the tile (bounds the segmented reductions)
    const int* tile_ncam;     // [n_tiles] C > 0: Gram tile with C distinct cameras (ba_pack.h)
    const int* tile_gt_off;   // [n_tiles] offset of its C x C destination table behind the observation pairs in pair_dst
    const unsigned char* slot_cidx;   // [n_slots] index of the slot's camera among the tile's distinct cameras
    const int* slot_campos_g; // [n_slots] like slot_campos, for the S assembly (one writer per distinct camera of a Gram tile)
    // (round 6) per-camera sums of a RAGGED Gram tile go through LDS with the lanes' values deposited sorted by camera; where a lane
    // deposits and which run a (camera, value) lane adds are properties of the tiling — computed once per context (k_gram_runs)
    // instead of by a ballot loop over the tile's cameras in every launch of k_linearize and k_schur_pairs:
    const unsigned char* slot_gpos;   // [n_slots] position of the slot's lane in the camera-sorted order of its tile (63 = no observation)
    const int* tile_run;      // [n_tiles][kTileRunLd] per distinct camera: run start | length << 8 | first lane << 16
    const int* cam_ptr_g;     // [n_cams+1]
    // cameras
    CamRec* cam; CamRec* cam_cand; const int* cam_model; const unsigned char* cam_const; const int* cam_ptr;
    double* cam_act;    // [Nc] 1.0 if any rank observes the camera (cameras without observations are not in the program)
    // points (AoS xyz)
    double* P; double* P_cand; const unsigned char* pt_const;
    // Jacobi scaling
    double* scale_c; double* scale_p;
    // linearisation (SoA over slots: component-major)
    double* rt; double* Jp;   // rt[2][n]: robustified residual; Jp[6][n]: sqrt(rho') * d r / d Pc (2x3) — the camera (2x6)
                              // and point (2x3) blocks are rebuilt from it by load_FE() ("compressed J": 64 instead of 160 B/obs)
    CamLin* camrec;           // [Nc] per-camera linearisation record (rotation matrix, scale*mask of the 6 columns)
    double* Hpp; double* gp; double* Hinv;
    double* Hc;          // [n_pts][6] lower Cholesky factor of Hinv (c00 c10 c20 c11 c21 c22): S assembly uses V = W Hc
    double* camlin;     // [Nc][12]: diag(Hcc) (6), gc (6)
    double* Dc2;        // [Nc][6]
    double* camS;       // [Nc][28]: Scc upper (21), rb (6), pad
    double* Minv;       // [Nc][21]
    double* b;          // [Nc][6]
    // PCG vectors [Nc][6]
    double* px; double* pr; double* pz; double* pp; double* pq;
    double* yp;         // [Np][3] point step (scaled coords)
    double* scat;       // camera-major scatter buffer, [n_obs][28]
    double* part;       // per-item partial sums, 4 * n_items
    double* campart;    // per-camera partials, 2 * n_cams
    double* ptpart;     // per-workgroup partials of point kernels, cdiv(n_pts, 256)
    double* pcgpart;    // [3 + kGauge][n_cams] per-camera partial dot products of the PCG iteration (p.q | r.z | r.r | W_k.r)
    double* pcgW;       // [kGauge][n_cams][6] gauge coarse space of the PCG preconditioner (k_pcg_gauge), nullptr = block-Jacobi alone
    double* pcgSW;      // [kGauge][n_cams][6] (S + D^2) W (set-up of the coarse matrix)
    double* pcgE;       // [kGauge * kGauge] inverse of the coarse matrix W^T (S + D^2) W (k_pcg_coarse)
    double* scal;       // S_COUNT scalars
    PcgStatus* st;
};

// Developer aid (-DXBA_TIMELINE, tools/timeline.py): cycle stamps of sampled waves inside the streaming kernels.
#ifdef XBA_TIMELINE
__device__ unsigned long long g_stamps[3][64][16];
#define XBA_STAMP(kern, i) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 64 && (threadIdx.x >> 6) == 0) g_stamps[kern][blockIdx.x / 97][i] = __builtin_readcyclecounter(); } while (0)
// ... the same from wave 1 of the workgroup (what the other waves of a factor workgroup do while wave 0 sweeps)
#define XBA_STAMP_W1(kern, i) do { if ((threadIdx.x & 63) == 0 && (blockIdx.x % 97) == 0 && blockIdx.x / 97 < 64 && (threadIdx.x >> 6) == 1) g_stamps[kern][blockIdx.x / 97][i] = __builtin_readcyclecounter(); } while (0)
#else
#define XBA_STAMP(kern, i) do {} while (0)
#define XBA_STAMP_W1(kern, i) do {} while (0)
#endif

// Parity-hardening build (-DXBA_POISON, tests/test_gpu_hardening.py): every per-lane temporary of the streaming kernels that a lane
// without an observation (or a lane that is not the head of its track) is NOT supposed to read is initialised with NaN
// instead of 0.  The result of a solve must not change by one bit: a value that leaks from such a lane through a shuffle,
// a reduction or a store turns into NaN and fails the test.  (Lanes whose values ARE read under a 0/1 mask — the operands of
// seg_reduce / strided_reduce — must hold zeros and keep them.)
// In the shipped build such a temporary is "some value": XBA_DEAD_VALUE(x) defines x without an instruction (an initialiser is a
// v_mov per register pair on every lane — 46 of the ~790 vector instructions k_schur_pairs spends on a tile, 28 of k_backsub's ~285).
#ifdef XBA_POISON
#define XBA_DEAD (__builtin_nan(""))
#define XBA_DEAD1 (__builtin_nan(""))
#define XBA_DEAD_VALUE(x) ((x) = __builtin_nan(""))
#else
#define XBA_DEAD 0.0
#define XBA_DEAD1 1.0
#define XBA_DEAD_VALUE(x) asm volatile("" : "=v"(x))
#endif
template <int N> __device__ __forceinline__ void dead_values(double (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) XBA_DEAD_VALUE(v[k]);
}
// Register-allocation probe (-DXBA_BACKSUB_WAVES=5, tools/backsub_waves_probe.py): k_backsub built for 5 waves per SIMD.
#ifndef XBA_BACKSUB_WAVES
#define XBA_BACKSUB_WAVES 0
#endif

// ---------------------------------------------------------------- wave helpers
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// Segmented suffix-sum over contiguous segments (key = track id); the head lane
// of each segment ends up with the segment total.  Fixed tree order.
template <int N>
__device__ __forceinline__ void seg_reduce(double (&v)[N], int key, int lane, int maxlen = kWave) {
    // maxlen: wave-uniform upper bound of the segment length (longest track of the tile): offsets >= maxlen cannot
    // stay inside a segment, so log2(maxlen) steps suffice (2 instead of 6 for 4-observation tracks)
    for (int off = 1; off < maxlen; off <<= 1) {
        const int okey = __shfl_down(key, off, kWave);
        const double m = ((lane + off < kWave) && (okey == key)) ? 1.0 : 0.0;    // 0/1 mask inside the fma: no selects
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = fma(__shfl_down(v[k], off, kWave), m, v[k]);
    }
}

// Regular tile (every track has the same L cameras, lane = track*L + rank): sum over the tracks, the lanes
// < L end up with the per-camera totals.  Lanes without data must hold zeros.  Fixed order.
// A lane whose partner lane + off lies outside the wave must not add (__shfl_down hands it its own value): the partner
// is multiplied by a 0/1 mask inside the fma that does the addition (fma(o, 1, v) == v + o exactly), which costs nothing,
// where a select would cost two v_cndmask per value.  For power-of-two strides the steps that stay inside a row of 16
// lanes (off < 16) are DPP row shifts (v_mov_b32 row_shl, zero past the row end: no LDS traffic) — there the summation
// tree of a lane l < L never leaves its row, and what the other lanes accumulate is never used.
template <int SHIFT>
__device__ __forceinline__ double row_shl_d(double v) {     // lane i <- lane i + SHIFT of the same 16-lane row (0 past its end)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x100 + SHIFT, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x100 + SHIFT, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int N, int SHIFT>
__device__ __forceinline__ void row_step(double (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] += row_shl_d<SHIFT>(v[k]);
}
template <int N>
__device__ __forceinline__ void strided_reduce(double (&v)[N], int stride, int lane) {
    int off = stride;
    if ((stride & (stride - 1)) == 0) {          // wave-uniform
        if (off == 1) { row_step<N, 1>(v); off = 2; }
        if (off == 2) { row_step<N, 2>(v); off = 4; }
        if (off == 4) { row_step<N, 4>(v); off = 8; }
        if (off == 8) { row_step<N, 8>(v); off = 16; }
    }
    for (; off < kWave; off <<= 1) {
        const double m = (lane + off < kWave) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = fma(__shfl_down(v[k], off, kWave), m, v[k]);
    }
}

__device__ __forceinline__ int seg_head_lane(bool head, int lane) {
    const unsigned long long m = __ballot(head);
    const unsigned long long below = m & ((2ull << lane) - 1ull);
    return 63 - __clzll((long long)below);
}

constexpr int kTileRunLd = 10;            // = kGramMaxCams (ba_pack.h)
// one wave per tile, once per context
__global__ __launch_bounds__(256) void k_gram_runs(const int* __restrict__ slot_cam, const unsigned char* __restrict__ slot_cidx, const int* __restrict__ tile_ncam,
                                                   int n_tiles, unsigned char* __restrict__ slot_gpos, int* __restrict__ tile_run) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int s = 64 * t + lane;
    const int C = tile_ncam[t];
    int mypos = 63, mypk = 0;
    if (C > 0) {                                        // (wave-uniform)
        const int cidx = slot_cam[s] >= 0 ? (int)slot_cidx[s] : -1;
        const unsigned long long lt = (1ull << lane) - 1ull;
        int run = 0;
        for (int cc = 0; cc < C; ++cc) {
            const unsigned long long m = __ballot(cidx == cc);
            const int cnt = __popcll(m);
            const int pk = run | (cnt << 8) | ((__ffsll((long long)m) - 1) << 16);
            if (cidx == cc) mypos = run + __popcll(m & lt);
            if (lane == cc) mypk = pk;
            run += cnt;
        }
    }
    slot_gpos[s] = (unsigned char)mypos;
    if (lane < kTileRunLd) tile_run[(size_t)t * kTileRunLd + lane] = mypk;
}

struct SlotCtx {
    int slot, cam, pt, lane;
    bool valid, head;
};

__device__ __forceinline__ SlotCtx load_slot(const Dev& d, int tile, int lane) {
    SlotCtx s;
    s.lane = lane;
    s.slot = tile * kWave + lane;
    s.cam = d.slot_cam[s.slot];
    s.pt = d.slot_pt[s.slot];
    s.valid = s.cam >= 0;
    const int prev = __shfl_up(s.pt, 1, kWave);
    s.head = s.valid && (lane == 0 || prev != s.pt);
    return s;
}


// Rebuild the Jacobi-scaled, robustified blocks of one observation:
//   F (2x6) = [ -2 (j x M P) * sq | j * st ],   E (2x3) = (j M) * sp        (j = rows of Jp)
__device__ __forceinline__ void load_FE(const Dev& d, int slot, int cam, int pt, double (&F)[12], double (&E)[6]) {
    const size_t ns = (size_t)d.n_slots;
    double j[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) j[k] = d.Jp[k * ns + slot];
    const CamLin& c = d.camrec[cam];
    double M[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = c.M[k];
    const double* P = d.P + 3 * (size_t)pt;
    const double P0 = P[0], P1 = P[1], P2 = P[2];
    const double rp0 = M[0] * P0 + M[1] * P1 + M[2] * P2;
    const double rp1 = M[3] * P0 + M[4] * P1 + M[5] * P2;
    const double rp2 = M[6] * P0 + M[7] * P1 + M[8] * P2;
    const double* spp = d.scale_p + 3 * (size_t)pt;
    const double mp = d.pt_const[pt] ? 0.0 : 1.0;
    const double sp0 = spp[0] * mp, sp1 = spp[1] * mp, sp2 = spp[2] * mp;
#pragma unroll
    for (int row = 0; row < 2; ++row) {
        const double a = j[3 * row], b = j[3 * row + 1], cc = j[3 * row + 2];
        F[6 * row + 0] = -2.0 * (b * rp2 - cc * rp1) * c.sq[0];
        F[6 * row + 1] = -2.0 * (cc * rp0 - a * rp2) * c.sq[1];
        F[6 * row + 2] = -2.0 * (a * rp1 - b * rp0) * c.sq[2];
        F[6 * row + 3] = a * c.st[0];
        F[6 * row + 4] = b * c.st[1];
        F[6 * row + 5] = cc * c.st[2];
        E[3 * row + 0] = (a * M[0] + b * M[3] + cc * M[6]) * sp0;
        E[3 * row + 1] = (a * M[1] + b * M[4] + cc * M[7]) * sp1;
        E[3 * row + 2] = (a * M[2] + b * M[5] + cc * M[8]) * sp2;
    }
}

// Per-camera linearisation record (run before every linearisation: cameras, scales and masks as they are now).
__device__ __forceinline__ void cam_lin_one(const Dev& d, int c, const double q[4], CamLin* __restrict__ out) {
    CamLin r;
    quat_to_mat(q, r.M);
    const unsigned cc = d.cam_const[c];
    const double* sc = d.scale_c + 6 * (size_t)c;
    for (int k = 0; k < 3; ++k) { r.sq[k] = (cc & 1u) ? 0.0 : sc[k]; r.st[k] = (cc & 2u) ? 0.0 : sc[3 + k]; }
    r.pad = 0.0;
    out[c] = r;
}
__global__ void k_cam_lin(Dev d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n_cams) return;
    const double q[4] = {d.cam[c].q[0], d.cam[c].q[1], d.cam[c].q[2], d.cam[c].q[3]};
    cam_lin_one(d, c, q, d.camrec);
}

// ---------------------------------------------------------------- linearise
// Residuals, robustified + Jacobi-scaled Jacobian blocks, per-track H_pp / g_p
// (segmented wave reduction), per-observation camera-side terms into the
// camera-major scatter buffer, cost and |x_points|^2 partials.
// LONG = the item is one track of more than 64 observations spread over several tiles; the accumulators that carries across
// tiles exist only in that instantiation (they would cost the common single-tile path ~20 VGPRs).
template <bool LONG>
__device__ __forceinline__ void linearize_item(const Dev& d, const Item& it, int item, int lane, double huber_a) {
    constexpr bool is_long = LONG;
    // long item: the 9 sums that carry over its tiles live in the wave's slice of the dynamic LDS (lane 0 only; the slice is
    // otherwise used by Gram tiles, which a long item never is) — in registers they cost the loop a spilled value
    extern __shared__ __attribute__((aligned(16))) double lin_smem_acc[];
    double* acc = lin_smem_acc + (threadIdx.x >> 6) * (kWave * 13);
    if (is_long && lane == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    }
    double cost = 0.0, xn2 = 0.0, gm = 0.0;
    int long_pt = -1;
    for (int tl = 0; tl < (LONG ? it.n_tiles : 1); ++tl) {
        const SlotCtx s = load_slot(d, it.first_tile + tl, lane);
        const int cp = d.slot_campos_g[s.slot];        // (requested with the slot record, not after the arithmetic)
        const int tile = it.first_tile + tl;           // the tile's descriptors likewise (scalar loads)
        const int stride = d.tile_stride[tile];
        const int Cg = LONG ? 0 : d.tile_ncam[tile];
        const int maxlen = d.tile_maxlen[tile];
        double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double cs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        double spk[3] = {XBA_DEAD1, XBA_DEAD1, XBA_DEAD1};   // point scaling, kept for the gradient norm after the reduction (a reload
        if (s.valid) {                                 // there would be a third dependent memory round trip)
            const CamRec& c = d.cam[s.cam];
            double q[4] = {c.q[0], c.q[1], c.q[2], c.q[3]};
            double t[3] = {c.t[0], c.t[1], c.t[2]};
            const double* P = d.P + 3 * (size_t)s.pt;
            const double Pw[3] = {P[0], P[1], P[2]};
            double M[9];
            quat_to_mat(q, M);
            Proj pr;
            project<true>(M, t, c.intr, d.cam_model[s.cam], Pw, d.slot_u[s.slot], d.slot_v[s.slot], pr);
            double rho1;
            const double rho = huber(pr.r0 * pr.r0 + pr.r1 * pr.r1, huber_a, rho1);
            cost = cost + rho;
            const double sw = sqrt(rho1);
            const double r0 = pr.r0 * sw, r1 = pr.r1 * sw;
            const unsigned cc = d.cam_const[s.cam];
            const double mq = (cc & 1u) ? 0.0 : sw, mt = (cc & 2u) ? 0.0 : sw;
            const bool pt_fixed = d.pt_const[s.pt] != 0;     // read once: a second read after the stores below is a reload (char aliases)
            const double mp = pt_fixed ? 0.0 : sw;
            const double* sc = d.scale_c + 6 * (size_t)s.cam;
            const double* sp = d.scale_p + 3 * (size_t)s.pt;
            spk[0] = sp[0]; spk[1] = sp[1]; spk[2] = sp[2];
            double F[12], E[6];
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const double* j = pr.jp + 3 * row;
                // row^T (-2 [RP]x) = -2 (row x RP)
                F[6 * row + 0] = -2.0 * (j[1] * pr.rp[2] - j[2] * pr.rp[1]) * mq * sc[0];
                F[6 * row + 1] = -2.0 * (j[2] * pr.rp[0] - j[0] * pr.rp[2]) * mq * sc[1];
                F[6 * row + 2] = -2.0 * (j[0] * pr.rp[1] - j[1] * pr.rp[0]) * mq * sc[2];
                F[6 * row + 3] = j[0] * mt * sc[3];
                F[6 * row + 4] = j[1] * mt * sc[4];
                F[6 * row + 5] = j[2] * mt * sc[5];
                E[3 * row + 0] = (j[0] * M[0] + j[1] * M[3] + j[2] * M[6]) * mp * spk[0];
                E[3 * row + 1] = (j[0] * M[1] + j[1] * M[4] + j[2] * M[7]) * mp * spk[1];
                E[3 * row + 2] = (j[0] * M[2] + j[1] * M[5] + j[2] * M[8]) * mp * spk[2];
            }
            const size_t ns = (size_t)d.n_slots;
            d.rt[s.slot] = r0; d.rt[ns + s.slot] = r1;
#pragma unroll
            for (int k = 0; k < 6; ++k) d.Jp[k * ns + s.slot] = pr.jp[k] * sw;
            v[0] = E[0] * E[0] + E[3] * E[3]; v[1] = E[0] * E[1] + E[3] * E[4]; v[2] = E[0] * E[2] + E[3] * E[5];
            v[3] = E[1] * E[1] + E[4] * E[4]; v[4] = E[1] * E[2] + E[4] * E[5]; v[5] = E[2] * E[2] + E[5] * E[5];
            v[6] = E[0] * r0 + E[3] * r1; v[7] = E[1] * r0 + E[4] * r1; v[8] = E[2] * r0 + E[5] * r1;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                cs[k] = F[k] * F[k] + F[6 + k] * F[6 + k];
                cs[6 + k] = F[k] * r0 + F[6 + k] * r1;
            }
            if (s.head && (!is_long || tl == 0) && !pt_fixed)
                xn2 += Pw[0] * Pw[0] + Pw[1] * Pw[1] + Pw[2] * Pw[2];
            if (is_long && lane == 0 && tl == 0) long_pt = s.pt;
        }
        {   // camera-side terms: one partial per camera of the tile for a regular tile (wave pre-reduction) and for a Gram tile
            // with ragged tracks (sum per distinct camera through LDS, like k_schur_pairs), one per observation otherwise
            if (stride > 0) {
                strided_reduce<12>(cs, stride, lane);
                if (cp >= 0) {
                    double2* out = reinterpret_cast<double2*>(d.scat + 12 * (size_t)cp);
#pragma unroll
                    for (int k = 0; k < 6; ++k) out[k] = make_double2(cs[2 * k], cs[2 * k + 1]);
                }
            } else if (Cg > 0) {
                extern __shared__ __attribute__((aligned(16))) double lin_smem[];
                double* red = lin_smem + (threadIdx.x >> 6) * (kWave * 13);          // this wave's [64][13]
                // (round 4: values deposited sorted by camera — position = lanes of earlier cameras + earlier lanes of the own one — so
                //  that a reducer lane walks a contiguous run with a counted loop instead of peeling a lane mask: k_schur_pairs, ba_chol.h)
                // (round 6: the lane's deposit position and the runs of the (camera, value) lanes come from the per-context tables of
                //  k_gram_runs — until round 5 a ballot loop over the tile's cameras here and in k_schur_pairs)
                const int mypos = (int)d.slot_gpos[s.slot];                          // pk: run start | length << 8 | first lane << 16
                const int* trun = d.tile_run + (size_t)tile * kTileRunLd;
                const int c0 = lane / 12, c1 = (lane + 64) / 12;
                const int pk0 = c0 < Cg ? trun[c0] : 0, pk1 = c1 < Cg ? trun[c1] : 0;
#pragma unroll
                for (int k = 0; k < 12; ++k) red[mypos * 13 + k] = cs[k];            // (lanes without an observation: row 63, in no run)
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int q = lane + 64 * rd;
                    const int pk = rd == 0 ? pk0 : pk1;
                    const bool on = q < 12 * Cg;
                    const int cpr = __shfl(cp, on ? (pk >> 16) : 0, kWave);
                    if (on) {
                        const int k = q % 12, n = (pk >> 8) & 255;
                        const double* src = red + (pk & 255) * 13 + k;
                        double sum = 0.0;
                        int j = 0;
                        for (; j + 4 <= n; j += 4) {
                            const double a0 = src[j * 13], a1 = src[(j + 1) * 13], a2 = src[(j + 2) * 13], a3 = src[(j + 3) * 13];
                            sum += a0; sum += a1; sum += a2; sum += a3;
                        }
                        for (; j < n; ++j) sum += src[j * 13];
                        d.scat[12 * (size_t)cpr + k] = sum;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            } else if (cp >= 0) {
                double2* out = reinterpret_cast<double2*>(d.scat + 12 * (size_t)cp);
#pragma unroll
                for (int k = 0; k < 6; ++k) out[k] = make_double2(cs[2 * k], cs[2 * k + 1]);
            }
        }
        seg_reduce<9>(v, s.pt, lane, maxlen);
        if (!is_long) {
            if (s.head) {
                double* H = d.Hpp + 6 * (size_t)s.pt;
#pragma unroll
                for (int k = 0; k < 6; ++k) H[k] = v[k];
                double* g = d.gp + 3 * (size_t)s.pt;
                g[0] = v[6]; g[1] = v[7]; g[2] = v[8];
                gm = fmax(gm, fmax(fabs(v[6] / spk[0]), fmax(fabs(v[7] / spk[1]), fabs(v[8] / spk[2]))));   // max-norm of the unscaled point gradient
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

// 128 VGPRs: 4 waves per SIMD (measured 118 -> 98 us at config L; 5 waves spill and are slower)
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_linearize(Dev d, double huber_a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // wave-uniform: item / tile records through scalar loads
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    if (it.n_tiles > 1) linearize_item<true>(d, it, item, lane, huber_a);
    else linearize_item<false>(d, it, item, lane, huber_a);
}

// Cost only (1/2 sum rho is formed on the host side of the reduction), candidate state.
__global__ __launch_bounds__(kBlock) void k_cost(Dev d, double huber_a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // wave-uniform: item / tile records through scalar loads
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    double cost = 0.0;
    for (int tl = 0; tl < it.n_tiles; ++tl) {
        const int slot = (it.first_tile + tl) * kWave + lane;
        const int cam = d.slot_cam[slot];
        if (cam >= 0) {
            const CamRec& c = d.cam_cand[cam];
            double q[4] = {c.q[0], c.q[1], c.q[2], c.q[3]};
            double t[3] = {c.t[0], c.t[1], c.t[2]};
            const double* P = d.P_cand + 3 * (size_t)d.slot_pt[slot];
            const double Pw[3] = {P[0], P[1], P[2]};
            double M[9];
            quat_to_mat(q, M);
            Proj pr;
            project<false>(M, t, c.intr, d.cam_model[cam], Pw, d.slot_u[slot], d.slot_v[slot], pr);
            double rho1;
            cost += huber(pr.r0 * pr.r0 + pr.r1 * pr.r1, huber_a, rho1);
        }
    }
    cost = wave_sum(cost);
    if (lane == 0) d.part[item] = cost;
}

// ---------------------------------------------------------------- camera-major segmented sum
// One workgroup per camera; scat holds K doubles per observation in camera-major
// order.  Thread (g,k) accumulates component k over observations g, g+G, ...;
// the G partials are then added in fixed order.
// WT: the result is stored write-through (agent-scope store), for a consumer in the same launch (k_lin_tail).
template <int K, bool WT = false>
__device__ __forceinline__ void segsum_body(const double* __restrict__ scat, const int* __restrict__ ptr, double* __restrict__ out, int c,
                                            double* lds_out = nullptr) {
    constexpr int G = kBlock / K;
    __shared__ double lds[G * K];
    const int t = threadIdx.x;
    const int beg = ptr[c], end = ptr[c + 1];
    if (t < G * K) {
        const int g = t / K;
        double acc = 0.0;
        const double* base = scat + (size_t)beg * K + t;
        const int n = end - beg;
        int o = g;
        for (; o + 3 * G < n; o += 4 * G) {          // four loads in flight, added in list order
            const double v0 = base[0], v1 = base[(size_t)G * K], v2 = base[(size_t)2 * G * K], v3 = base[(size_t)3 * G * K];
            acc += v0; acc += v1; acc += v2; acc += v3;
            base += (size_t)4 * G * K;
        }
        for (; o < n; o += G) { acc += *base; base += (size_t)G * K; }
        lds[t] = acc;
    }
    __syncthreads();
    if (t < K) {
        double s = 0.0;
#pragma unroll 4
        for (int g = 0; g < G; ++g) s += lds[g * K + t];
        if (WT) __hip_atomic_store(out + (size_t)c * K + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[(size_t)c * K + t] = s;
        if (lds_out) lds_out[t] = s;
    }
}

template <int K>
__global__ __launch_bounds__(kBlock) void k_cam_segsum(const double* __restrict__ scat, const int* __restrict__ cam_ptr,
                                                       double* __restrict__ out, const PcgStatus* st) {
    if (st && st->done) return;
    segsum_body<K>(scat, cam_ptr, out, blockIdx.x);
}

// Cholesky path, one launch: workgroups [0, n_cams) sum the 28 diagonal-block / rhs values of a camera, workgroups
// [n_cams, n_cams + n_blocks) the 36 values of an off-diagonal block of S (both in list order: deterministic)
__global__ __launch_bounds__(kBlock) void k_chol_segsum(const double* __restrict__ scat, const int* __restrict__ cam_ptr,
                                                        double* __restrict__ camS, int n_cams, const double* __restrict__ scat2,
                                                        const int* __restrict__ blk_ptr, double* __restrict__ Sblk) {
    if ((int)blockIdx.x < n_cams) segsum_body<28>(scat, cam_ptr, camS, blockIdx.x);
    else segsum_body<36>(scat2, blk_ptr, Sblk, blockIdx.x - n_cams);
}

// ---- S assembly of collections with long tracks (round 4): blocks formed where they are summed
// A track seen by n photos contributes n (n - 1) / 2 camera-pair blocks; an unordered collection (BASELINE config 5's shape) has a
// tail of tracks with 40-80 photos: 54 M per-pair blocks of 288 bytes, written by k_schur_pairs and read back here — 16 GB each way
// per LM iteration, 28 % of the solve.  Instead k_schur_pairs stores the OPERAND of every such observation (V = W chol(Hinv),
// 6 x 3, 144 bytes: pair_v) and the entry list of a block names its two observations (ent_src, filled once per problem by
// k_pair_sources; x < 0: the entry is a Gram tile's cell in scat2, as before); the sum kernel forms every entry's 36 elements — the
// same three products per element as the per-pair code — adds them in a fixed order and writes the block once.  The operands of a
// block's entries are re-read from L2 (each is used by n - 1 blocks).
__global__ __launch_bounds__(kWave) void k_pair_sources(Dev d, const int* __restrict__ item_list, const int* __restrict__ slot_pair_ptr,
                                                        const int* __restrict__ pair_dst, int2* __restrict__ ent_src) {
    const Item it = d.items[item_list[blockIdx.x]];
    const int s_begin = it.first_tile * kWave, s_end = s_begin + it.n_tiles * kWave;
    for (int sa = s_begin + (int)threadIdx.x; sa < s_end; sa += kWave) {
        if (d.slot_cam[sa] < 0) continue;
        const int pbase = slot_pair_ptr[sa], npair = slot_pair_ptr[sa + 1] - pbase;
        for (int dd = 1; dd <= npair; ++dd) ent_src[pair_dst[pbase + dd - 1]] = make_int2(sa, sa + dd);
    }
}
// With stored operands the scatter buffer holds the Gram tiles' cells only: every cell with a destination gets a compact record
// number — its rank among the cells with a destination, in cell order (round 6: deterministic; until round 5 an atomic counter
// numbered them in arrival order, so the record layout differed from run to run and from rank to rank: ADVICE round 5) —, which
// replaces the entry index in the tile's destination table and is noted in the entry's source record (x < 0: a Gram cell, y = its
// record).  Three tiny launches: cells with a destination per block of 256, a serial exclusive scan of the block counts (one
// workgroup; the table of a collection has a few thousand cells), rank inside the block + block offset.
__global__ __launch_bounds__(256) void k_gram_compact_count(const int* __restrict__ cell_dst, int n_cells, int* __restrict__ block_cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n_cells && cell_dst[i] >= 0;
    const int c = __syncthreads_count(on);
    if (threadIdx.x == 0) block_cnt[blockIdx.x] = c;
}
__global__ void k_gram_compact_scan(int* __restrict__ block_cnt, int n_blocks) {       // in place: counts -> exclusive offsets (one thread: a few thousand blocks at most)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int run = 0;
    for (int b = 0; b < n_blocks; ++b) { const int v = block_cnt[b]; block_cnt[b] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_gram_compact(int* __restrict__ cell_dst, int n_cells, int2* __restrict__ ent_src, const int* __restrict__ block_off) {
    __shared__ int wave_cnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int e = i < n_cells ? cell_dst[i] : -1;
    const unsigned long long m = __ballot(e >= 0);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    if (e < 0) return;
    int rec = block_off[blockIdx.x] + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) rec += wave_cnt[w];
    ent_src[e] = make_int2(-1, rec);
    cell_dst[i] = rec;
}
// One WAVE per block (four blocks per workgroup, no workgroup barrier): the sources of the next kPairVAhead entries are fetched with one
// load (lane u: entry e0 + u) and handed round as scalars; lanes 0..17 / 18..35 fetch Va / Vb with one coalesced load per entry (two
// runs of 144 bytes), park them in the wave's LDS rows, and lane k < 36 forms element k = (rb, ca) from six LDS reads.  Entries
// are added in list order: deterministic.  Measured at config T (1.65 M blocks of 33 entries on average): thread (g, k) of seven
// 36-thread groups loading its six operand values itself: 8.7 ms per launch; one workgroup per block, a wave per entry stream,
// 4 or 8 entries in flight: 5.6 ms whatever the order of the blocks (cluster-coherent or by shuffled id) — not the operands'
// locality but a chain of three dependent round trips per workgroup at full occupancy; this form: four times the blocks in flight.
constexpr int kPairVAhead = 8;
__global__ __launch_bounds__(kBlock) void k_chol_segsum_v(const double* __restrict__ scat, const int* __restrict__ cam_ptr,
                                                          double* __restrict__ camS, int n_cams, const double* __restrict__ scat2,
                                                          const int* __restrict__ blk_ptr, double* __restrict__ Sblk, int n_blocks,
                                                          const int2* __restrict__ ent_src, const double* __restrict__ pair_v) {
    if ((int)blockIdx.x < n_cams) { segsum_body<28>(scat, cam_ptr, camS, blockIdx.x); return; }
    constexpr int K = 36, NW = kBlock / kWave;
    __shared__ double ops[NW][kPairVAhead][K];
    const int t = threadIdx.x, lane = t & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = ((int)blockIdx.x - n_cams) * NW + wave;
    if (b >= n_blocks) return;
    const int beg = blk_ptr[b], end = blk_ptr[b + 1];
    const int k = lane < K ? lane : 0, rb3 = 3 * (k / 6), ca3 = 3 * (k % 6);
    double acc = 0.0;
    for (int e0 = beg; e0 < end; e0 += kPairVAhead) {
        const int2 mine = (lane < kPairVAhead && e0 + lane < end) ? ent_src[e0 + lane] : make_int2(-1, -1);
        int sx[kPairVAhead], sy[kPairVAhead];
        double v[kPairVAhead];
#pragma unroll
        for (int u = 0; u < kPairVAhead; ++u) {
            sx[u] = __builtin_amdgcn_readlane(mine.x, u); sy[u] = __builtin_amdgcn_readlane(mine.y, u);
            v[u] = 0.0;
            if (e0 + u < end && lane < K)                    // (the first condition is wave-uniform)
                v[u] = sx[u] < 0 ? scat2[36 * (size_t)sy[u] + lane] : pair_v[18 * (size_t)(lane < 18 ? sx[u] : sy[u]) + (lane < 18 ? lane : lane - 18)];
        }
#pragma unroll
        for (int u = 0; u < kPairVAhead; ++u) {
            if (e0 + u >= end) break;
            if (sx[u] < 0) { acc += v[u]; continue; }
            double* row = ops[wave][u];
            if (lane < K) row[lane] = v[u];
            __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): LDS operations of one wave complete in order
            __builtin_amdgcn_wave_barrier();
            const double* Va = row + ca3;
            const double* Vb = row + 18 + rb3;
            acc += Vb[0] * Va[0] + Vb[1] * Va[1] + Vb[2] * Va[2];
        }
    }
    if (lane < K) Sblk[(size_t)b * K + lane] = acc;
}

// Diagnostics only: materialise the 2x6 / 2x3 blocks (SoA over slots) the consumers rebuild on the fly.
__global__ void k_debug_materialize(Dev d, double* __restrict__ Fs, double* __restrict__ Es) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= d.n_slots || d.slot_cam[slot] < 0) return;
    double F[12], E[6];
    load_FE(d, slot, d.slot_cam[slot], d.slot_pt[slot], F, E);
    const size_t ns = (size_t)d.n_slots;
    for (int k = 0; k < 12; ++k) Fs[k * ns + slot] = F[k];
    for (int k = 0; k < 6; ++k) Es[k * ns + slot] = E[k];
}

// Jacobi scaling from the column norms of the unscaled Jacobian: 1/(1+|col|).
__global__ void k_scale_from_norms(Dev d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.n_cams * 6) {
        const int c = i / 6, k = i % 6;
        d.scale_c[i] = 1.0 / (1.0 + sqrt(d.camlin[12 * (size_t)c + k]));
    }
    if (i < d.n_pts * 3) {
        const int p = i / 3, k = i % 3;
        const int di = (k == 0) ? 0 : (k == 1 ? 3 : 5);
        d.scale_p[i] = 1.0 / (1.0 + sqrt(d.Hpp[6 * (size_t)p + di]));
    }
}

__global__ void k_fill(double* p, double v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------- LM damping + point block inverse
// (workgroups >= n_pt_blocks: the LM diagonal of the camera blocks)
__global__ void k_point_prep(Dev d, double radius, double dmin, double dmax, int n_pt_blocks) {
    if ((int)blockIdx.x >= n_pt_blocks) {
        const int i = (blockIdx.x - n_pt_blocks) * blockDim.x + threadIdx.x;
        if (i < d.n_cams * 6) d.Dc2[i] = clampd(d.camlin[12 * (size_t)(i / 6) + i % 6], dmin, dmax) / radius;
        return;
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.n_pts) return;
    const double* H = d.Hpp + 6 * (size_t)p;
    double h[6] = {H[0], H[1], H[2], H[3], H[4], H[5]};
    h[0] += clampd(h[0], dmin, dmax) / radius;
    h[3] += clampd(h[3], dmin, dmax) / radius;
    h[5] += clampd(h[5], dmin, dmax) / radius;
    double inv[6];
    sym3_inverse(h, inv);
    double* o = d.Hinv + 6 * (size_t)p;
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = inv[k];
    // Hinv = C C^T (Hinv is SPD: the damped point block is): W Hinv W^T = (W C)(W C)^T, so the S assembly stages ONE
    // operand matrix for its Gram products
    const double c00 = sqrt(fmax(inv[0], 0.0)), r0 = c00 > 0.0 ? 1.0 / c00 : 0.0;
    const double c10 = inv[1] * r0, c20 = inv[2] * r0;
    const double c11 = sqrt(fmax(inv[3] - c10 * c10, 0.0)), r1 = c11 > 0.0 ? 1.0 / c11 : 0.0;
    const double c21 = (inv[4] - c20 * c10) * r1;
    const double c22 = sqrt(fmax(inv[5] - c20 * c20 - c21 * c21, 0.0));
    double* oc = d.Hc + 6 * (size_t)p;
    oc[0] = c00; oc[1] = c10; oc[2] = c20; oc[3] = c11; oc[4] = c21; oc[5] = c22;
}

// Per observation: block-Jacobi diagonal block and reduced right-hand side terms
//   S_cc += F^T F - (F^T E) Hinv (E^T F),   rb -= (F^T E) Hinv g_p
__global__ __launch_bounds__(kBlock) void k_schur_prep(Dev d) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;      // blocks of 256 slots = 4 whole tiles, wave = tile
    if (slot >= d.n_slots) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int cam = d.slot_cam[slot];
    double o[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) o[k] = 0.0;
    if (cam >= 0) {
        const int pt = d.slot_pt[slot];
        double F[12], E[6];
        load_FE(d, slot, cam, d.slot_pt[slot], F, E);
        const double* Hi = d.Hinv + 6 * (size_t)pt;
        const double h[6] = {Hi[0], Hi[1], Hi[2], Hi[3], Hi[4], Hi[5]};
        const double* g = d.gp + 3 * (size_t)pt;
        const double g0 = g[0], g1 = g[1], g2 = g[2];
        double W[18], WH[18];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = 0; b < 3; ++b) W[3 * a + b] = F[a] * E[b] + F[6 + a] * E[3 + b];
            WH[3 * a + 0] = W[3 * a] * h[0] + W[3 * a + 1] * h[1] + W[3 * a + 2] * h[2];
            WH[3 * a + 1] = W[3 * a] * h[1] + W[3 * a + 1] * h[3] + W[3 * a + 2] * h[4];
            WH[3 * a + 2] = W[3 * a] * h[2] + W[3 * a + 1] * h[4] + W[3 * a + 2] * h[5];
        }
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c)
                o[idx++] = F[a] * F[c] + F[6 + a] * F[6 + c]
                           - (WH[3 * a] * W[3 * c] + WH[3 * a + 1] * W[3 * c + 1] + WH[3 * a + 2] * W[3 * c + 2]);
#pragma unroll
        for (int a = 0; a < 6; ++a) o[21 + a] = -(WH[3 * a] * g0 + WH[3 * a + 1] * g1 + WH[3 * a + 2] * g2);
    }
    const int stride = d.tile_stride[slot >> 6];
    if (stride > 0) strided_reduce<28>(o, stride, lane);
    const int cp = d.slot_campos[slot];
    if (cp >= 0) {
        double2* out = reinterpret_cast<double2*>(d.scat + 28 * (size_t)cp);
#pragma unroll
        for (int k = 0; k < 14; ++k) out[k] = make_double2(o[2 * k], o[2 * k + 1]);
    }
}

// Per camera: M = S_cc + D_c^2, store M^-1 (symmetric, 21) via Cholesky; b = g_c + rb.
__global__ void k_cam_factor(Dev d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n_cams) return;
    const double* S = d.camS + 28 * (size_t)c;
    double A[6][6];
    int idx = 0;
    for (int a = 0; a < 6; ++a)
        for (int b2 = a; b2 < 6; ++b2) { A[a][b2] = S[idx]; A[b2][a] = S[idx]; ++idx; }
    for (int a = 0; a < 6; ++a) A[a][a] += d.Dc2[6 * (size_t)c + a];
    // Cholesky A = L L^T (lower, in place)
    double L[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) L[i][j] = 0.0;
    for (int j = 0; j < 6; ++j) {
        double s = A[j][j];
        for (int k = 0; k < j; ++k) s -= L[j][k] * L[j][k];
        const double ljj = sqrt(s);
        L[j][j] = ljj;
        const double inv = 1.0 / ljj;
        for (int i = j + 1; i < 6; ++i) {
            double v = A[i][j];
            for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
            L[i][j] = v * inv;
        }
    }
    // Linv (lower)
    double Li[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) Li[i][j] = 0.0;
    for (int j = 0; j < 6; ++j) {
        Li[j][j] = 1.0 / L[j][j];
        for (int i = j + 1; i < 6; ++i) {
            double v = 0.0;
            for (int k = j; k < i; ++k) v -= L[i][k] * Li[k][j];
            Li[i][j] = v / L[i][i];
        }
    }
    // Minv = Li^T Li
    double* Mo = d.Minv + 21 * (size_t)c;
    idx = 0;
    for (int a = 0; a < 6; ++a)
        for (int b2 = a; b2 < 6; ++b2) {
            double v = 0.0;
            for (int k = b2; k < 6; ++k) v += Li[k][a] * Li[k][b2];
            Mo[idx++] = v;
        }
    for (int a = 0; a < 6; ++a) d.b[6 * (size_t)c + a] = d.camlin[12 * (size_t)c + 6 + a] + S[21 + a];
}

// ---------------------------------------------------------------- PCG
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* lds) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < NT / kWave; ++i) s += lds[i];
    return s;  // same value in every thread
}

__device__ __forceinline__ void sym6_mul(const double* __restrict__ m, const double* x, double* y) {
    // m: upper triangle row-major (21); fully unrolled, the packed index folds to a constant (no private array)
    double mm[21];
#pragma unroll
    for (int k = 0; k < 21; ++k) mm[k] = m[k];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            s += mm[6 * lo - lo * (lo - 1) / 2 + (hi - lo)] * x[b];
        }
        y[a] = s;
    }
}

constexpr int kPcgThreads = 1024;

// ---- two-level preconditioner (round 5): block-Jacobi + the GAUGE coarse space
// The reduced camera matrix has (up to) seven eigenvalues at the level of the LM damping: a similarity transform of the whole scene
// — translation (3), rotation (3), scale (1) — changes no residual, and what two constant translations (the reference's gauge,
// ba_solver.cc:611-614) leave of it is held by the damping alone.  Block-Jacobi does nothing for them: the iteration count grows as
// the trust region opens (config V: 37 PCG iterations at the first LM step, 85 at the fourth, 365 on average over a solve).  With
// M^-1 = blockdiag(S_cc)^-1 + W (W^T S W)^-1 W^T and W = those seven vectors restricted to the cameras (closed form below, in the
// Jacobi-scaled tangent coordinates, rows of constant blocks zeroed) the count stays at ~15 whatever the radius (tools/pcg_coarse.py).
//   world:  X' = X + w x X + tau + sigma X   =>   camera (R, t):  delta_q = -1/2 R w  (Plus(q, d) = dq(d) * q),  delta_t = sigma t - R tau
constexpr int kGauge = 7;
__global__ void k_pcg_gauge(Dev d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n_cams) return;
    const CamRec& cur = d.cam[c];
    const double q[4] = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
    double M[9];
    quat_to_mat(q, M);
    const unsigned cc = d.cam_const[c];
    const bool active = d.cam_act[c] > 0.0;
    const double* sc = d.scale_c + 6 * (size_t)c;
    const double mq = (active && !(cc & 1u)) ? 1.0 : 0.0, mt = (active && !(cc & 2u)) ? 1.0 : 0.0;
    const size_t n6 = 6 * (size_t)d.n_cams;
    for (int k = 0; k < kGauge; ++k) {
        double v[6] = {0, 0, 0, 0, 0, 0};
        if (k < 3) { v[3] = -M[k]; v[4] = -M[3 + k]; v[5] = -M[6 + k]; }                          // tau = e_k
        else if (k < 6) { v[0] = -0.5 * M[k - 3]; v[1] = -0.5 * M[3 + k - 3]; v[2] = -0.5 * M[6 + k - 3]; }   // w = e_k
        else { v[3] = cur.t[0]; v[4] = cur.t[1]; v[5] = cur.t[2]; }                               // sigma = 1
        double* out = d.pcgW + k * n6 + 6 * (size_t)c;
#pragma unroll
        for (int j = 0; j < 6; ++j) out[j] = (j < 3 ? mq : mt) * v[j] / sc[j];                  // scaled coordinates: x = scale * y
    }
}
// SW_k += D_c^2 W_k (the product kernels leave S W_k there)
__global__ void k_pcg_gauge_damp(Dev d) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n6 = 6 * (size_t)d.n_cams;
    if (i >= n6) return;
#pragma unroll
    for (int k = 0; k < kGauge; ++k) d.pcgSW[k * n6 + i] += d.Dc2[i] * d.pcgW[k * n6 + i];
}
// E = W^T (S + D^2) W (symmetrised), Cholesky with a guard — a direction whose pivot is not positive relative to its diagonal
// entry (a gauge freedom every camera of which is constant) is dropped — and its inverse to d.pcgE.  One workgroup.
__global__ __launch_bounds__(kPcgThreads) void k_pcg_coarse(Dev d) {
    __shared__ double lds[kPcgThreads / kWave];
    __shared__ double E[kGauge][kGauge], L[kGauge][kGauge], Li[kGauge][kGauge];
    __shared__ int keep[kGauge];
    const size_t n6 = 6 * (size_t)d.n_cams;
#pragma clang loop unroll(disable)
    for (int a = 0; a < kGauge; ++a)
#pragma clang loop unroll(disable)
        for (int b = 0; b <= a; ++b) {
            double sum = 0.0;
            for (size_t i = threadIdx.x; i < n6; i += kPcgThreads) sum += 0.5 * (d.pcgW[a * n6 + i] * d.pcgSW[b * n6 + i] + d.pcgW[b * n6 + i] * d.pcgSW[a * n6 + i]);
            sum = block_sum<kPcgThreads>(sum, lds);
            if (threadIdx.x == 0) { E[a][b] = sum; E[b][a] = sum; }
        }
    __syncthreads();
    if (threadIdx.x == 0) {        // (one thread, 7 x 7, LDS-resident: a few hundred scalar operations per LM step; loops kept rolled)
#pragma clang loop unroll(disable)
        for (int i = 0; i < kGauge; ++i)
    #pragma clang loop unroll(disable)
        for (int j = 0; j < kGauge; ++j) { L[i][j] = 0.0; Li[i][j] = 0.0; }
#pragma clang loop unroll(disable)
        for (int j = 0; j < kGauge; ++j) {
            double sjj = E[j][j];
    #pragma clang loop unroll(disable)
        for (int k = 0; k < j; ++k) sjj -= L[j][k] * L[j][k];
            keep[j] = (E[j][j] > 0.0 && sjj > 1e-12 * E[j][j]) ? 1 : 0;
            if (!keep[j]) continue;                              // row / column j stay zero: the direction takes no part
            const double ljj = sqrt(sjj);
            L[j][j] = ljj;
    #pragma clang loop unroll(disable)
        for (int i = j + 1; i < kGauge; ++i) {
                double v = E[i][j];
        #pragma clang loop unroll(disable)
        for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
                L[i][j] = v / ljj;
            }
        }
#pragma clang loop unroll(disable)
        for (int j = 0; j < kGauge; ++j) {                      // Li = L^-1 on the kept directions
            if (!keep[j]) continue;
            Li[j][j] = 1.0 / L[j][j];
    #pragma clang loop unroll(disable)
        for (int i = j + 1; i < kGauge; ++i) {
                if (!keep[i]) continue;
                double v = 0.0;
        #pragma clang loop unroll(disable)
        for (int k = j; k < i; ++k) v -= L[i][k] * Li[k][j];
                Li[i][j] = v / L[i][i];
            }
        }
#pragma clang loop unroll(disable)
        for (int a = 0; a < kGauge; ++a)
    #pragma clang loop unroll(disable)
        for (int b = 0; b < kGauge; ++b) {
                double v = 0.0;
        #pragma clang loop unroll(disable)
        for (int k = 0; k < kGauge; ++k) v += Li[k][a] * Li[k][b];
                d.pcgE[a * kGauge + b] = v;
            }
    }
}
// z += W (E^-1 wr): the coarse term of the preconditioner for one camera; returns nothing, wr: the seven reduced dot products W_k.r
__device__ __forceinline__ void pcg_coarse_coef(const double* Einv, const double* wr, double* coef) {
#pragma unroll
    for (int a = 0; a < kGauge; ++a) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < kGauge; ++b) v += Einv[a * kGauge + b] * wr[b];
        coef[a] = v;
    }
}

// x = 0, r = b, z = M^-1 r, p = z
__global__ __launch_bounds__(kPcgThreads) void k_pcg_init(Dev d) {
    __shared__ double lds[kPcgThreads / kWave];
    double rz = 0.0, bb = 0.0;
    const size_t n6 = 6 * (size_t)d.n_cams;
    double wr[kGauge] = {0, 0, 0, 0, 0, 0, 0};
    for (int c = threadIdx.x; c < d.n_cams; c += kPcgThreads) {
        double r[6], z[6];
        for (int k = 0; k < 6; ++k) r[k] = d.b[6 * (size_t)c + k];
        sym6_mul(d.Minv + 21 * (size_t)c, r, z);
        for (int k = 0; k < 6; ++k) {
            d.px[6 * (size_t)c + k] = 0.0; d.pr[6 * (size_t)c + k] = r[k];
            d.pz[6 * (size_t)c + k] = z[k]; d.pp[6 * (size_t)c + k] = z[k];
            rz += r[k] * z[k]; bb += r[k] * r[k];
        }
        if (d.pcgW)
            for (int g = 0; g < kGauge; ++g)
                for (int k = 0; k < 6; ++k) wr[g] += d.pcgW[g * n6 + 6 * (size_t)c + k] * r[k];
    }
    rz = block_sum<kPcgThreads>(rz, lds);
    bb = block_sum<kPcgThreads>(bb, lds);
    if (d.pcgW) {        // z = Minv r + W E^-1 W^T r,  r.z = r.Minv r + (W^T r).(E^-1 W^T r)
        for (int g = 0; g < kGauge; ++g) wr[g] = block_sum<kPcgThreads>(wr[g], lds);
        double coef[kGauge];
        pcg_coarse_coef(d.pcgE, wr, coef);
        for (int g = 0; g < kGauge; ++g) rz += wr[g] * coef[g];
        for (int c = threadIdx.x; c < d.n_cams; c += kPcgThreads)
            for (int k = 0; k < 6; ++k) {
                double add = 0.0;
                for (int g = 0; g < kGauge; ++g) add += d.pcgW[g * n6 + 6 * (size_t)c + k] * coef[g];
                d.pp[6 * (size_t)c + k] += add;
            }
    }
    if (threadIdx.x == 0) {
        d.st->rz = rz; d.st->rr = bb; d.st->bb = bb; d.st->pq = 0.0; d.st->it = 0; d.st->arrived = 0u;
        d.st->done = (bb == 0.0) ? 1 : 0;
    }
}

// Implicit Schur product, track side:  per observation v = F p_c, w = E^T v,
// per track u = Hinv sum w (segmented wave reduction), z = v - E u, and the
// camera-side term F^T z goes to the camera-major scatter buffer.
__global__ __launch_bounds__(kBlock) void k_schur_matvec(Dev d, const double* __restrict__ pvec) {
    if (d.st->done) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // wave-uniform: item / tile records through scalar loads
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    if (it.n_tiles == 1) {
        const SlotCtx s = load_slot(d, it.first_tile, lane);
        double F[12], E[6], v0, v1;
        double w[3] = {0, 0, 0};                   // operand of the segmented sum: zeros on lanes without an observation
        if (s.valid) {
            load_FE(d, s.slot, s.cam, s.pt, F, E);
            v0 = 0.0; v1 = 0.0;
            const double* p = pvec + 6 * (size_t)s.cam;
#pragma unroll
            for (int k = 0; k < 6; ++k) { const double pk = p[k]; v0 += F[k] * pk; v1 += F[6 + k] * pk; }
            w[0] = E[0] * v0 + E[3] * v1; w[1] = E[1] * v0 + E[4] * v1; w[2] = E[2] * v0 + E[5] * v1;
        } else { dead_values(F); dead_values(E); XBA_DEAD_VALUE(v0); XBA_DEAD_VALUE(v1); }
        seg_reduce<3>(w, s.pt, lane, d.tile_maxlen[it.first_tile]);
        double u[3];
        if (s.head) {
            const double* h = d.Hinv + 6 * (size_t)s.pt;
            u[0] = h[0] * w[0] + h[1] * w[1] + h[2] * w[2];
            u[1] = h[1] * w[0] + h[3] * w[1] + h[4] * w[2];
            u[2] = h[2] * w[0] + h[4] * w[1] + h[5] * w[2];
        } else dead_values(u);
        const int hl = seg_head_lane(s.head || !s.valid, lane);
        u[0] = __shfl(u[0], hl, kWave); u[1] = __shfl(u[1], hl, kWave); u[2] = __shfl(u[2], hl, kWave);
        double y[6] = {0, 0, 0, 0, 0, 0};
        if (s.valid) {
            const double z0 = v0 - (E[0] * u[0] + E[1] * u[1] + E[2] * u[2]);
            const double z1 = v1 - (E[3] * u[0] + E[4] * u[1] + E[5] * u[2]);
#pragma unroll
            for (int k = 0; k < 6; ++k) y[k] = F[k] * z0 + F[6 + k] * z1;
        }
        const int stride = d.tile_stride[it.first_tile];
        if (stride > 0) strided_reduce<6>(y, stride, lane);
        const int cp = d.slot_campos[s.slot];
        if (cp >= 0) {
            double2* out = reinterpret_cast<double2*>(d.scat + 6 * (size_t)cp);
            out[0] = make_double2(y[0], y[1]); out[1] = make_double2(y[2], y[3]); out[2] = make_double2(y[4], y[5]);
        }
        return;
    }
    // long track: all lanes of all tiles belong to one track
    double wsum[3] = {0, 0, 0};
    int pt0 = -1;
    for (int tl = 0; tl < it.n_tiles; ++tl) {
        const int slot = (it.first_tile + tl) * kWave + lane;
        const int cam = d.slot_cam[slot];
        if (cam >= 0) {
            pt0 = d.slot_pt[slot];
            const double* p = pvec + 6 * (size_t)cam;
            double F[12], E[6];
            load_FE(d, slot, cam, pt0, F, E);
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { v0 += F[k] * p[k]; v1 += F[6 + k] * p[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) wsum[k] += E[k] * v0 + E[3 + k] * v1;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { wsum[k] = wave_sum(wsum[k]); wsum[k] = __shfl(wsum[k], 0, kWave); }
    pt0 = __shfl(pt0, 0, kWave);
    const double* h = d.Hinv + 6 * (size_t)pt0;
    const double u0 = h[0] * wsum[0] + h[1] * wsum[1] + h[2] * wsum[2];
    const double u1 = h[1] * wsum[0] + h[3] * wsum[1] + h[4] * wsum[2];
    const double u2 = h[2] * wsum[0] + h[4] * wsum[1] + h[5] * wsum[2];
    for (int tl = 0; tl < it.n_tiles; ++tl) {
        const int slot = (it.first_tile + tl) * kWave + lane;
        const int cam = d.slot_cam[slot];
        if (cam >= 0) {
            double F[12], E[6];
            load_FE(d, slot, cam, d.slot_pt[slot], F, E);
            const double* p = pvec + 6 * (size_t)cam;
            double v0 = 0.0, v1 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { v0 += F[k] * p[k]; v1 += F[6 + k] * p[k]; }
            const double z0 = v0 - (E[0] * u0 + E[1] * u1 + E[2] * u2);
            const double z1 = v1 - (E[3] * u0 + E[4] * u1 + E[5] * u2);
            double* out = d.scat + 6 * (size_t)d.slot_campos[slot];
#pragma unroll
            for (int k = 0; k < 6; ++k) out[k] = F[k] * z0 + F[6 + k] * z1;
        }
    }
}

// One PCG iteration's vector work on the camera vectors, three short multi-workgroup launches (thread = camera).  The dot
// products are deterministic: per-camera partials, and EVERY workgroup of the next launch adds the partial array in the same
// fixed order (a few thousand values from L2) instead of waiting for a single-workgroup reduction kernel.
constexpr int kPcgBlock = 256;
__device__ __forceinline__ double reduce_partials(const double* __restrict__ a, int n, double* lds) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kPcgBlock) s += a[i];
    return block_sum<kPcgBlock>(s, lds);
}
// N partial arrays (n values each, `stride` apart) in ONE pass: N loads in flight per step and one workgroup barrier pair for all of
// them (nine reductions one after the other cost k_pcg_p 12 us of its 20).  lds: N * kPcgBlock / kWave doubles.  Every thread gets all sums.
template <int N>
__device__ __forceinline__ void reduce_partials_multi(const double* __restrict__ a, size_t stride, int n, double* lds, double (&out)[N]) {
    double s[N];
#pragma unroll
    for (int j = 0; j < N; ++j) s[j] = 0.0;
    for (int i = threadIdx.x; i < n; i += kPcgBlock) {
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] += a[j * stride + i];
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < N; ++j) s[j] = wave_sum(s[j]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < N; ++j) lds[j * (kPcgBlock / kWave) + w] = s[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double v = 0.0;
#pragma unroll
        for (int i = 0; i < kPcgBlock / kWave; ++i) v += lds[j * (kPcgBlock / kWave) + i];
        out[j] = v;
    }
}

// (1) q = S p (all-reduced sum over the observations) + D_c^2 p;  partial p.q per camera
__global__ __launch_bounds__(kPcgBlock) void k_pcg_q(Dev d, double* __restrict__ part) {
    if (d.st->done) return;
    const int c = blockIdx.x * kPcgBlock + threadIdx.x;
    if (c >= d.n_cams) return;
    double pq = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const size_t i = 6 * (size_t)c + k;
        const double pv = d.pp[i];
        const double qv = d.pq[i] + d.Dc2[i] * pv;
        d.pq[i] = qv;
        pq += pv * qv;
    }
    part[c] = pq;
}

// (1') one rank: (1) inside the per-camera sum of the product — the workgroup that adds a camera's partials of S p holds its six values of q
__global__ __launch_bounds__(kBlock) void k_pcg_segsum_q(Dev d, double* __restrict__ part) {
    if (d.st->done) return;
    __shared__ double q6[6];
    const int c = blockIdx.x;
    segsum_body<6>(d.scat, d.cam_ptr, d.pq, c, q6);
    __syncthreads();
    if (threadIdx.x < kWave) {
        double pq = 0.0;
        if (threadIdx.x < 6) {
            const size_t i = 6 * (size_t)c + threadIdx.x;
            const double pv = d.pp[i];
            const double qv = q6[threadIdx.x] + d.Dc2[i] * pv;
            d.pq[i] = qv;
            pq = pv * qv;
        }
        // lanes 0..5 in lane order, as k_pcg_q adds them (k = 0..5): ((((p0 q0 + p1 q1) + p2 q2) + ...)
        double acc = __shfl(pq, 0, kWave);
#pragma unroll
        for (int k = 1; k < 6; ++k) acc += __shfl(pq, k, kWave);
        if (threadIdx.x == 0) part[c] = acc;
    }
}

// (2) alpha = rz / p.q;  x += alpha p, r -= alpha q, z = M^-1 r;  partial r.z and r.r per camera
__global__ __launch_bounds__(kPcgBlock) void k_pcg_xr(Dev d, double* __restrict__ part) {
    if (d.st->done) return;
    __shared__ double lds[kPcgBlock / kWave];
    const double pq = reduce_partials(part, d.n_cams, lds);
    const double alpha = d.st->rz / pq;
    const int c = blockIdx.x * kPcgBlock + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) d.st->pq = pq;
    if (c >= d.n_cams) return;
    double r[6], z[6], rz = 0.0, rr = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const size_t i = 6 * (size_t)c + k;
        d.px[i] += alpha * d.pp[i];
        r[k] = d.pr[i] - alpha * d.pq[i];
        d.pr[i] = r[k];
    }
    sym6_mul(d.Minv + 21 * (size_t)c, r, z);
#pragma unroll
    for (int k = 0; k < 6; ++k) { d.pz[6 * (size_t)c + k] = z[k]; rz += r[k] * z[k]; rr += r[k] * r[k]; }
    part[(size_t)d.n_cams + c] = rz;
    part[2 * (size_t)d.n_cams + c] = rr;
    if (d.pcgW) {        // the camera's share of W_k.r (the coarse term of z follows in k_pcg_p, once the seven sums are known)
        const size_t n6 = 6 * (size_t)d.n_cams;
#pragma unroll
        for (int g = 0; g < kGauge; ++g) {
            double w = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) w += d.pcgW[g * n6 + 6 * (size_t)c + k] * r[k];
            part[(3 + g) * (size_t)d.n_cams + c] = w;
        }
    }
}

// (3) beta = rz_new / rz;  p = z + beta p;  status
__global__ __launch_bounds__(kPcgBlock) void k_pcg_p(Dev d, const double* __restrict__ part, double tol, int max_it) {
    if (d.st->done) return;
    __shared__ double lds[kPcgBlock / kWave];
    double rz_new, rr;
    double coef[kGauge] = {0, 0, 0, 0, 0, 0, 0};
    if (d.pcgW) {        // z = Minv r + W E^-1 W^T r: every workgroup adds the nine partial arrays in the same order
        __shared__ double lds9[(2 + kGauge) * (kPcgBlock / kWave)];
        __shared__ double sE[kGauge * kGauge];
        if (threadIdx.x < kGauge * kGauge) sE[threadIdx.x] = d.pcgE[threadIdx.x];      // (visible after the barriers of the reduction)
        double sums[2 + kGauge], wr[kGauge];
        reduce_partials_multi<2 + kGauge>(part + d.n_cams, (size_t)d.n_cams, d.n_cams, lds9, sums);
        rz_new = sums[0]; rr = sums[1];
#pragma unroll
        for (int g = 0; g < kGauge; ++g) wr[g] = sums[2 + g];
        pcg_coarse_coef(sE, wr, coef);
#pragma unroll
        for (int g = 0; g < kGauge; ++g) rz_new += wr[g] * coef[g];
    } else {
        rz_new = reduce_partials(part + d.n_cams, d.n_cams, lds);
        rr = reduce_partials(part + 2 * (size_t)d.n_cams, d.n_cams, lds);
    }
    const double rz = d.st->rz, bb = d.st->bb, pq = d.st->pq;
    const int it0 = d.st->it;
    const double beta = rz_new / rz;
    const int c = blockIdx.x * kPcgBlock + threadIdx.x;
    if (c < d.n_cams) {
        const size_t n6 = 6 * (size_t)d.n_cams;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const size_t i = 6 * (size_t)c + k;
            double z = d.pz[i];
            if (d.pcgW) {
#pragma unroll
                for (int g = 0; g < kGauge; ++g) z += d.pcgW[g * n6 + i] * coef[g];
            }
            d.pp[i] = z + beta * d.pp[i];
        }
    }
    // the status is written by the LAST workgroup to get here, after every workgroup has read the old one
    __shared__ bool last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last = (atomicAdd(&d.st->arrived, 1u) + 1u == gridDim.x);
    __syncthreads();
    if (last && threadIdx.x == 0) {
        d.st->rz = rz_new; d.st->rr = rr; d.st->it = it0 + 1; d.st->arrived = 0u;
        d.st->done = (rr <= tol * tol * bb || it0 + 1 >= max_it || !(pq > 0.0)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------- back-substitution + update
// Candidate cameras: Plus(x, -y*scale); |dx|^2 and |x|^2 per camera (ambient, variable blocks only).
// y: the camera's 6 entries of the (scaled) solution.  camrec_cand (optional): the candidate's linearisation record is
// written as well (the level-scheduled backward substitution produces the candidate cameras tile by tile, ba_chol.h).
__device__ __forceinline__ void cam_update_one(const Dev& d, int c, const double* y, CamLin* __restrict__ camrec_cand) {
    // (field by field: a `CamRec nxt = cur` copy lived in scratch memory — 96 bytes of private segment for the whole kernel)
    const CamRec& cur = d.cam[c];
    const double q[4] = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
    const double t[3] = {cur.t[0], cur.t[1], cur.t[2]};
    double tn[3] = {t[0], t[1], t[2]};
    const unsigned cc = d.cam_const[c];
    const bool active = d.cam_act[c] > 0.0;
    const double* sc = d.scale_c + 6 * (size_t)c;
    double step2 = 0.0, xn2 = 0.0;
    const bool rot = active && !(cc & 1u);
    double qn[4] = {q[0], q[1], q[2], q[3]};
    if (rot) {
        const double dl[3] = {-y[0] * sc[0], -y[1] * sc[1], -y[2] * sc[2]};
        quat_plus(q, dl, qn);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double df = qn[k] - q[k]; step2 += df * df; xn2 += q[k] * q[k]; }
    }
    if (active && !(cc & 2u)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tn[k] = t[k] + (-y[3 + k] * sc[3 + k]);
            const double df = tn[k] - t[k]; step2 += df * df; xn2 += t[k] * t[k];
        }
    }
    CamRec& out = d.cam_cand[c];
#pragma unroll
    for (int k = 0; k < 4; ++k) out.q[k] = qn[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) out.t[k] = tn[k];
    out.pad = cur.pad;
#pragma unroll
    for (int k = 0; k < 8; ++k) out.intr[k] = cur.intr[k];
    d.campart[c] = step2;
    d.campart[d.n_cams + c] = xn2;
    if (camrec_cand) cam_lin_one(d, c, qn, camrec_cand);
}

// y_p = Hinv (g_p - sum E^T F y_c); model cost change; candidate points.
// Workgroups >= n_item_blocks: candidate cameras from the camera part of the solution (thread = camera; cam_update_one).
// PREP = true: the damped point block is factored here from Hpp and the radius (point_factor(), u = C (C^T a)); false: read
// Hinv as k_point_prep stored it (PCG path, XRSFM_BA_PREP_FUSED=0).
template <bool PREP>
#if XBA_BACKSUB_WAVES > 0
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(XBA_BACKSUB_WAVES, XBA_BACKSUB_WAVES)))
#else
__global__ __launch_bounds__(kBlock)
#endif
void k_backsub(Dev d, int n_item_blocks, CamLin* __restrict__ camrec_cand, double radius) {
    if ((int)blockIdx.x >= n_item_blocks) {
        const int c = (blockIdx.x - n_item_blocks) * kBlock + threadIdx.x;
        if (c < d.n_cams) cam_update_one(d, c, d.px + 6 * (size_t)c, camrec_cand);
        return;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));   // wave-uniform: item / tile records through scalar loads
    if (item >= d.n_items) return;
    const Item it = d.items[item];
    const size_t ns = (size_t)d.n_slots;
    double model = 0.0, step2 = 0.0;
    if (it.n_tiles == 1) {
        const SlotCtx s = load_slot(d, it.first_tile, lane);
        const int maxlen = d.tile_maxlen[it.first_tile];
        double E[6], v0, v1, r0, r1;                                            // read by lanes with an observation only
        double w[3] = {0, 0, 0};                                                // operand of the segmented sum: zeros on the other lanes
        // what the head lane of a track needs after the reduction depends on the point only: every lane of the track requests
        // it now (same addresses: one transaction), so that it does not cost a third memory round trip after the shuffles
        double hh[6], gg[3], spv[3], Pv[3];
        bool var = false;
        if (s.valid) {
            const double* h = (PREP ? d.Hpp : d.Hinv) + 6 * (size_t)s.pt;
            const double* g = d.gp + 3 * (size_t)s.pt;
            const double* sp = d.scale_p + 3 * (size_t)s.pt;
            const double* P = d.P + 3 * (size_t)s.pt;
#pragma unroll
            for (int k = 0; k < 6; ++k) hh[k] = h[k];
            gg[0] = g[0]; gg[1] = g[1]; gg[2] = g[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) { spv[k] = sp[k]; Pv[k] = P[k]; }
            var = !d.pt_const[s.pt];
        } else { dead_values(hh); dead_values(gg); dead_values(spv); dead_values(Pv); }
        if (s.valid) {
            const double* yp = d.px + 6 * (size_t)s.cam;
            const double y[6] = {yp[0], yp[1], yp[2], yp[3], yp[4], yp[5]};
            r0 = d.rt[s.slot]; r1 = d.rt[ns + s.slot];
            double F[12];
            load_FE(d, s.slot, s.cam, s.pt, F, E);
            v0 = 0.0; v1 = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) { v0 += F[k] * y[k]; v1 += F[6 + k] * y[k]; }
            w[0] = E[0] * v0 + E[3] * v1; w[1] = E[1] * v0 + E[4] * v1; w[2] = E[2] * v0 + E[5] * v1;
        } else { dead_values(E); XBA_DEAD_VALUE(v0); XBA_DEAD_VALUE(v1); XBA_DEAD_VALUE(r0); XBA_DEAD_VALUE(r1); }
        seg_reduce<3>(w, s.pt, lane, maxlen);
        double u[3];                                       // head lanes compute it, the lanes of the track fetch it from their head
        if (s.head) {
            const double a0 = gg[0] - w[0], a1 = gg[1] - w[1], a2 = gg[2] - w[2];
            if (PREP) {
                double cf[6];
                point_factor(hh, radius, cf);                     // Hinv = C C^T, C upper {c00 c01 c02 c11 c12 c22}
                const double s0 = cf[0] * a0, s1 = cf[1] * a0 + cf[3] * a1, s2 = cf[2] * a0 + cf[4] * a1 + cf[5] * a2;
                u[0] = cf[0] * s0 + cf[1] * s1 + cf[2] * s2;
                u[1] = cf[3] * s1 + cf[4] * s2;
                u[2] = cf[5] * s2;
            } else {
                u[0] = hh[0] * a0 + hh[1] * a1 + hh[2] * a2;
                u[1] = hh[1] * a0 + hh[3] * a1 + hh[4] * a2;
                u[2] = hh[2] * a0 + hh[4] * a1 + hh[5] * a2;
            }
            double* Pc = d.P_cand + 3 * (size_t)s.pt;
            double* yo = d.yp + 3 * (size_t)s.pt;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double dl = var ? -u[k] * spv[k] : 0.0;
                const double pn = Pv[k] + dl;
                Pc[k] = pn; yo[k] = u[k];
                const double df = pn - Pv[k];
                step2 += df * df;
            }
        } else dead_values(u);
        const int hl = seg_head_lane(s.head || !s.valid, lane);
        u[0] = __shfl(u[0], hl, kWave); u[1] = __shfl(u[1], hl, kWave); u[2] = __shfl(u[2], hl, kWave);
        if (s.valid) {
            const double m0 = v0 + E[0] * u[0] + E[1] * u[1] + E[2] * u[2];
            const double m1 = v1 + E[3] * u[0] + E[4] * u[1] + E[5] * u[2];
            model = m0 * (r0 - 0.5 * m0) + m1 * (r1 - 0.5 * m1);
        }
    } else {
        double wsum[3] = {0, 0, 0};
        int pt0 = -1;
        for (int tl = 0; tl < it.n_tiles; ++tl) {
            const int slot = (it.first_tile + tl) * kWave + lane;
            const int cam = d.slot_cam[slot];
            if (cam >= 0) {
                pt0 = d.slot_pt[slot];
                const double* y = d.px + 6 * (size_t)cam;
                double F[12], E[6];
                load_FE(d, slot, cam, pt0, F, E);
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) { v0 += F[k] * y[k]; v1 += F[6 + k] * y[k]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) wsum[k] += E[k] * v0 + E[3 + k] * v1;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { wsum[k] = wave_sum(wsum[k]); wsum[k] = __shfl(wsum[k], 0, kWave); }
        pt0 = __shfl(pt0, 0, kWave);
        const double* h = (PREP ? d.Hpp : d.Hinv) + 6 * (size_t)pt0;
        const double* g = d.gp + 3 * (size_t)pt0;
        const double a0 = g[0] - wsum[0], a1 = g[1] - wsum[1], a2 = g[2] - wsum[2];
        double u[3];
        if (PREP) {
            const double hv[6] = {h[0], h[1], h[2], h[3], h[4], h[5]};
            double cf[6];
            point_factor(hv, radius, cf);
            const double s0 = cf[0] * a0, s1 = cf[1] * a0 + cf[3] * a1, s2 = cf[2] * a0 + cf[4] * a1 + cf[5] * a2;
            u[0] = cf[0] * s0 + cf[1] * s1 + cf[2] * s2;
            u[1] = cf[3] * s1 + cf[4] * s2;
            u[2] = cf[5] * s2;
        } else {
            u[0] = h[0] * a0 + h[1] * a1 + h[2] * a2;
            u[1] = h[1] * a0 + h[3] * a1 + h[4] * a2;
            u[2] = h[2] * a0 + h[4] * a1 + h[5] * a2;
        }
        if (lane == 0) {
            const double* sp = d.scale_p + 3 * (size_t)pt0;
            const double* P = d.P + 3 * (size_t)pt0;
            double* Pc = d.P_cand + 3 * (size_t)pt0;
            double* yo = d.yp + 3 * (size_t)pt0;
            const bool var = !d.pt_const[pt0];
            for (int k = 0; k < 3; ++k) {
                const double dl = var ? -u[k] * sp[k] : 0.0;
                const double pn = P[k] + dl;
                Pc[k] = pn; yo[k] = u[k];
                const double df = pn - P[k];
                step2 += df * df;
            }
        }
        for (int tl = 0; tl < it.n_tiles; ++tl) {
            const int slot = (it.first_tile + tl) * kWave + lane;
            const int cam = d.slot_cam[slot];
            if (cam >= 0) {
                const double* y = d.px + 6 * (size_t)cam;
                double F[12], E[6];
                load_FE(d, slot, cam, pt0, F, E);
                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                for (int k = 0; k < 6; ++k) { v0 += F[k] * y[k]; v1 += F[6 + k] * y[k]; }
                const double m0 = v0 + E[0] * u[0] + E[1] * u[1] + E[2] * u[2];
                const double m1 = v1 + E[3] * u[0] + E[4] * u[1] + E[5] * u[2];
                model += m0 * (d.rt[slot] - 0.5 * m0) + m1 * (d.rt[ns + slot] - 0.5 * m1);
            }
        }
    }
    model = wave_sum(model);
    step2 = wave_sum(step2);
    if (lane == 0) { d.part[2 * d.n_items + item] = model; d.part[3 * d.n_items + item] = step2; }
}

__global__ void k_cam_update(Dev d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.n_cams) return;
    cam_update_one(d, c, d.px + 6 * (size_t)c, nullptr);
}

// Ceres' gradient max-norm |x - Plus(x, -g)|_inf with the unscaled gradient.
// rank_max (multi-rank only): the ranks' point-gradient maxima, folded to *out_pts here.
__device__ __forceinline__ double cam_gradmax_one(const Dev& d, int c, const double* g) {      // g: the camera's 6 gradient entries
    double m = 0.0;
    const unsigned cc = d.cam_const[c];
    const bool active = d.cam_act[c] > 0.0;
    const double* sc = d.scale_c + 6 * (size_t)c;
    if (active && !(cc & 1u)) {
        const CamRec& cur = d.cam[c];
        const double q[4] = {cur.q[0], cur.q[1], cur.q[2], cur.q[3]};
        const double dl[3] = {-g[0] / sc[0], -g[1] / sc[1], -g[2] / sc[2]};
        double qn[4];
        quat_plus(q, dl, qn);
        for (int k = 0; k < 4; ++k) m = fmax(m, fabs(q[k] - qn[k]));
    }
    if (active && !(cc & 2u))
        for (int k = 0; k < 3; ++k) m = fmax(m, fabs(g[3 + k] / sc[3 + k]));
    return m;
}
__global__ __launch_bounds__(kPcgThreads) void k_gradmax_cams(Dev d, double* __restrict__ out, const double* __restrict__ rank_max,
                                                           int n_ranks, double* __restrict__ out_pts) {     // one workgroup
    __shared__ double lds[kPcgThreads / kWave];
    double m = 0.0;
    for (int c = threadIdx.x; c < d.n_cams; c += kPcgThreads) m = fmax(m, cam_gradmax_one(d, c, d.camlin + 12 * (size_t)c + 6));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, kWave));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < kPcgThreads / kWave; ++i) r = fmax(r, lds[i]);
        *out = r;
        if (rank_max) {
            double m2 = 0.0;
            for (int i = 0; i < n_ranks; ++i) m2 = fmax(m2, rank_max[i]);
            *out_pts = m2;
        }
    }
}

// Up to 8 independent reductions in one launch (one workgroup each); op 0 = sum, 1 = max.
struct ReduceJobs { const double* in[8]; int n[8]; double* out[8]; int op[8]; };

// Deterministic single-workgroup reductions of partial arrays.
__global__ __launch_bounds__(kPcgThreads) void k_reduce_sum(const double* __restrict__ in, int n, double* out) {
    __shared__ double lds[kPcgThreads / kWave];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kPcgThreads) s += in[i];
    s = block_sum<kPcgThreads>(s, lds);
    if (threadIdx.x == 0) *out = s;
}

__global__ __launch_bounds__(kPcgThreads) void k_reduce_multi(ReduceJobs jobs) {
    __shared__ double lds[kPcgThreads / kWave];
    const double* in = jobs.in[blockIdx.x];
    const int n = jobs.n[blockIdx.x];
    if (jobs.op[blockIdx.x] == 0) {
        double s = 0.0;
        int i = threadIdx.x;
        for (; i + 7 * kPcgThreads < n; i += 8 * kPcgThreads) {     // 8 loads in flight, fixed order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[i + u * kPcgThreads];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; i < n; i += kPcgThreads) s += in[i];
        s = block_sum<kPcgThreads>(s, lds);
        if (threadIdx.x == 0) *jobs.out[blockIdx.x] = s;
    } else {
        double m = 0.0;
        int i = threadIdx.x;
        for (; i + 7 * kPcgThreads < n; i += 8 * kPcgThreads) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = in[i + u * kPcgThreads];
#pragma unroll
            for (int u = 0; u < 8; ++u) m = fmax(m, v[u]);
        }
        for (; i < n; i += kPcgThreads) m = fmax(m, in[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, kWave));
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            double r = 0.0;
            for (int i2 = 0; i2 < kPcgThreads / kWave; ++i2) r = fmax(r, lds[i2]);
            *jobs.out[blockIdx.x] = r;
        }
    }
}

__global__ __launch_bounds__(kPcgThreads) void k_reduce_max(const double* __restrict__ in, int n, double* out) {
    __shared__ double lds[kPcgThreads / kWave];
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += kPcgThreads) m = fmax(m, in[i]);
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

// ---------------------------------------------------------------- tail of a linearisation, one launch
// What used to be k_cam_segsum<12> -> k_reduce_multi -> k_gradmax_cams -> k_publish (four dependent launches of 4-10 us).
// At most one workgroup per CU (kTailGrid): a release fence or a ticket per CAMERA workgroup costs more than the launches
// it replaces (measured: 1000 workgroups, each with an L2 write-back and an arrival on one counter, 31 us).
//   * workgroup b adds the camera-major partials of the cameras b, b + G, ... (diag H_cc, g_c) in fixed order, and
//   * reduces ITS slice of every job's partial array to one value (two-level deterministic reduction: slice boundaries
//     depend on the sizes alone); both results leave the CU write-through (agent-scope stores), so no release fence;
//   * the workgroup that arrives last (every storing wave drains, one ticket per workgroup, ONE acquire by the last one — the
//     only inter-workgroup hand-off, and its consumer does nothing another workgroup waits for) adds the per-workgroup
//     values of every job in workgroup order, takes Ceres' gradient max-norm over the cameras and, on one rank, hands the
//     scalar block to the host through coherent memory (what k_publish did).
constexpr int kTailGrid = 1024;
struct TailArgs {
    ReduceJobs jobs; int njobs;
    double* part2;             // [8][gridDim.x] per-workgroup values
    unsigned* ticket;          // 9 counters, 128 bytes apart (8 shards + top), zero between launches (the last workgroup resets them)
    int gradmax;               // 1: camera gradient max-norm -> scal[S_GRADMAX_CAMS] (one rank; with several ranks it follows the all-reduce)
    double* host; unsigned long long seq;   // != nullptr: publish the scalar block
};

constexpr int kTailJobs = 8;       // slots of the second stage: up to 7 reduction jobs + the camera gradient max-norm

__global__ __launch_bounds__(kBlock) void k_lin_tail(Dev d, TailArgs a) {
    __shared__ double sums[12];
    __shared__ double sc[S_COUNT];
    __shared__ bool last;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, G = gridDim.x, b = blockIdx.x;
    // ---- stage 1: this workgroup's cameras and its slice of every job
    double gm = 0.0;
    for (int c = b; c < d.n_cams; c += G) {           // (uniform per workgroup: segsum_body has barriers)
        __syncthreads();
        segsum_body<12, true>(d.scat, d.cam_ptr_g, d.camlin, c, sums);
        __syncthreads();
        if (t == 0 && a.gradmax) gm = fmax(gm, cam_gradmax_one(d, c, sums + 6));
    }
    if (t == 0 && a.gradmax) __hip_atomic_store(a.part2 + (size_t)(kTailJobs - 1) * G + b, gm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int j = wave; j < a.njobs; j += kBlock / kWave) {        // one wave per job: no workgroup barriers
        const double* in = a.jobs.in[j];
        const int n = a.jobs.n[j], op = a.jobs.op[j];
        const int len = (n + G - 1) / G, lo = min(n, b * len), hi = min(n, lo + len);
        double v = 0.0;
        for (int i = lo + lane; i < hi; i += kWave) v = (op == 0) ? v + in[i] : fmax(v, in[i]);
        if (op == 0) v = wave_sum(v);
        else {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, kWave));
        }
        if (lane == 0) __hip_atomic_store(a.part2 + (size_t)j * G + b, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // hand-off to the last arriver (MI355X guide, inter-workgroup visibility, form R1: write-through payload, every storing
    // wave drains its stores, then one relaxed agent-scope arrival per workgroup)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        // arrivals are sharded over 8 counters (one word takes ~88 returning atomics per microsecond: 1000 arrivals on one
        // word cost 11 us); the last arrival of a shard arrives on the top counter
        const int sh = b & 7, nsh = min(G, 8), n_in_shard = (G - sh + 7) >> 3;
        bool l = false;
        if (__hip_atomic_fetch_add(a.ticket + 32 * sh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)n_in_shard)
            l = (__hip_atomic_fetch_add(a.ticket + 32 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)nsh);
        last = l;
    }
    __syncthreads();
    if (!last) return;
    // ---- stage 2 (one workgroup): 32 lanes per slot add the G per-workgroup values in a fixed order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (t < S_COUNT) sc[t] = __hip_atomic_load(d.scal + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    {
        const int j = t >> 5, l32 = t & 31;
        const bool isgm = (j == kTailJobs - 1);
        const bool on = isgm ? (a.gradmax != 0) : (j < a.njobs);
        const int op = isgm ? 1 : (on ? a.jobs.op[j] : 0);
        const double* in = a.part2 + (size_t)j * G;
        double v = 0.0;
        // (round 6) eight loads in flight per lane, added in index order as before: the loop used to wait for every load in turn — up to 32
        // dependent L2 round trips on the one workgroup the host is waiting for
        if (on)
            for (int i0 = l32; i0 < G; i0 += 32 * 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = i0 + 32 * u; x[u] = (i < G) ? in[i] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 32 * u < G) v = (op == 0) ? v + x[u] : fmax(v, x[u]);
            }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            const double ov = __shfl_down(v, off, 32);
            v = (op == 0) ? v + ov : fmax(v, ov);
        }
        if (on && l32 == 0) {
            double* out = isgm ? d.scal + S_GRADMAX_CAMS : a.jobs.out[j];
            *out = v;
            const long long idx = out - d.scal;
            if (idx >= 0 && idx < S_COUNT) sc[idx] = v;
        }
    }
    __syncthreads();
    if (t < 9) __hip_atomic_store(a.ticket + 32 * t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.host) {
        if (t < S_COUNT) a.host[t] = sc[t];
        __threadfence_system();
        __syncthreads();
        if (t == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.host + S_COUNT), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace xba
