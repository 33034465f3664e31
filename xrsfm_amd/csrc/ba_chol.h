// Explicit reduced camera matrix S and its Cholesky solve (the exact solve Ceres'
// SPARSE_SCHUR performs, /root/reference/src/optimization/ba_solver.cc:74).
//
//  * Off-diagonal 6x6 blocks  S(b,a) = - sum_tracks W_b Hinv W_a^T  are produced per
//    track pair by wave shuffles, written to a block-major scatter buffer and summed
//    in fixed order (bit-reproducible, no FP atomics).
//  * S is expanded into dense row-major storage in 64x64 tiles; only tiles that are
//    structurally non-zero after symbolic factorisation are touched, so a banded
//    (sequential) problem costs O(N_c) and an unordered one degrades to dense.
//  * Tile kernels: potrf (one workgroup, LDS), trsm and the rank-64 trailing update
//    as 64x64x64 tile products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64),
//    forward / backward substitution one panel per launch.
#pragma once
#include "ba_kernels.h"
#include "ba_pack.h"

namespace xba {

constexpr int kNB = 64;          // tile size (ba_plan.h: kPlanTile)
constexpr int kLdT = 66;         // LDS row stride (doubles): conflict-free ds_read_b64 for MFMA operands
constexpr int kCamsPerTileDev = 10;   // cameras per 64-row tile (ba_plan.h: kCamsPerTile)

struct CholDev {
    int n, n_pad, T;
    double* S;        // tile storage of the lower triangle: dense [n_pad][n_pad] row-major (tmap == nullptr, ld = n_pad), or PACKED:
                      // only the structurally non-zero tiles, 64 x 64 row-major each (ld = 64), tile (i,k) at S + tmap[i*T+k] * tstride
                      // (tstride = 4096 + 64 doubles: consecutive tiles start on different channels); tiles outside the pattern map to
                      // one shared all-zero tile, like the zero regions of the dense form
    const int* tmap; size_t ld; size_t tstride;
    double* Linv;     // [T][64][64] inverse of the diagonal tiles of L
    double* y;        // [n_pad] forward-substituted rhs
    double* rhs;      // [n_pad] working copy of b
    double* x;        // [n_pad]
    const int* cam_off;   // [n_cams] first scalar row of the camera's 6x6 diagonal block (tile-aligned groups)
    const int* tile_rows; // [T] leading rows of the tile that hold cameras (the rest is identity padding)
    int cw, cpt;          // unknowns per camera and cameras per tile: 6 / 10, or 9 / 7 in bal9 mode (ba_wide.h)
};

__device__ __forceinline__ double* tile_ptr(const CholDev& c, int i, int k) {
    return c.tmap ? c.S + (size_t)c.tmap[i * c.T + k] * c.tstride : c.S + (size_t)(i * kNB) * c.ld + k * kNB;
}

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------ S assembly
// Off-diagonal blocks  block(b,a) = sum_tracks W_b Hinv W_a^T  (camera b > camera a of the same track), plus — fused,
// because they need the same loads and the same W — the per-observation diagonal-block and reduced-rhs terms (the 28
// values k_schur_prep produces on the PCG path), written to the camera-major scatter buffer.
// With Hinv = C C^T (k_point_prep) every term is a product of V = W C with itself:  W_b Hinv W_a^T = V_b V_a^T.
// One wavefront per tile (64-thread workgroups: the staged operand of a wave lives in its own LDS; the kernel is
// latency-bound, so LDS and VGPR footprints are sized for 4 waves per SIMD).
//  * Gram tile (T tracks that together see C <= 10 distinct cameras; ba_pack.h): the lanes stage V as a [6C x 3T] matrix in
//    LDS (zero where a track does not see a camera) and the wave forms the Gram matrix G = V V^T with
//    v_mfma_f64_16x16x4_f64: G holds every camera-pair block already summed over the T tracks, written once per tile
//    (a regular tile, every track with the same L cameras, is the dense special case);
//  * other tiles (more distinct cameras): lane = observation a, partner b = a + d in the same track via shfl_down, one
//    block per pair;
//  * long tracks: lane loops over all later observations of its track.
constexpr int kRedLd = 15;      // row stride (doubles) of the LDS reduction buffer of k_schur_pairs: 64 x 15 x 8 B

// Destination table of a Gram tile in LDS: kGramTabLd x kGramTabLd ints, entry [ra][rb] = block id of the camera pair (ra < rb,
// both < C, the pair occurs in the tile), -1 otherwise.  The fixed row stride makes an accumulator element's table offset a
// function of the lane and of compile-time indices only, and folds "row < R && col < R && rb > ra" into the one test dst >= 0.
// (A host-built "store plan" — one int per accumulator element — was measured in round 3 and removed again: 107.6-111.4 us per
//  launch at config L against 103.6 for index arithmetic; its loads are one more dependent round trip at the end of every tile.)
constexpr int kGramTabLd = kGramMaxCams + 1;      // rows 60..63 of the last operand tile map to "camera 10": always -1
static_assert(kTileRunLd == kGramMaxCams, "one run record per distinct camera of a Gram tile (ba_kernels.h: k_gram_runs)");

// UPPER = false: hc = lower factor {c00 c10 c20 c11 c21 c22} stored by k_point_prep;  UPPER = true: hc = upper factor
// {c00 c01 c02 c11 c12 c22} of point_factor() (formed in the kernel).  Either way hc hc^T = Hinv and V = W hc.
template <bool UPPER>
__device__ __forceinline__ void pairs_V(const double* F, const double* E, const double* __restrict__ hc, double* V) {
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const double w0 = F[a] * E[0] + F[6 + a] * E[3], w1 = F[a] * E[1] + F[6 + a] * E[4], w2 = F[a] * E[2] + F[6 + a] * E[5];
        if (UPPER) {
            V[3 * a + 0] = w0 * hc[0];
            V[3 * a + 1] = w0 * hc[1] + w1 * hc[3];
            V[3 * a + 2] = w0 * hc[2] + w1 * hc[4] + w2 * hc[5];
        } else {
            V[3 * a + 0] = w0 * hc[0] + w1 * hc[1] + w2 * hc[2];
            V[3 * a + 1] = w1 * hc[3] + w2 * hc[4];
            V[3 * a + 2] = w2 * hc[5];
        }
    }
}

// diagonal block of S and reduced rhs of one observation:  F^T F - V V^T (upper triangle, 21) | -V C^T g (6) | 0
template <bool UPPER>
__device__ __forceinline__ void pairs_diag(const double* F, const double* V, const double* __restrict__ hc,
                                           const double* __restrict__ g, double* o28) {
    int idx = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c2 = a; c2 < 6; ++c2)
            o28[idx++] = F[a] * F[c2] + F[6 + a] * F[6 + c2]
                         - (V[3 * a] * V[3 * c2] + V[3 * a + 1] * V[3 * c2 + 1] + V[3 * a + 2] * V[3 * c2 + 2]);
    const double g0 = g[0], g1 = g[1], g2 = g[2];
    // u = hc^T g
    const double u0 = UPPER ? hc[0] * g0 : hc[0] * g0 + hc[1] * g1 + hc[2] * g2;
    const double u1 = UPPER ? hc[1] * g0 + hc[3] * g1 : hc[3] * g1 + hc[4] * g2;
    const double u2 = UPPER ? hc[2] * g0 + hc[4] * g1 + hc[5] * g2 : hc[5] * g2;
#pragma unroll
    for (int a = 0; a < 6; ++a) o28[21 + a] = -(V[3 * a] * u0 + V[3 * a + 1] * u1 + V[3 * a + 2] * u2);
    o28[27] = 0.0;
}

// G = V V^T for a Gram tile whose operand has NI 16-row tiles (row stride Cp doubles).  The tracks are staged in `passes`
// rounds of Th tracks (two rounds when one would not fit the small LDS class: the accumulators simply carry over); per
// K-step of 4 every lane reads ONE value per row tile — the same register is the A operand of the products in its tile row
// and the B operand of those in its tile column (identical layouts) — and all NI(NI+1)/2 accumulators advance.  Then the
// blocks (camera rb > camera ra) go to their destinations (dtab: [C][C], -1 = the pair never occurs in the tile).
// CW: operand rows per camera = width of the camera blocks (6; 9 in bal9 mode, ba_wide.h: k9_pairs_gram — up to 7 cameras = 63 rows).
template <int NI, int CW = 6>
__device__ __forceinline__ void gram_tile(double* __restrict__ Vst, int R, int Cp, int C, const int* __restrict__ dtab,
                                          double* __restrict__ scat2, int lane, const double (&V)[3 * CW], bool valid, int t, int cidx,
                                          int T, int Th, int passes, bool dense) {
    static_assert(NI >= 1 && NI <= 4, "a Gram tile has at most 10 cameras = 60 operand rows (ba_pack.h: kGramMaxCams)");
    const int li = lane & 15, lk = lane >> 4;
    v4d acc[NI * (NI + 1) / 2];
#pragma unroll
    for (int p = 0; p < NI * (NI + 1) / 2; ++p) acc[p] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* rowp[NI];
#pragma unroll
    for (int I = 0; I < NI; ++I) rowp[I] = Vst + min(16 * I + li, R - 1) * Cp + lk;
    for (int h = 0; h < passes; ++h) {
        const int t0 = h * Th, tn = min(Th, T - t0);           // tracks t0 .. t0+tn-1 in this round
        const int C4 = (3 * tn + 3) & ~3;
        if (dense && passes == 1) {
            // dense (regular) tile: only the K padding columns 3T..C4-1 the products read need zeros — none when 3T is a multiple
            // of 4 (16 tracks of 4 cameras) — (rows beyond R are never staged: reads are clamped, results discarded)
            const int padc = C4 - 3 * tn;
            for (int e = lane; e < R * padc; e += kWave) {
                const int row = e / padc, cc = 3 * tn + (e - row * padc);
                Vst[row * Cp + cc] = 0.0;
            }
        } else {
            // LDS operations of one wave complete in order; 16-byte stores (the operand is 16-byte aligned).  With kGramPad = 1 the row
            // stride Cp is ODD: an odd R * Cp leaves one last element, which is zeroed on its own (it is the pad column of the last row,
            // never read by the K loop today — zeroed anyway so that the layout can change without a silent gap; a 16-byte store there
            // would run into the destination table that follows the operand)
            for (int e = lane; e < (R * Cp) / 2; e += kWave) reinterpret_cast<double2*>(Vst)[e] = make_double2(0.0, 0.0);
            if (((R * Cp) & 1) && lane == kWave - 1) Vst[R * Cp - 1] = 0.0;
        }
        if (valid && t >= t0 && t < t0 + tn) {
#pragma unroll
            for (int i = 0; i < CW; ++i)
#pragma unroll
                for (int m = 0; m < 3; ++m) Vst[(CW * cidx + i) * Cp + 3 * (t - t0) + m] = V[3 * i + m];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the staged operand (and dtab) are in LDS.  Not vmcnt: the stores
        __builtin_amdgcn_wave_barrier();               // of the diagonal terms issued just before are still in flight and stay so
        XBA_STAMP(0, 6);
        for (int k0 = 0; k0 < C4; k0 += 4) {
            double a[NI];
#pragma unroll
            for (int I = 0; I < NI; ++I) a[I] = rowp[I][k0];
            int p = 0;
#pragma unroll
            for (int I = 0; I < NI; ++I)
#pragma unroll
                for (int J = 0; J <= I; ++J) { acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], a[J], acc[p], 0, 0, 0); ++p; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);            // the reads above are done before the next round overwrites the operand
        __builtin_amdgcn_wave_barrier();
    }
    XBA_STAMP(0, 7);
    // blocks (camera rb > camera ra) to their destinations: dtab[ra][rb] (fixed stride), -1 = nothing to store
    (void)C; (void)R;
    int tcol[NI], jcol[NI];
#pragma unroll
    for (int J = 0; J < NI; ++J) { const int col = 16 * J + li; const int ra = col / CW; tcol[J] = ra * kGramTabLd; jcol[J] = col - CW * ra; }
    int p = 0;
#pragma unroll
    for (int I = 0; I < NI; ++I) {
        int trow[4], irow[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const int row = 16 * I + lk + 4 * g; const int rb = row / CW; trow[g] = rb; irow[g] = CW * (row - CW * rb); }
#pragma unroll
        for (int J = 0; J <= I; ++J) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dst = dtab[tcol[J] + trow[g]];
                if (dst >= 0) scat2[(CW * CW) * (size_t)dst + irow[g] + jcol[J]] = acc[p][g];
            }
            ++p;
        }
    }
}

// ---- (round 5) the same Gram product on v_mfma_f64_4x4x4_4b_f64: four independent 4x4 result blocks per instruction.
// FP64 matrix instructions run on the SIMD's vector data path on gfx950 — they do not overlap with the vector instructions of the
// SIMD's other waves (tools/bench_pipes.hip: 2 vector + 2 matrix waves take exactly the sum of the two alone) and deliver the vector
// rate (16x16x4: 64 cycles, 4x4x4 x 4 blocks: 16.5) — so what a 16x16 result tile computes outside the camera-pair blocks is vector
// time lost.  A tile of 4 cameras (24 operand rows) needs 6 blocks of 36 elements: the 16-row form computes 3 x 256 elements per
// K-step (28 % wanted, 192 cycles), 4-row groups 17 blocks of 16 in 5 instructions (64 % wanted, 83 cycles).
// Schedule: the operand rows are cut into groups of 4; block (rg >= cg) is wanted when some row of group rg belongs to a later
// camera than some row of group cg.  e[C] lists the wanted blocks of a tile of C cameras, four per instruction (rg | cg << 8; padded
// with block (0,0): one camera, nothing of it is stored), n[C] = instructions.
// Measured (tools/runs/r05_call19.sh, _call21.sh; results bit-identical to the 16x16 form): config L (tiles of 4 cameras, 5 instructions)
// 99.6-101.2 -> 96.2 us per launch; tiles of 7-8 cameras (config R: 15-18 instructions against 6 of the 16-row form per K-step) and
// 9-wide blocks (config Lb9) gain nothing — the form is used for schedules of at most kGram4MaxInst instructions (6-wide: <= 4 cameras).
// Operand / result layout of the instruction (found by one-hot probing, tools/probe_mfma4.hip): lane l holds A_b[i][k] and B_b[k][j]
// with i = j = l & 3, b = (l >> 2) & 3, k = l >> 4, and receives D_b[l >> 4][l & 3].
constexpr int kGram4MaxInst = 6;
template <int CW>
struct GramSched4 {
    static constexpr int kMaxC = CW == 6 ? kGramMaxCams : kGramMaxCamsWide;
    static constexpr int kMaxG = (CW * kMaxC + 3) / 4;
    static constexpr int kMaxE = ((kMaxG * (kMaxG + 1) / 2 + 3) / 4) * 4;
    unsigned short e[kMaxC + 1][kMaxE];
    int n[kMaxC + 1];
};
template <int CW>
constexpr GramSched4<CW> make_gram_sched4() {
    GramSched4<CW> s{};
    for (int C = 1; C <= GramSched4<CW>::kMaxC; ++C) {
        const int R = CW * C, G = (R + 3) / 4;
        int cnt = 0;
        for (int rg = 0; rg < G; ++rg)
            for (int cg = 0; cg <= rg; ++cg) {
                const int last_row = 4 * rg + 3 < R - 1 ? 4 * rg + 3 : R - 1;
                if (last_row / CW > (4 * cg) / CW) s.e[C][cnt++] = (unsigned short)(rg | (cg << 8));
            }
        while (cnt & 3) s.e[C][cnt++] = 0;
        s.n[C] = cnt / 4;
    }
    return s;
}
__device__ const GramSched4<6> g_gram_sched6 = make_gram_sched4<6>();
template <int CW> __device__ __forceinline__ const GramSched4<CW>& gram_sched4();
template <> __device__ __forceinline__ const GramSched4<6>& gram_sched4<6>() { return g_gram_sched6; }

// The lane's entry of the schedule of a tile of C cameras (n_inst = 0: the tile takes the 16x16 form), requested at kernel start and parked in LDS behind
// the destination table once that is written: a lookup in global memory where the products start would be a memory round trip in
// the middle of every tile (measured: +3 000 cycles per wave).
template <int CW>
__device__ __forceinline__ int gram4_sched_load(int C, int lane, int& n_inst) {
    const GramSched4<CW>& S = gram_sched4<CW>();
    n_inst = S.n[C];
    if (n_inst > kGram4MaxInst) { n_inst = 0; return 0; }          // (uniform) larger tiles: 16x16 result tiles
    const int ne = 4 * n_inst;
    return lane < ne ? (int)S.e[C][lane] : 0;        // (kGram4MaxInst x 4 entries: one per lane is enough)
}
static_assert(4 * kGram4MaxInst <= kWave, "one schedule entry per lane");
__device__ __forceinline__ void gram4_sched_store(unsigned short* sched, int se, int lane, int n_inst) {
    if (lane < 4 * n_inst) sched[lane] = (unsigned short)se;
}

// row / CW for row < 64 without an integer division (v_mul_hi is a quarter-rate instruction)
template <int CW> __device__ __forceinline__ unsigned gram_cam_of_row(unsigned row) {
    static_assert(CW == 6 || CW == 9, "6-wide or bal9 camera blocks");
    return CW == 6 ? (row * 43) >> 8 : (row * 57) >> 9;
}

// NB instructions of the schedule (entries ent[0], ent[4], ...: the lane's block of each, in LDS): K loop, then the wanted elements to their
// destinations — element (row, col) of the tile, camera rb = row / CW > camera ra = col / CW, goes to block dtab[ra][rb] (-1: nothing).
template <int CW, int NB>
__device__ __forceinline__ void gram4_batch(const double* __restrict__ Vst, int R, int Cp, int C4, const unsigned short* __restrict__ ent,
                                            const int* __restrict__ dtab, double* __restrict__ scat2, int lane) {
    const int i4 = lane & 3, kk = lane >> 4;
    int e[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) e[u] = ent[4 * u];
    typedef const __attribute__((address_space(3))) char* lds_ptr;       // (explicit: as generic pointers the induction variables below become flat loads)
    typedef const __attribute__((address_space(3))) double* lds_dptr;
    lds_ptr pa[NB];                                     // the lane's element of the two operands of each instruction, advancing with K
    lds_ptr pb[NB];
    double acc[NB];
    lds_ptr Vb = (lds_ptr)Vst + 8 * kk;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int rg = e[u] & 255, cg = e[u] >> 8;
        pa[u] = Vb + 8 * (min(4 * rg + i4, R - 1) * Cp);      // (rows past the operand: clamped, their results are never stored)
        pb[u] = Vb + 8 * (min(4 * cg + i4, R - 1) * Cp);
        acc[u] = 0.0;
    }
    int k0 = 0;
    for (; k0 + 16 <= C4; k0 += 16) {                   // four K-steps per round: one address update per operand, the rest immediate offsets
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < NB; ++u)
                acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(*(lds_dptr)(pa[u] + 32 * j), *(lds_dptr)(pb[u] + 32 * j), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NB; ++u) { pa[u] += 128; pb[u] += 128; }
    }
    for (; k0 < C4; k0 += 4) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(*(lds_dptr)pa[u], *(lds_dptr)pb[u], acc[u], 0, 0, 0);
            pa[u] += 32; pb[u] += 32;
        }
    }
    // (the destinations of all NB values are requested before the first store: one LDS round trip, not NB in a row; unsigned
    //  arithmetic: one 64-bit multiply-add and one shift-add form an address — with ints the compiler sign-extends every term)
    int dst[NB];
    unsigned off[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const unsigned row = 4u * (unsigned)(e[u] & 255) + (unsigned)kk, col = 4u * (unsigned)(e[u] >> 8) + (unsigned)i4;
        const unsigned rb = gram_cam_of_row<CW>(row), ra = gram_cam_of_row<CW>(col);
        dst[u] = dtab[ra * kGramTabLd + rb];
        off[u] = CW * (row - CW * rb) + (col - CW * ra);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u)
        if (dst[u] >= 0) scat2[(unsigned long long)(unsigned)dst[u] * (unsigned)(CW * CW) + off[u]] = acc[u];
}

// gram_tile for a tile staged in ONE round (passes == 1: every tile of the small LDS class that fits it, ba_pack.h: gram_lds_need)
template <int CW = 6>
__device__ __forceinline__ void gram_tile4(double* __restrict__ Vst, int R, int Cp, int C, const int* __restrict__ dtab,
                                           double* __restrict__ scat2, int lane, const double (&V)[3 * CW], bool valid, int t, int cidx,
                                           int T, bool dense, const unsigned short* __restrict__ sched, int n_inst) {
    const unsigned short* ent = sched + ((lane >> 2) & 3);      // (LDS copy of the tile's schedule: gram4_sched_load / _store)
    const int C4 = (3 * T + 3) & ~3;
    if (dense) {
        const int padc = C4 - 3 * T;
        for (int e = lane; e < R * padc; e += kWave) {
            const int row = e / padc, cc = 3 * T + (e - row * padc);
            Vst[row * Cp + cc] = 0.0;
        }
    } else {
        for (int e = lane; e < (R * Cp) / 2; e += kWave) reinterpret_cast<double2*>(Vst)[e] = make_double2(0.0, 0.0);
        if (((R * Cp) & 1) && lane == kWave - 1) Vst[R * Cp - 1] = 0.0;        // (odd row stride: see gram_tile)
    }
    if (valid) {
#pragma unroll
        for (int i = 0; i < CW; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) Vst[(CW * cidx + i) * Cp + 3 * t + m] = V[3 * i + m];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the staged operand (and dtab) are in LDS
    __builtin_amdgcn_wave_barrier();
    XBA_STAMP(0, 6);
    switch (n_inst) {                              // (uniform; the caller takes this path for schedules of at most kGram4MaxInst instructions)
        case 6: gram4_batch<CW, 6>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
        case 5: gram4_batch<CW, 5>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
        case 4: gram4_batch<CW, 4>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
        case 3: gram4_batch<CW, 3>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
        case 2: gram4_batch<CW, 2>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
        default: gram4_batch<CW, 1>(Vst, R, Cp, C4, ent, dtab, scat2, lane); break;
    }
    XBA_STAMP(0, 7);
}

// GRAM = true: the item list holds Gram tiles only (the common case, compiled without the other paths so that their register
// needs do not shape its allocation: 118 VGPRs, no spills); GRAM = false: per-pair tiles and long tracks.  The two
// instantiations write disjoint outputs and run concurrently on two streams.
// PREP = true: the damped point block's factor is formed here from Hpp and the radius (point_factor(): no k_point_prep launch,
// no Hinv / Hc arrays); false: read from d.Hc (round-2 schedule, XRSFM_BA_PREP_FUSED=0).
// NIK: GRAM = true — the operand height of the tiles of this launch in 16-row MFMA tiles (1..4; ba_plan.h sorts the Gram tiles
// into one launch per height, so the 80 accumulator registers of a 10-camera tile exist only in the instantiation that
// needs them; a single instantiation with a switch over 1..4 spilled 21-26 VGPRs in its common path, and one such build produced
// wrong blocks at scale — DESIGN.md section 5); GRAM = false — 0.
// (round 6) NIK = 0 with GRAM = true: the heights 1..3 in ONE launch, the height taken from the tile's camera count — the 24
// accumulator registers of a 3-tile operand fit the 128-register budget of 4 waves per SIMD, which the common path has anyway, so
// only the 10-camera tiles (NIK = 4: 40 accumulators, 3 waves per SIMD) keep a launch of their own.  A ragged map had four to six
// launches per pass, most of them too small to fill the chip and run side by side on two streams; now one (tiles in descending
// height: the long ones start first).
template <bool GRAM, bool PREP, int NIK>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu((GRAM && NIK < 4) ? 4 : (GRAM ? 3 : 2), (GRAM && NIK < 4) ? 4 : 3)))      // no instantiation may spill (tests/test_capi_cpu.py)
void k_schur_pairs(Dev d, const int* __restrict__ item_list, const int* __restrict__ slot_pair_ptr, const int* __restrict__ pair_dst,
                   int n_obs_pairs, double* __restrict__ scat2, double radius, double* __restrict__ pair_v = nullptr, int gram4 = 1) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    XBA_STAMP(0, 0);
    // one launch per LDS class (ba_plan.h); the Gram classes list tiles, the other class items
    const int entry = item_list[(GRAM && NIK == 0) ? gridDim.x - 1 - blockIdx.x : blockIdx.x];      // (merged launch: the list ascends in height)
    Item it;
    if (GRAM) { it.first_tile = entry; it.n_tiles = 1; }
    else it = d.items[entry];
    if (GRAM || it.n_tiles == 1) {
        // (Gram tile) the destination table of its camera pairs and the lane's entry in the camera-major scatter buffer are
        // requested before anything else: two dependent loads whose latency then hides behind the operand loads and the
        // diagonal terms instead of sitting in front of the Gram stage
        int dt0 = -1, dt1 = -1;         // entries lane and lane + 64 of the kGramTabLd x kGramTabLd table
        int g4_se = 0, g4_n = 0;        // ... and of the tile's schedule of 4x4 result blocks (gram_tile4)
        if (GRAM) {
            const int C0 = d.tile_ncam[it.first_tile];
            const int* src = pair_dst + n_obs_pairs + d.tile_gt_off[it.first_tile];
            const int a0 = lane / kGramTabLd, b0 = lane - kGramTabLd * a0;
            const int a1 = (lane + kWave) / kGramTabLd, b1 = lane + kWave - kGramTabLd * a1;
            if (b0 > a0 && b0 < C0) dt0 = src[a0 * C0 + b0];
            if (b1 > a1 && b1 < C0 && a1 < kGramTabLd) dt1 = src[a1 * C0 + b1];
            if (NIK <= 2 && gram4) g4_se = gram4_sched_load<6>(C0, lane, g4_n);       // (NIK = 0: g4_n = 0 for the tiles that take the 16x16 form)
        }
        const SlotCtx s = load_slot(d, it.first_tile, lane);
        const int cp = d.slot_campos_g[s.slot];
        const int cidx_raw = GRAM ? (int)d.slot_cidx[s.slot] : 0;
        const int L = d.tile_stride[it.first_tile];
        int g_pos = kWave - 1, g_pk0 = 0, g_pk1 = 0, g_pk2 = 0;      // ragged Gram tile: the per-camera sums' tables (used far below: the loads ride with the slot record)
        if (GRAM && L == 0) {
            const int C0 = d.tile_ncam[it.first_tile];
            const int* trun = d.tile_run + (size_t)it.first_tile * kTileRunLd;
            const int c0 = lane / 14, c1 = (lane + 64) / 14, c2 = (lane + 128) / 14;
            g_pos = (int)d.slot_gpos[s.slot];
            if (c0 < C0) g_pk0 = trun[c0];
            if (c1 < C0) g_pk1 = trun[c1];
            if (c2 < C0) g_pk2 = trun[c2];
        }
        double V[18];                 // (lanes without an observation: never staged, never a pair partner — XBA_POISON checks it)
        int npair = 0, pbase = 0;
        {
            double o28[28];
            XBA_STAMP(0, 1);
            if (s.valid) {
                // the point's factor of Hinv and gradient are requested with the Jacobian records (left where they are used, the
                // compiler issues them after the first batch of loads has returned: one more memory round trip per tile)
                double hcv[6], gv[3];
                {
                    const double* hc = (PREP ? d.Hpp : d.Hc) + 6 * (size_t)s.pt;
                    const double* g = d.gp + 3 * (size_t)s.pt;
#pragma unroll
                    for (int k = 0; k < 6; ++k) hcv[k] = hc[k];
                    gv[0] = g[0]; gv[1] = g[1]; gv[2] = g[2];
                }
                double F[12], E[6];
                load_FE(d, s.slot, s.cam, s.pt, F, E);
                if (PREP) { double hf[6]; point_factor(hcv, radius, hf);
#pragma unroll
                    for (int k = 0; k < 6; ++k) hcv[k] = hf[k]; }
                XBA_STAMP(0, 2);
                pairs_V<PREP>(F, E, hcv, V);
                XBA_STAMP(0, 3);
                pairs_diag<PREP>(F, V, hcv, gv, o28);
            } else if (PREP) { dead_values(V); dead_values(o28); }
            else {                 // (the round-2 schedule, a test oracle: plain initialisers — the definitions without an instruction cost it two spilled registers)
#pragma unroll
                for (int k = 0; k < 18; ++k) V[k] = XBA_DEAD;
#pragma unroll
                for (int k = 0; k < 28; ++k) o28[k] = XBA_DEAD;
            }
            XBA_STAMP(0, 4);
            const int Cg = GRAM ? d.tile_ncam[it.first_tile] : 0;
            if (L > 0) {
                // Regular tile: sum the 28 values over the tracks through the wave's LDS (the region the operand V is staged
                // in afterwards): every lane deposits its values, then one lane per (camera, value) adds the T entries in
                // track order.  ~100 instructions instead of the ~340 of a shuffle tree; two halves of 14 values to fit.
                const int T = __popcll(__ballot(s.valid)) / L;
                double* red = smem;                                         // [64][kRedLd]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int k = 0; k < 14; ++k) red[lane * kRedLd + k] = o28[14 * h + k];
                    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0)
                    __builtin_amdgcn_wave_barrier();
                    // tracks of 2, 3 or 4 observations in a full tile (what most of a map consists of): one round, compile-time
                    // strides — LDS reads with immediate offsets, no loop control (−1 VALU and −3 SALU per term)
                    if ((L == 4 && T == 16) || (L == 2 && T == 32) || (L == 3 && T == 21)) {
                        const bool on = lane < 14 * L;
                        const int r = on ? lane / 14 : 0, k = lane - 14 * r;
                        const int cpr = __shfl(cp, r, kWave);
                        if (on) {
                            const double* src = red + r * kRedLd + k;
                            double sum = 0.0;
                            if (L == 4) {
#pragma unroll
                                for (int t = 0; t < 16; ++t) sum += src[t * 4 * kRedLd];
                            } else if (L == 2) {
#pragma unroll
                                for (int t = 0; t < 32; ++t) sum += src[t * 2 * kRedLd];
                            } else {
#pragma unroll
                                for (int t = 0; t < 21; ++t) sum += src[t * 3 * kRedLd];
                            }
                            d.scat[28 * (size_t)cpr + 14 * h + k] = sum;
                        }
                    } else
                    for (int q0 = 0; q0 < 14 * L; q0 += kWave) {                 // uniform trip count: the shuffle below reads lanes
                        const int q = q0 + lane;                                 // that a per-lane loop bound would already have retired
                        const bool on = q < 14 * L;                              // (L = 14, 19, ...: the last round's source lane)
                        const int r = on ? q / 14 : 0, k = q - 14 * r;
                        const int cpr = __shfl(cp, r, kWave);                    // lanes < L are the writers of the tile
                        if (on) {
                            const double* src = red + r * kRedLd + k;
                            double sum = 0.0;
                            for (int t = 0; t < T; ++t) sum += src[t * L * kRedLd];
                            d.scat[28 * (size_t)cpr + 14 * h + k] = sum;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (Cg > 0) {
                // Gram tile with ragged tracks: the same through LDS, per DISTINCT camera of the tile: the lane that owns
                // (camera c, value k) adds the entries of the lanes whose observation is in camera c, in lane order; the
                // first of them holds the camera's entry in the scatter buffer.
                // (round 4) The lanes deposit their values SORTED BY CAMERA — position = (lanes of earlier cameras) + (earlier lanes
                // of the same camera), from the ballot masks — so that the lane of (camera c, value k) walks a contiguous run of
                // count_c entries with a counted loop (independent LDS reads) instead of peeling a lane mask bit by bit (a dependent
                // ffs / read / clear chain per term: 26 % of the wave's life on config R).  Same terms in the same (lane) order.
                double* red = smem;
                const int nq = 14 * Cg;
                // (round 6) deposit position and runs from the per-context tables (k_gram_runs, ba_kernels.h), requested at the head of the
                // kernel — until round 5 a ballot loop over the tile's cameras: ~110 vector and ~80 scalar instructions of a ragged tile's 1 240 / 440
                const int mypos = g_pos;                                    // (lanes without an observation park their zeros in the last row, 63)
                const int pk0 = g_pk0, pk1 = g_pk1, pk2 = g_pk2;           // run start | length << 8 | first lane << 16 of the camera of q = lane, + 64, + 128
                // (run < 64 whenever a lane has no observation, so row 63 — where those lanes park their dead values — is in no camera's run)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int k = 0; k < 14; ++k) red[mypos * kRedLd + k] = o28[14 * h + k];
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int rd = 0; rd < 3; ++rd) {
                        const int q = lane + 64 * rd;
                        const int pk = rd == 0 ? pk0 : (rd == 1 ? pk1 : pk2);
                        const int st = pk & 255, n = (pk >> 8) & 255, first = pk >> 16;
                        const bool on = q < nq;
                        const int k = q % 14;
                        const int cpr = __shfl(cp, on ? first : 0, kWave);
                        if (on) {
                            const double* src = red + st * kRedLd + k;
                            double sum = 0.0;
                            int j = 0;
                            for (; j + 4 <= n; j += 4) {
                                const double a0 = src[j * kRedLd], a1 = src[(j + 1) * kRedLd], a2 = src[(j + 2) * kRedLd], a3 = src[(j + 3) * kRedLd];
                                sum += a0; sum += a1; sum += a2; sum += a3;
                            }
                            for (; j < n; ++j) sum += src[j * kRedLd];
                            d.scat[28 * (size_t)cpr + 14 * h + k] = sum;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            } else if (cp >= 0) {
                double2* out = reinterpret_cast<double2*>(d.scat + 28 * (size_t)cp);
#pragma unroll
                for (int k = 0; k < 14; ++k) out[k] = make_double2(o28[2 * k], o28[2 * k + 1]);
            }
        }
        XBA_STAMP(0, 5);
        const int C = d.tile_ncam[it.first_tile];
        if (GRAM) {
            // Gram tile: rows = (distinct camera of the tile, 6), columns = (track, 3); cells of cameras a track does not see
            // stay zero.  G = V V^T then holds every camera-pair block of the tile summed over its tracks.
            const unsigned long long headmask = __ballot(s.head);
            const int T = __popcll(headmask);
            const int nvalid = __popcll(__ballot(s.valid));
            const int t = __popcll(headmask & ((2ull << lane) - 1ull)) - 1;        // rank of the lane's track in the tile
            const int cidx = s.valid ? cidx_raw : 0;
            int passes = 1;
            (void)gram_lds_need(C, T, &passes);                             // one staging round, or two with half the tracks each
            const int Th = (T + passes - 1) / passes;
            const int R = 6 * C, Rp = (R + 15) & ~15, Cp = ((3 * Th + 3) & ~3) + kGramPad;
            double* Vst = smem;                                             // only the R rows that hold data are staged
            int* dtab = reinterpret_cast<int*>(smem + R * Cp);              // [kGramTabLd][kGramTabLd] destination of block (cb > ca) at [ca][cb], -1 none
            dtab[lane] = dt0;
            if (lane + kWave < kGramTabLd * kGramTabLd) dtab[lane + kWave] = dt1;
            const bool dense = nvalid == T * C;
            (void)Rp;
            // one staging round: 4x4 result blocks (gram_tile4); two rounds (the accumulators carry over): 16x16 tiles.  gram4 = 0: always
            // the latter (XRSFM_BA_GRAM4=0, the A/B oracle of tests/test_gpu_parity.py)
            unsigned short* sched = reinterpret_cast<unsigned short*>(dtab + kGramTabLd * kGramTabLd);
            if (NIK <= 2 && g4_n > 0 && passes == 1) {
                gram4_sched_store(sched, g4_se, lane, g4_n);
                gram_tile4<6>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, dense, sched, g4_n);
            }
            else if (NIK == 0) {
                if (R > 32) gram_tile<3>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, Th, passes, dense);
                else if (R > 16) gram_tile<2>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, Th, passes, dense);
                else gram_tile<1>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, Th, passes, dense);
            }
            else gram_tile<(NIK > 0 ? NIK : 1)>(Vst, R, Cp, C, dtab, scat2, lane, V, s.valid, t, cidx, T, Th, passes, dense);
            XBA_STAMP(0, 8);
            return;
        }
        if (GRAM) return;           // (not reached: keeps the per-pair code out of the Gram instantiation)
        if (pair_v) {
            // (round 4) collections with long tracks: the camera-pair blocks are not written per pair — 288 bytes each, tens of millions
            // of them, read back by the segmented sum — but formed where they are summed (ba_kernels.h: k_chol_segsum_v) from the
            // operands V, 144 bytes per observation, stored here
            if (s.valid) {
                double2* out = reinterpret_cast<double2*>(pair_v + 18 * (size_t)s.slot);
#pragma unroll
                for (int k = 0; k < 9; ++k) out[k] = make_double2(V[2 * k], V[2 * k + 1]);
            }
            return;
        }
        if (s.valid) {
            pbase = slot_pair_ptr[s.slot];
            npair = slot_pair_ptr[s.slot + 1] - pbase;
        }
        int maxp = npair;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxp = max(maxp, __shfl_xor(maxp, off, kWave));
        for (int dd = 1; dd <= maxp; ++dd) {
            double Vb[18];
#pragma unroll
            for (int k = 0; k < 18; ++k) Vb[k] = __shfl_down(V[k], dd, kWave);
            if (dd <= npair) {
                double2* out = reinterpret_cast<double2*>(scat2 + 36 * (size_t)pair_dst[pbase + dd - 1]);
#pragma unroll
                for (int rb = 0; rb < 6; ++rb) {
                    double o[6];
#pragma unroll
                    for (int ca = 0; ca < 6; ++ca)
                        o[ca] = Vb[3 * rb] * V[3 * ca] + Vb[3 * rb + 1] * V[3 * ca + 1] + Vb[3 * rb + 2] * V[3 * ca + 2];
                    out[3 * rb + 0] = make_double2(o[0], o[1]);
                    out[3 * rb + 1] = make_double2(o[2], o[3]);
                    out[3 * rb + 2] = make_double2(o[4], o[5]);
                }
            }
        }
        return;
    }
    if (GRAM) return;
    // long track: lane handles observation a, loops over all later observations b of the track
    const int s_begin = it.first_tile * kWave, s_end = s_begin + it.n_tiles * kWave;
    for (int sa = s_begin + lane; sa < s_end; sa += kWave) {
        if (d.slot_cam[sa] < 0) continue;
        const int pt = d.slot_pt[sa];
        double hc[6];
        {
            const double* hp = (PREP ? d.Hpp : d.Hc) + 6 * (size_t)pt;
            double hv[6];
            for (int k = 0; k < 6; ++k) hv[k] = hp[k];
            if (PREP) point_factor(hv, radius, hc);
            else for (int k = 0; k < 6; ++k) hc[k] = hv[k];
        }
        double Va[18];
        {
            double Fa[12], Ea[6], o28[28];
            load_FE(d, sa, d.slot_cam[sa], pt, Fa, Ea);
            pairs_V<PREP>(Fa, Ea, hc, Va);
            pairs_diag<PREP>(Fa, Va, hc, d.gp + 3 * (size_t)pt, o28);
            double* out = d.scat + 28 * (size_t)d.slot_campos_g[sa];
            for (int k = 0; k < 28; ++k) out[k] = o28[k];
        }
        if (pair_v) {
            double2* out = reinterpret_cast<double2*>(pair_v + 18 * (size_t)sa);
            for (int k = 0; k < 9; ++k) out[k] = make_double2(Va[2 * k], Va[2 * k + 1]);
            continue;
        }
        const int pbase = slot_pair_ptr[sa];
        const int npair = slot_pair_ptr[sa + 1] - pbase;
        for (int dd = 1; dd <= npair; ++dd) {
            const int sb = sa + dd;
            double Fb[12], Eb[6], Vb[18];
            load_FE(d, sb, d.slot_cam[sb], pt, Fb, Eb);
            pairs_V<PREP>(Fb, Eb, hc, Vb);
            double* out = scat2 + 36 * (size_t)pair_dst[pbase + dd - 1];
            for (int rb = 0; rb < 6; ++rb)
                for (int ca = 0; ca < 6; ++ca)
                    out[6 * rb + ca] = Vb[3 * rb] * Va[3 * ca] + Vb[3 * rb + 1] * Va[3 * ca + 1] + Vb[3 * rb + 2] * Va[3 * ca + 2];
        }
    }
}

// Dense fill, one workgroup per structurally non-zero tile (ti,tj): the tile is composed in LDS — zeros (fill-in), the
// off-diagonal blocks -Sblk[b] (block (rb > ca) goes to the lower triangle in the elimination order cam_off, transposed if
// rb is ordered before ca), on diagonal tiles the camera blocks S_cc + D_c^2 and a unit pivot on the padding rows (tile slots
// without a camera stay decoupled) — and written once.  Diagonal tiles also write their 64 rows of the right-hand side
// b = g_c + rb in elimination order (padding rows 0).
// Compose tile q = (ti,tj) in LDS (row stride LD doubles) and, for a diagonal tile, its 64 right-hand-side rows in rl.
// Called by all 256 threads; ends with a barrier.
struct FillLists { const int* tiles; const int* tptr; const int* tent; const double* Sblk; const int* blk_rc;
                   double radius; };      // radius > 0: the LM diagonal of the camera blocks is formed here (no k_point_prep launch), else read from d.Dc2
template <int LD>
__device__ __forceinline__ void compose_tile(const CholDev& c, const Dev& d, const FillLists& f, int q, double* A, double* rl) {
    const int ti = f.tiles[2 * q], tj = f.tiles[2 * q + 1];
    const int t = threadIdx.x;
    const int nrows = c.tile_rows[ti];
    for (int e = t; e < kNB * kNB; e += 256) {
        const int r = e >> 6, col = e & 63;
        A[r * LD + col] = (ti == tj && r == col && r >= nrows) ? 1.0 : 0.0;
    }
    if (rl && t < kNB) rl[t] = 0.0;
    __syncthreads();
    const int q0 = f.tptr[q], q1 = f.tptr[q + 1];
    for (int w = t; w < (q1 - q0) * 36; w += 256) {
        const int ent = f.tent[q0 + w / 36], e = w % 36, r = e / 6, col = e % 6;
        if (ent >= 0) {
            const int orow = c.cam_off[f.blk_rc[2 * ent]], ocol = c.cam_off[f.blk_rc[2 * ent + 1]];
            const double v = -f.Sblk[36 * (size_t)ent + e];
            if (orow > ocol) A[((orow & 63) + r) * LD + (ocol & 63) + col] = v;
            else A[((ocol & 63) + col) * LD + (orow & 63) + r] = v;
        } else {
            const int cam = -ent - 1, o = c.cam_off[cam] & 63;
            const double* S = d.camS + 28 * (size_t)cam;
            if (col >= r) {
                double v = S[6 * r - r * (r - 1) / 2 + (col - r)];      // packed upper triangle (r, col)
                if (r == col) v += f.radius > 0.0 ? clampd(d.camlin[12 * (size_t)cam + r], kLmDiagMin, kLmDiagMax) / f.radius : d.Dc2[6 * (size_t)cam + r];
                A[(o + col) * LD + o + r] = v;
            }
            if (rl && e < 6) rl[o + e] = d.camlin[12 * (size_t)cam + 6 + e] + S[21 + e];
        }
    }
    __syncthreads();
}

// list (or nullptr = every tile): the tiles to compose — the tiles outside the first level's columns when there are thousands of
// them (k_lv_factor<true> composes them in trailing workgroups otherwise, but holds one workgroup per CU: 40 000 tiles of a
// dissected photo collection took 2.1 ms there)
__global__ __launch_bounds__(256) void k_tile_fill(CholDev c, Dev d, const int* __restrict__ tiles, const int* __restrict__ tptr,
                                                   const int* __restrict__ tent, const double* __restrict__ Sblk,
                                                   const int* __restrict__ blk_rc, double radius, const int* __restrict__ list = nullptr) {
    __shared__ double A[kNB * (kNB + 1)];
    __shared__ double rl[kNB];
    const FillLists f{tiles, tptr, tent, Sblk, blk_rc, radius};
    const int q = list ? list[blockIdx.x] : (int)blockIdx.x;
    const int ti = tiles[2 * q], tj = tiles[2 * q + 1];
    const int t = threadIdx.x;
    compose_tile<kNB + 1>(c, d, f, q, A, rl);
    double* base = tile_ptr(c, ti, tj);
    for (int e = t; e < kNB * kNB; e += 256) {
        const int r = e >> 6, col = e & 63;
        base[(size_t)r * c.ld + col] = A[r * (kNB + 1) + col];
    }
    if (ti == tj && t < kNB) c.rhs[ti * kNB + t] = rl[t];
}

// the solution back in camera order
__global__ void k_sol_gather(CholDev c, double* __restrict__ out, int n_cams) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cams * c.cw) out[i] = c.x[c.cam_off[i / c.cw] + i % c.cw];
}

// ---- small helpers for the in-register 16x16 diagonal-block factorisation
__device__ __forceinline__ double readlane_d(double v, int lane) {   // lane must be wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// Broadcast of lane l (0..15) of every 16-lane row to the whole row: two v_mov_b32 row_newbcast.  Cheaper than v_readlane
// here (no SGPR round trip, no hazard nops; measured 31 -> 25 us per tile); l must fold to a constant (fully unrolled
// callers).  (One v_mov_b64_dpp through inline asm was slower: the asm is a scheduling barrier and needs manual nops.)
template <int L>
__device__ __forceinline__ double row_bcast_c(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + L, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + L, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_bcast(double v, int l) {
    switch (l) {
        case 0: return row_bcast_c<0>(v); case 1: return row_bcast_c<1>(v); case 2: return row_bcast_c<2>(v); case 3: return row_bcast_c<3>(v);
        case 4: return row_bcast_c<4>(v); case 5: return row_bcast_c<5>(v); case 6: return row_bcast_c<6>(v); case 7: return row_bcast_c<7>(v);
        case 8: return row_bcast_c<8>(v); case 9: return row_bcast_c<9>(v); case 10: return row_bcast_c<10>(v); case 11: return row_bcast_c<11>(v);
        case 12: return row_bcast_c<12>(v); case 13: return row_bcast_c<13>(v); case 14: return row_bcast_c<14>(v); default: return row_bcast_c<15>(v);
    }
}
// The same broadcast as ONE v_mov_b64_dpp (gfx90a+: 64-bit DPP moves exist for row_newbcast; the 64-bit builtin makes the
// compiler emit it and schedule / pad it itself — an inline-asm v_mov_b64_dpp was measured slower in round 1: scheduling barrier).
template <int L>
__device__ __forceinline__ double row_bcast64_c(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const long long r = __builtin_amdgcn_update_dpp((long long)0, x, 0x150 + L, 0xf, 0xf, true);
    return __builtin_bit_cast(double, r);
}
__device__ __forceinline__ double row_bcast64(double v, int l) {
    switch (l) {
        case 0: return row_bcast64_c<0>(v); case 1: return row_bcast64_c<1>(v); case 2: return row_bcast64_c<2>(v); case 3: return row_bcast64_c<3>(v);
        case 4: return row_bcast64_c<4>(v); case 5: return row_bcast64_c<5>(v); case 6: return row_bcast64_c<6>(v); case 7: return row_bcast64_c<7>(v);
        case 8: return row_bcast64_c<8>(v); case 9: return row_bcast64_c<9>(v); case 10: return row_bcast64_c<10>(v); case 11: return row_bcast64_c<11>(v);
        case 12: return row_bcast64_c<12>(v); case 13: return row_bcast64_c<13>(v); case 14: return row_bcast64_c<14>(v); default: return row_bcast64_c<15>(v);
    }
}
// Diagonal tile: L = chol(A) and Linv = L^-1, blocked 16x16.
//   per block column kb: (a) wave 0 factors the 16x16 diagonal block in registers (lane = row; column
//   broadcasts are DPP row_newbcast moves, one reciprocal per column, square roots applied once at the end) and
//   inverts it; (b) the rows below are multiplied by Linv11^T and (c) the trailing blocks are updated
//   with 16x16x16 products on the FP64 matrix cores.  3 barriers per block column.
//   With rptr/rj (level schedule) the forward substitution of the panel is folded in:
//   y_k = Linv_k (rhs_k - sum_{j in row(k)} L_kj y_j); every L_kj and y_j belongs to a lower level.
// The factorisation proper, on a tile that is already in LDS: A (lower triangle valid, upper zero) becomes L, Li becomes
// L^-1 (Li must hold the identity on the padding rows >= 16 nb and zeros elsewhere).  Called by all 256 threads of the
// workgroup after a barrier; ends with a barrier.
// (a) of potrf_lds: wave-level factorisation + inverse of the 16x16 diagonal block at (b0,b0), in registers
// PV: how a column update gets its broadcast operand — 0: two v_mov_b32_dpp (rounds 1-3), 1: one v_mov_b64_dpp (compiler-scheduled
// builtin).  Both form the same products with the same roundings.  Measured (tools/bench_potrf, MI355X, one 64x64 tile incl. the full
// inverse): PV 0 12.7 us, PV 1 11.9, PV 1 with OVL 10.6 — all bit-identical; the DEFAULT since round 5 is PV 4 + OVL (below), 9.7 us.
// (Measured in round 4 and removed in round 5: the broadcast folded into v_fmac_f64_dpp through inline asm, 13.4 us — the DPP form of
//  a 64-bit FMA issues slower than a DPP move + a plain FMA; the broadcasts of a group issued one group ahead of their FMAs, 12.0 us;
//  the elimination sweep WITHOUT the inverse and the forward substitution L X = I afterwards, operands from LDS: 15.2 us.)
#ifndef XBA_POTRF_PV
#define XBA_POTRF_PV 4
#endif
#ifndef XBA_POTRF_OVL
#define XBA_POTRF_OVL 1
#endif
// ---- PV 4 (round 5): the same 16x16 factorisation + inverse with the rank-1 updates of a PANEL of four pivot columns applied to the
// trailing columns as ONE rank-4 product on the FP64 matrix cores.
// The sweep above spends 4 900 cycles on ~480 dependent-ish VALU instructions of one wave (120 column updates x [DPP move + 2 FMA]);
// its dependent chain — pivot, reciprocal, two Newton steps, scale, update: 82 cycles per column — is a quarter of that.  Here the
// block lives in the accumulator layout of v_mfma_f64_16x16x4_f64: lane (li, lk) holds row li, columns lk, lk+4, lk+8, lk+12
// (four registers; lk = 16-lane row group of the wave), and so does the running right-hand side of L X = I (lane = column of X).
// Panel p = columns 4p .. 4p+3 = register p of the four lane groups:
//   1. all-gather the panel's four columns across the lane groups (v_permlane16_swap + v_permlane32_swap: 12 cheap VALU per 64-bit
//      value, no LDS) so that every lane holds its row's four entries;
//   2. the four elimination steps of the panel, every lane group redundantly (as the sweep above mirrors its rows): 3 + 2 + 1
//      in-panel column updates with DPP row broadcasts — the 82-cycle chain per column stays, the 120 updates shrink to 24;
//   3. trailing columns: acc -= T A_p^T with A operand = the lane's own final panel entry of column 4p+lk and B operand = its scaled
//      entry tl (and xs for the inverse): two v_mfma_f64_16x16x4_f64 per panel, results land in the distributed layout directly.
// The same products as the sweep, and the matrix instruction adds a panel's four terms in k order like the sweep does: measured
// bit-identical to PV 0 / PV 1 on tools/bench_potrf's tiles (profiles/r05_potrf.txt); that is an observation, not a contract —
// every caller uses ONE variant (the A/B tests compare schedules, not variants).
__device__ __forceinline__ void allgather4(double v, double (&out)[4]) {     // v: row group g holds x_g  ->  out[k] = x_k in every group
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l1 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);      // [x0 x0 x2 x2] | [x1 x1 x3 x3]
    const auto h1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto le = __builtin_amdgcn_permlane32_swap(l1[0], l1[0], false, false); // [x0 x0 x0 x0] | [x2 x2 x2 x2]
    const auto he = __builtin_amdgcn_permlane32_swap(h1[0], h1[0], false, false);
    const auto lo2 = __builtin_amdgcn_permlane32_swap(l1[1], l1[1], false, false); // [x1 ...] | [x3 ...]
    const auto ho2 = __builtin_amdgcn_permlane32_swap(h1[1], h1[1], false, false);
    out[0] = __hiloint2double((int)he[0], (int)le[0]); out[2] = __hiloint2double((int)he[1], (int)le[1]);
    out[1] = __hiloint2double((int)ho2[0], (int)lo2[0]); out[3] = __hiloint2double((int)ho2[1], (int)lo2[1]);
}
// (every index below folds to a constant: the panel's values live in named registers, not in arrays a loop variable indexes)
struct Panel4 { double a0, a1, a2, a3, x0, x1, x2, x3; };
template <int JJ>          // elimination step of pivot column JJ (= 4P + K) inside its panel: scale, then update the panel's later columns
__device__ __forceinline__ void potrf_panel_step(double& ak, double& xk, double& r, double& piv,
                                                 double* a_next, double* x_next, const int n_next) {
    piv = row_bcast64_c<JJ>(ak);
    r = fast_rcp(piv);
    const double tl = ak * r;               // u_ij / u_jj
    const double xs = xk * r;
    if (n_next >= 1) { const double bv = row_bcast64_c<(JJ + 1) & 15>(ak); a_next[0] = fma(-tl, bv, a_next[0]); x_next[0] = fma(-bv, xs, x_next[0]); }
    if (n_next >= 2) { const double bv = row_bcast64_c<(JJ + 2) & 15>(ak); a_next[1] = fma(-tl, bv, a_next[1]); x_next[1] = fma(-bv, xs, x_next[1]); }
    if (n_next >= 3) { const double bv = row_bcast64_c<(JJ + 3) & 15>(ak); a_next[2] = fma(-tl, bv, a_next[2]); x_next[2] = fma(-bv, xs, x_next[2]); }
}
template <int P>
__device__ __forceinline__ void potrf_panel4(v4d& aa, v4d& ax, bool m1, bool m2, double& Lc, double& Xc, double& Pv) {
    double pa[4], px[4];
    allgather4(aa[P], pa);
    allgather4(ax[P], px);
    double a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], x0 = px[0], x1 = px[1], x2 = px[2], x3 = px[3];
    double r0, r1, r2, r3, j0, j1, j2, j3;
    {
        double an[3] = {a1, a2, a3}, xn[3] = {x1, x2, x3};
        potrf_panel_step<4 * P + 0>(a0, x0, r0, j0, an, xn, 3);
        a1 = an[0]; a2 = an[1]; a3 = an[2]; x1 = xn[0]; x2 = xn[1]; x3 = xn[2];
    }
    {
        double an[3] = {a2, a3, 0.0}, xn[3] = {x2, x3, 0.0};
        potrf_panel_step<4 * P + 1>(a1, x1, r1, j1, an, xn, 2);
        a2 = an[0]; a3 = an[1]; x2 = xn[0]; x3 = xn[1];
    }
    {
        double an[3] = {a3, 0.0, 0.0}, xn[3] = {x3, 0.0, 0.0};
        potrf_panel_step<4 * P + 2>(a2, x2, r2, j2, an, xn, 1);
        a3 = an[0]; x3 = xn[0];
    }
    {
        double an[3] = {0.0, 0.0, 0.0}, xn[3] = {0.0, 0.0, 0.0};
        potrf_panel_step<4 * P + 3>(a3, x3, r3, j3, an, xn, 0);
    }
    auto sel = [&](double v0, double v1, double v2, double v3) { return m2 ? (m1 ? v3 : v2) : (m1 ? v1 : v0); };
    // operands of the lane's own panel column 4P+lk: the selected entry, its pivot's reciprocal, and the two scaled entries formed
    // from them (the same products as tl / xs of the step that owns the column; one negation serves both products)
    const double a_op = sel(a0, a1, a2, a3), x_op = sel(x0, x1, x2, x3), r_op = sel(r0, r1, r2, r3);
    if (P < 3) {
        const double na = -a_op;
        aa = __builtin_amdgcn_mfma_f64_16x16x4f64(na, a_op * r_op, aa, 0, 0, 0);
        ax = __builtin_amdgcn_mfma_f64_16x16x4f64(na, x_op * r_op, ax, 0, 0, 0);
    }
    // unscaled: the square roots (1 / sqrt(u_jj): column jj of L, row jj of the inverse) are applied after the sweep, off the
    // dependent chain pivot -> reciprocal -> scale -> update -> next pivot (a wave issues in order)
    Lc = a_op;                              // column 4P+lk of L, this lane's row
    Xc = x_op;                              // row 4P+lk of the inverse, this lane's column
    Pv = sel(j0, j1, j2, j3);               // u_jj of that column
}
__device__ __forceinline__ void potrf_block16_mfma(double (*A)[kLdT], double (*Li)[kLdT], int b0, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    v4d aa, ax;
#pragma unroll
    for (int g = 0; g < 4; ++g) { aa[g] = A[b0 + li][b0 + lk + 4 * g]; ax[g] = (lk + 4 * g == li) ? 1.0 : 0.0; }
    const bool m1 = (lk & 1) != 0, m2 = (lk & 2) != 0;
    double L0, L1, L2, L3, X0, X1, X2, X3, u0, u1, u2, u3;
    potrf_panel4<0>(aa, ax, m1, m2, L0, X0, u0);
    potrf_panel4<1>(aa, ax, m1, m2, L1, X1, u1);
    potrf_panel4<2>(aa, ax, m1, m2, L2, X2, u2);
    potrf_panel4<3>(aa, ax, m1, m2, L3, X3, u3);
    {   // fast_rsqrt of the lane's four pivots, the four chains side by side (same operations as fast_rsqrt)
        double y0 = __builtin_amdgcn_rsq(u0), y1 = __builtin_amdgcn_rsq(u1), y2 = __builtin_amdgcn_rsq(u2), y3 = __builtin_amdgcn_rsq(u3);
        const double h0 = 0.5 * u0, h1 = 0.5 * u1, h2 = 0.5 * u2, h3 = 0.5 * u3;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const double f0 = fma(-h0 * y0, y0, 1.5), f1 = fma(-h1 * y1, y1, 1.5), f2 = fma(-h2 * y2, y2, 1.5), f3 = fma(-h3 * y3, y3, 1.5);
            y0 = y0 * f0; y1 = y1 * f1; y2 = y2 * f2; y3 = y3 * f3;
        }
        L0 *= y0; X0 *= y0; L1 *= y1; X1 *= y1; L2 *= y2; X2 *= y2; L3 *= y3; X3 *= y3;
    }
    // column cc = 4p + lk of L (zero above the diagonal) and row cc of the inverse (zero above the diagonal by construction)
    A[b0 + li][b0 + lk] = (lk <= li) ? L0 : 0.0;             Li[b0 + lk][b0 + li] = X0;
    A[b0 + li][b0 + 4 + lk] = (4 + lk <= li) ? L1 : 0.0;     Li[b0 + 4 + lk][b0 + li] = X1;
    A[b0 + li][b0 + 8 + lk] = (8 + lk <= li) ? L2 : 0.0;     Li[b0 + 8 + lk][b0 + li] = X2;
    A[b0 + li][b0 + 12 + lk] = (12 + lk <= li) ? L3 : 0.0;   Li[b0 + 12 + lk][b0 + li] = X3;
}

template <int PV>
__device__ __forceinline__ void potrf_block16(double (*A)[kLdT], double (*Li)[kLdT], int b0, int lane) {
    if constexpr (PV == 4) { potrf_block16_mfma(A, Li, b0, lane); return; }
    const int li = lane & 15;
    // lanes 0..15 hold row `lane` of the diagonal block; other lanes mirror lane (lane & 15)
    double a[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) a[cc] = A[b0 + li][b0 + cc];
    // Right-looking elimination and the inverse of the triangular factor in ONE sweep: at step jj the pivot u_jj and
    // column jj of the (unscaled) factor are final, so step jj of the forward substitution L X = I (lane = column of
    // X) can run next to the elimination step — both use the same 15-jj broadcasts, and the two fma streams are
    // independent.  L[i][jj] = a[jj]_i * s_jj with s_jj = 1/sqrt(u_jj).
    double lcol[16];          // running right-hand side of column `li`; entry r becomes Linv[r][li] at step r
#pragma unroll
    for (int r = 0; r < 16; ++r) lcol[r] = (r == li) ? 1.0 : 0.0;
    // Software-pipelined over the columns.  A wave issues in order, and the reciprocal square root of a pivot (v_rsq_f64 + two
    // Newton steps) is a chain of 7 dependent instructions: left to the compiler it is issued in one piece in front of the
    // 15 - jj column updates of the step (16 x [chain + updates] = 2.1 us per block, and the pivot tile of every elimination
    // level is on the critical path of the LM iteration).  Here the update of the NEXT pivot column comes first, its v_rsq_f64
    // is issued right away, and the six Newton instructions are dealt out between six groups of the remaining column updates
    // (scheduling barriers pin the order), so the chain's latency hides behind independent work.  Same operations on the same
    // values as the plain loop (fast_rsqrt spelled out): bit-identical factor.
#define XBA_POTRF_UPD(cc)                                                \
    {                                                                    \
        const double bv_ = (PV == 1) ? row_bcast64(a[jj], (cc)) : row_bcast(a[jj], (cc)); /* u_{cc,jj} */       \
        a[(cc)] = fma(-tl, bv_, a[(cc)]);                                \
        lcol[(cc)] = fma(-bv_, xs, lcol[(cc)]);                          \
        asm volatile("" : "+v"(lcol[(cc)]));   /* pin it here: left alone the compiler sinks every update of the inverse behind the   \
                                                  factorisation and keeps all 120 broadcast values alive for it (240 registers, through AGPRs) */ \
    }
    // The chain from one pivot to the next needs only the RECIPROCAL of the pivot (u_ij / u_jj and x_j / u_jj): v_rcp_f64 + two
    // Newton steps = 5 dependent instructions, against 8 for the reciprocal square root and its square; the square roots
    // (scaling of column jj of L and of row jj of the inverse) are applied after the sweep, 16 independent chains.
    double piv[16];
    piv[0] = row_bcast(a[0], 0);
    double rj = fast_rcp(piv[0]);
#pragma clang loop unroll(full)
    for (int jj = 0; jj < 16; ++jj) {
        const double tl = a[jj] * rj;                      // u_ij / u_jj
        const double xs = lcol[jj] * rj;
        double r = 0.0, un = 1.0, e = 0.0;
        if (jj + 1 < 16) {
            XBA_POTRF_UPD(jj + 1)
            un = (PV == 1) ? row_bcast64(a[jj + 1], jj + 1) : row_bcast(a[jj + 1], jj + 1);
            piv[jj + 1] = un;
            r = __builtin_amdgcn_rcp(un);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
        for (int st = 0; st < 4; ++st) {
#pragma clang loop unroll(full)
            for (int cc = jj + 2 + st; cc < 16; cc += 4) XBA_POTRF_UPD(cc)
            if (jj + 1 < 16) {                           // r <- r + r (1 - u r), twice: two instructions each (fast_rcp spelled out)
                if (st % 2 == 0) e = fma(-un, r, 1.0);
                else r = fma(r, e, r);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        rj = r;
        __builtin_amdgcn_sched_barrier(0);     // keep the broadcasts of later steps from being hoisted (register pressure)
    }
#pragma clang loop unroll(full)
    for (int jj = 0; jj < 16; ++jj) {
        const double sj = fast_rsqrt(piv[jj]);
        a[jj] *= sj;                                       // column jj of L (rows >= jj)
        lcol[jj] *= sj;                                    // row jj of the inverse
    }
#undef XBA_POTRF_UPD
    if (lane < 16) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            A[b0 + lane][b0 + cc] = (cc <= lane) ? a[cc] : 0.0;
            Li[b0 + cc][b0 + lane] = lcol[cc];               // Linv[r][col]: zero above the diagonal by construction
        }
    }
}

// OVL: the off-diagonal blocks of Linv are formed INSIDE the block-column loop — row kb of the inverse (blocks (kb, j), j < kb:
// one per wave 1..3) next to wave 0's in-register factorisation of diagonal block kb + 1, whose ~1-2 us the other waves would
// otherwise spend at the barrier; everything a block needs (L_{kb,j..kb-1}: panels of earlier block columns; Linv_{j..kb-1,j}:
// earlier rows; Linv_{kb,kb}: factored before the previous barrier) is complete by then.  The same products in the same order
// as the three rounds after the loop (OVL = false, rounds 1-3): bit-identical inverse, three workgroup barriers and ~1.3 us less.
template <int PV, bool OVL>
__device__ __forceinline__ void potrf_lds_t(double (*A)[kLdT], double (*Li)[kLdT], double (*Tb)[16][17], int nb) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lk = lane >> 4;
    auto inv_block = [&](int i, int j) {                  // Linv_ij = -Linv_ii * sum_{kb=j}^{i-1} L_i,kb Linv_kb,j   (scratch Tb[j])
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int kb = j; kb < i; ++kb)
#pragma unroll
            for (int k0 = 0; k0 < 16; k0 += 4)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * i + li][16 * kb + k0 + lk], Li[16 * kb + k0 + lk][16 * j + li], acc, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) Tb[j][lk + 4 * g][li] = acc[g];
        __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the wave's scratch block is written
        __builtin_amdgcn_wave_barrier();
        v4d acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 4)
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[16 * i + li][16 * i + k0 + lk], Tb[j][k0 + lk][li], acc2, 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) Li[16 * i + lk + 4 * g][16 * j + li] = -acc2[g];
    };
    // Look-ahead: wave 0 factors diagonal block kb+1 as soon as it has updated it, while waves 1..3 finish the rest of the
    // trailing update of step kb (they would otherwise wait at a barrier for the 2 us the in-register factorisation takes).
    // (round 5) The loop starts at kb = -1 — nothing but wave 0's factorisation of block 0 — so that the in-register factorisation
    // has ONE call site: its ~4 KB of straight-line code exist once per kernel (the factor kernels are 50-60 KB of code against a
    // 64 KB instruction cache that two CUs share).
#pragma clang loop unroll(disable)          // (one copy of the in-register factorisation in the loop, not nb: the kernel must stay in the instruction cache)
    for (int kb = -1; kb < nb; ++kb) {
        const int b0 = 16 * kb;
        if (kb >= 0) {
            XBA_STAMP(1, 3 + 2 * kb);
            // (b) rows below: X = A21 * Linv11^T, one 16-row block per wave
            if (kb + 1 + wave < nb) {
                const int rb = 16 * (kb + 1 + wave);
                v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int k0 = 0; k0 < 16; k0 += 4)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[rb + li][b0 + k0 + lk], Li[b0 + li][b0 + k0 + lk], acc, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) A[rb + lk + 4 * g][b0 + li] = acc[g];
            }
            __syncthreads();
            XBA_STAMP(1, 4 + 2 * kb);
        }
        // (c) trailing update A_ij -= X_i X_j^T for kb < j <= i < nb: wave 0 takes the next diagonal block and factors it,
        // the other blocks are dealt round-robin to waves 1..3
        if (wave == 0) {
            if (kb + 1 < nb) {
                const int i = kb + 1;
                if (kb >= 0) {
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0 += 4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * i + li][b0 + k0 + lk], A[16 * i + li][b0 + k0 + lk], acc, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) A[16 * i + lk + 4 * g][16 * i + li] -= acc[g];
                    __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the wave's own block is written
                    __builtin_amdgcn_wave_barrier();
                }
                potrf_block16<PV>(A, Li, 16 * i, lane);
            }
        } else if (kb >= 0) {
            int idx = 0;
            for (int i = kb + 1; i < nb; ++i)
                for (int j = kb + 1; j <= i; ++j) {
                    if (i == kb + 1) continue;                 // (kb+1,kb+1): wave 0
                    if ((idx++ % 3) + 1 != wave) continue;
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0 += 4)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[16 * i + li][b0 + k0 + lk], A[16 * j + li][b0 + k0 + lk], acc, 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) A[16 * i + lk + 4 * g][16 * j + li] -= acc[g];
                }
            if (kb == nb - 1) XBA_STAMP_W1(1, 14);
            if (OVL && wave - 1 < kb) inv_block(kb, wave - 1);     // row kb of the inverse (its operands are complete, see above)
            if (kb == nb - 1) XBA_STAMP_W1(1, 15);
        }
        __syncthreads();
    }
    XBA_STAMP(1, 11);
    if (OVL) return;
    // ---- off-diagonal blocks of Linv, block diagonals dd = 1,2,3:  Linv_ij = -Linv_ii * sum_{kb=j}^{i-1} L_i,kb Linv_kb,j  (i = j + dd).
    // Block j of a diagonal belongs to wave j: two chains of 16x16x16 products on the matrix cores with the intermediate
    // passed through the wave's own LDS scratch; one workgroup barrier per diagonal.
    for (int dd = 1; dd < nb; ++dd) {
        const int j = wave, i = j + dd;
        if (i < nb) inv_block(i, j);
        __syncthreads();
    }
}
__device__ __forceinline__ void potrf_lds(double (*A)[kLdT], double (*Li)[kLdT], double (*Tb)[16][17], int nb) {
    potrf_lds_t<XBA_POTRF_PV, (XBA_POTRF_OVL != 0)>(A, Li, Tb, nb);
}


// C(64x64) (op)= alpha * A(64x64) * B(64x64)^T on the FP64 matrix cores.  A, B row-major tiles in LDS
// (stride kLdT).  4 waves, each a 32x32 quadrant = 2x2 MFMA 16x16 tiles, 16 k-steps of 4.
__device__ __forceinline__ void tile_abt_mfma(const double* __restrict__ As, const double* __restrict__ Bs, v4d (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < kNB; k0 += 4) {
        double a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[m] = As[(r0 + 16 * m + li) * kLdT + k0 + lk];
            b[m] = Bs[(c0 + 16 * m + li) * kLdT + k0 + lk];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) acc[m][n2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n2], acc[m][n2], 0, 0, 0);
    }
}

// 64x64 doubles = 2048 double2, 8 per thread: global -> registers and registers -> LDS as two steps, so that the loads
// of the next tile can be in flight while the matrix cores work on the current one
__device__ __forceinline__ void load_tile_regs(double2 (&v)[8], const double* __restrict__ src, size_t ld) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = threadIdx.x + 256 * it;          // double2 index
        const int r = e >> 5, c2 = (e & 31) * 2;
        v[it] = *reinterpret_cast<const double2*>(src + (size_t)r * ld + c2);
    }
}
__device__ __forceinline__ void store_tile_lds(double* dst, const double2 (&v)[8]) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = threadIdx.x + 256 * it;
        const int r = e >> 5, c2 = (e & 31) * 2;
        dst[r * kLdT + c2] = v[it].x; dst[r * kLdT + c2 + 1] = v[it].y;
    }
}
__device__ __forceinline__ void load_tile_lds(double* dst, const double* __restrict__ src, size_t ld) {
    double2 v[8];
    load_tile_regs(v, src, ld);
    store_tile_lds(dst, v);
}

// A_ik <- A_ik * Linv_k^T for the tiles i listed in rows[]
__device__ __forceinline__ void tile_gemv(const double* __restrict__ M, size_t ld, const double* v, double* out, bool transpose, double* red) {
    // 4 threads per output element
    const int t = threadIdx.x, o = t >> 2, part = t & 3;
    double s = 0.0;
    for (int m = part * 16; m < part * 16 + 16; ++m) s += (transpose ? M[(size_t)m * ld + o] : M[(size_t)o * ld + m]) * v[m];
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    if (part == 0) out[o] = s;
    (void)red;
}

// backward substitution, panel k: x_k = Linv_k^T y_k;  y_j -= L_kj^T x_k for j in cols[]
__global__ __launch_bounds__(256) void k_bwd(CholDev c, int k, const int* __restrict__ cols) {
    // (requesting this workgroup's tile L_kj together with Linv_k and y_k — one memory round trip per column instead of two — was
    //  measured in round 3: 12.2 us per column instead of 9.2 at config T; the 16 strided loads queue in front of the pivot's)
    __shared__ double v[kNB], xk[kNB], tmp[kNB];
    if (threadIdx.x < kNB) v[threadIdx.x] = c.y[k * kNB + threadIdx.x];
    __syncthreads();
    tile_gemv(c.Linv + (size_t)k * kNB * kNB, kNB, v, xk, true, nullptr);
    __syncthreads();
    if (blockIdx.x == 0) {
        if (threadIdx.x < kNB) c.x[k * kNB + threadIdx.x] = xk[threadIdx.x];
        return;
    }
    const int j = cols[blockIdx.x - 1];
    tile_gemv(tile_ptr(c, k, j), c.ld, xk, tmp, true, nullptr);
    __syncthreads();
    if (threadIdx.x < kNB) c.y[j * kNB + threadIdx.x] -= tmp[threadIdx.x];
}

// ... two columns per launch: every workgroup forms x_k = Linv_k^T y_k, then y'_{k-1} = y_{k-1} - L_{k,k-1}^T x_k and
// x_{k-1} = Linv_{k-1}^T y'_{k-1} (the same values in the same order as two launches of k_bwd), workgroup 0 stores both, workgroup
// b > 0 pushes both into one row tile j < k-1 of the union list (ent: j, flags — bit 0 L_kj non-zero, bit 1 L_{k-1,j}).  Half the
// launches of the push-form backward substitution of a panel schedule (9.2 us each, one per tile column, at config T).
__global__ __launch_bounds__(256) void k_bwd2(CholDev c, int k, int link, const int* __restrict__ ent) {
    __shared__ double v[kNB], xk[kNB], xk1[kNB], tmp[kNB], tmp2[kNB];
    const int t = threadIdx.x;
    if (t < kNB) v[t] = c.y[k * kNB + t];
    __syncthreads();
    tile_gemv(c.Linv + (size_t)k * kNB * kNB, kNB, v, xk, true, nullptr);
    __syncthreads();
    if (link) tile_gemv(tile_ptr(c, k, k - 1), c.ld, xk, tmp, true, nullptr);
    __syncthreads();
    if (t < kNB) v[t] = c.y[(k - 1) * kNB + t] - (link ? tmp[t] : 0.0);
    __syncthreads();
    tile_gemv(c.Linv + (size_t)(k - 1) * kNB * kNB, kNB, v, xk1, true, nullptr);
    __syncthreads();
    if (blockIdx.x == 0) {
        if (t < kNB) { c.x[k * kNB + t] = xk[t]; c.x[(k - 1) * kNB + t] = xk1[t]; }
        return;
    }
    const int j = ent[2 * (blockIdx.x - 1)], f = ent[2 * (blockIdx.x - 1) + 1];
    if (f & 1) tile_gemv(tile_ptr(c, k, j), c.ld, xk, tmp, true, nullptr);
    if (f & 2) tile_gemv(tile_ptr(c, k - 1, j), c.ld, xk1, tmp2, true, nullptr);
    __syncthreads();
    if (t < kNB) c.y[j * kNB + t] -= ((f & 1) ? tmp[t] : 0.0) + ((f & 2) ? tmp2[t] : 0.0);
}

__global__ void k_copy_pad(double* dst, const double* src, int n, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) dst[i] = (i < n) ? src[i] : 0.0;
}

// ------------------------------------------------------------ partial products of split levels / panel schedules
// Panels whose elimination-tree level is equal are independent; each launch covers one level (or, look-ahead schedule, one column).
// Targets never overlap inside a launch and every sum runs in list order, so the result is deterministic.
// (Round 1's right-looking kernels k_potrf / k_trsm / k_update and the launch-per-phase level kernels k_ll_update / k_ll_trsm /
//  k_ll_fwd / k_ll_bwd, kept as A/B switches through round 2, were removed in round 3: every schedule now runs the fused
//  factor kernel below.)

// Thin levels near the root of the elimination tree have few targets with long lists: one workgroup per CHUNK [q0,q1) of a
// target's list writes its partial sum (a full tile, + for a diagonal target (k,k) the 64 values sum_j L_kj y_j of the
// forward substitution, formed from the tile that is in LDS anyway) to the workspace Wp; k_ll_update_reduce adds the
// partials of every target in list order: deterministic, no atomics.
constexpr int kPartStride = kNB * kNB + kNB;

// The operands are staged in halves of 32 columns (2 x 17 KB of LDS per workgroup, < 128 registers): four workgroups
// share a compute unit, so that the loads of three of them are in flight while the fourth feeds the matrix cores.
constexpr int kLdH = 34;
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_half_regs(v2d (&v)[4], const double* __restrict__ src, size_t ld) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = threadIdx.x + 256 * it;          // 16-byte element index in the 64 x 32 half tile
        v[it] = *reinterpret_cast<const v2d*>(src + (size_t)(e >> 4) * ld + (e & 15) * 2);
    }
}
__device__ __forceinline__ void store_half_lds(double* dst, const v2d (&v)[4]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int e = threadIdx.x + 256 * it;
        *reinterpret_cast<v2d*>(dst + (e >> 4) * kLdH + (e & 15) * 2) = v[it];
    }
}
__device__ __forceinline__ void half_abt_mfma(const double* __restrict__ As, const double* __restrict__ Bs, v4d (&acc)[2][2]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += 4) {
        double a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            a[m] = As[(r0 + 16 * m + li) * kLdH + k0 + lk];
            b[m] = Bs[(c0 + 16 * m + li) * kLdH + k0 + lk];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) acc[m][n2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n2], acc[m][n2], 0, 0, 0);
    }
}

// lds: 4 * kNB * kLdH doubles (two buffers of the two half-tile operands), 16-byte aligned; yv: 2 * kNB — LDS of the calling kernel.
// The operands of four half-product steps are in flight in registers and the LDS image is double-buffered: the stores of step
// s+1 and the loads of step s+5 are issued ahead of the matrix instructions of step s, one barrier per step — with one
// workgroup per CU (k_panel_slot) nothing else hides a step's staging behind its 32 matrix instructions (measured at a quarter
// of config T with a single buffer and the loads one step ahead: 5.5 us per product against 1.7 us of matrix instructions).
// Same products in the same order.
__device__ __forceinline__ void ll_update_part_body(const CholDev& c, const int bx, const int* __restrict__ tgt, const int* __restrict__ qr,
                                                    const int* __restrict__ cj, double* __restrict__ Wp, double* lds, double* yv, int out_slot = -1) {
    const int i = tgt[2 * bx], k = tgt[2 * bx + 1];
    const bool diag = (i == k);
    v4d acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) acc[m][n2] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int q0 = qr[2 * bx], q1 = qr[2 * bx + 1];
    const int t = threadIdx.x, o = t >> 2, part = t & 3;
    double sv = 0.0;
    constexpr int kAhead = 4;
    v2d ra[kAhead][4], rb[kAhead][4];
    const int ns = 2 * (q1 - q0);
#pragma unroll
    for (int u = 0; u < kAhead; ++u)
        if (u < ns) {
            const int j = cj[q0 + (u >> 1)], col = (u & 1) * 32;
            load_half_regs(ra[u], tile_ptr(c, i, j) + col, c.ld);
            load_half_regs(rb[u], tile_ptr(c, k, j) + col, c.ld);
        }
    double yreg = (diag && t < kNB) ? c.y[cj[q0] * kNB + t] : 0.0;      // y_j of the product that starts at an even step
    // prologue: step 0 into buffer 0
    store_half_lds(lds, ra[0]);
    store_half_lds(lds + kNB * kLdH, rb[0]);
    if (diag && t < kNB) yv[t] = yreg;
    if (kAhead < ns) {
        const int j = cj[q0 + (kAhead >> 1)], col = (kAhead & 1) * 32;
        load_half_regs(ra[0], tile_ptr(c, i, j) + col, c.ld);
        load_half_regs(rb[0], tile_ptr(c, k, j) + col, c.ld);
    }
    if (diag && 2 < ns && t < kNB) yreg = c.y[cj[q0 + 1] * kNB + t];
    __syncthreads();
    for (int s0 = 0; s0 < ns; s0 += kAhead) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int s = s0 + u;
            if (s < ns) {                              // (uniform)
                const int kh = s & 1, buf = s & 1;
                double* As = lds + (size_t)buf * 2 * kNB * kLdH;
                double* Bs = As + kNB * kLdH;
                if (s + 1 < ns) {                      // operands of step s+1: registers -> the other buffer (free since the barrier)
                    const int un = (u + 1) % kAhead;      // (a constant once the loop over u is unrolled)
                    double* An = lds + (size_t)(buf ^ 1) * 2 * kNB * kLdH;
                    store_half_lds(An, ra[un]);
                    store_half_lds(An + kNB * kLdH, rb[un]);
                    if (diag && ((s + 1) & 1) == 0 && t < kNB) yv[(((s + 1) >> 1) & 1) * kNB + t] = yreg;
                    if (s + 1 + kAhead < ns) {         // ... and the registers refilled with step s+1+kAhead
                        const int j = cj[q0 + ((s + 1 + kAhead) >> 1)], col = ((s + 1 + kAhead) & 1) * 32;
                        load_half_regs(ra[un], tile_ptr(c, i, j) + col, c.ld);
                        load_half_regs(rb[un], tile_ptr(c, k, j) + col, c.ld);
                    }
                    if (diag && ((s + 1) & 1) == 0 && s + 3 < ns && t < kNB) yreg = c.y[cj[q0 + ((s + 3) >> 1)] * kNB + t];
                }
                if (diag) {
                    const double* yq = yv + ((s >> 1) & 1) * kNB;
#pragma unroll
                    for (int m = 0; m < 8; ++m) sv += As[o * kLdH + part * 8 + m] * yq[kh * 32 + part * 8 + m];
                }
                half_abt_mfma(As, Bs, acc);
                __syncthreads();                       // buffer `buf` is free for step s+2, buffer buf^1 is complete
            }
        }
    }
    double* out = Wp + (size_t)(out_slot >= 0 ? out_slot : bx) * kPartStride;
    {
        const int lane = t & 63, wave = t >> 6;
        const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = r0 + 16 * m + (lane >> 4) + 4 * g, col = c0 + 16 * n2 + (lane & 15);
                    out[r * kNB + col] = acc[m][n2][g];
                }
    }
    if (diag) {
        sv += __shfl_xor(sv, 1, kWave);
        sv += __shfl_xor(sv, 2, kWave);
        if (part == 0) out[kNB * kNB + o] = sv;
    }
}
// The launch of its own (level schedules): the same products in the same order with ONE LDS image of the two half-tile operands
// (35 KB) and the operands of one step ahead in registers (< 128 VGPRs): four workgroups per CU instead of the two the
// double-buffered body above allows (70 KB, 236 VGPRs — built for k_panel_slot, where a chunk workgroup has its CU to itself).  A
// chunk is short (six products at config T): its cold start and its partial-tile store are a quarter of its life, and with two
// workgroups per CU nothing else runs meanwhile (matrix cores busy 61 % of a busy CU's cycles).  XBA_CHUNK_SB=0: the other body.
#ifndef XBA_CHUNK_SB
#define XBA_CHUNK_SB 1
#endif
__device__ __forceinline__ void ll_update_part_body_sb(const CholDev& c, const int bx, const int* __restrict__ tgt, const int* __restrict__ qr,
                                                       const int* __restrict__ cj, double* __restrict__ Wp, double* As, double* Bs, double* yv, int out_slot) {
    const int i = tgt[2 * bx], k = tgt[2 * bx + 1];
    const bool diag = (i == k);
    v4d acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) acc[m][n2] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int q0 = qr[2 * bx], q1 = qr[2 * bx + 1];
    const int t = threadIdx.x, o = t >> 2, part = t & 3;
    double sv = 0.0;
    v2d ra[4], rb[4];
    {
        const int j = cj[q0];
        load_half_regs(ra, tile_ptr(c, i, j), c.ld);
        load_half_regs(rb, tile_ptr(c, k, j), c.ld);
    }
    const int ns = 2 * (q1 - q0);
    for (int s = 0; s < ns; ++s) {
        const int kh = s & 1;
        __syncthreads();                       // the previous half product no longer reads LDS
        store_half_lds(As, ra);
        store_half_lds(Bs, rb);
        if (diag && kh == 0 && t < kNB) yv[t] = c.y[cj[q0 + (s >> 1)] * kNB + t];
        __syncthreads();
        if (s + 1 < ns) {
            const int j = cj[q0 + ((s + 1) >> 1)], col = ((s + 1) & 1) * 32;
            load_half_regs(ra, tile_ptr(c, i, j) + col, c.ld);
            load_half_regs(rb, tile_ptr(c, k, j) + col, c.ld);
        }
        if (diag) {
#pragma unroll
            for (int m = 0; m < 8; ++m) sv += As[o * kLdH + part * 8 + m] * yv[kh * 32 + part * 8 + m];
        }
        half_abt_mfma(As, Bs, acc);
    }
    double* out = Wp + (size_t)out_slot * kPartStride;
    {
        const int lane = t & 63, wave = t >> 6;
        const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = r0 + 16 * m + (lane >> 4) + 4 * g, col = c0 + 16 * n2 + (lane & 15);
                    out[r * kNB + col] = acc[m][n2][g];
                }
    }
    if (diag) {
        sv += __shfl_xor(sv, 1, kWave);
        sv += __shfl_xor(sv, 2, kWave);
        if (part == 0) out[kNB * kNB + o] = sv;
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(XBA_CHUNK_SB ? 4 : 1, XBA_CHUNK_SB ? 4 : 2)))
void k_ll_update_part(CholDev c, const int* __restrict__ tgt, const int* __restrict__ qr, const int* __restrict__ cj, double* __restrict__ Wp,
                      const int* __restrict__ slot) {
    // slot (ba_plan.h: sp_slot): the partial slot of the chunk — its index in the level's list, or, with the level look-ahead, its
    // place inside its target's contiguous range (the early and the late chunks of a level are two launches)
    const int out_slot = slot[blockIdx.x];
#if XBA_CHUNK_SB
    __shared__ __attribute__((aligned(16))) double As[kNB * kLdH];
    __shared__ __attribute__((aligned(16))) double Bs[kNB * kLdH];
    __shared__ double yv[kNB];
    ll_update_part_body_sb(c, blockIdx.x, tgt, qr, cj, Wp, As, Bs, yv, out_slot);
#else
    __shared__ __attribute__((aligned(16))) double lds[4 * kNB * kLdH];
    __shared__ double yv[2 * kNB];
    ll_update_part_body(c, blockIdx.x, tgt, qr, cj, Wp, lds, yv, out_slot);
#endif
}

// Dense part of a panel schedule: 128x128 macro tile = rows (i0,i1) x columns (k0,k1), contributions j in [q0,q1): every
// operand tile that goes through LDS feeds two products (16 KB of traffic per 64x64x64 product instead of 32 KB) and every
// wave runs 16 matrix instructions per 8 LDS reads.  Wave w forms the full 64x64 tile (row w>>1, column w&1); tiles above
// the diagonal and the missing second row of an odd count are skipped.  The operands move in quarters of 16 columns through
// a double-buffered LDS image: the stores of step s+1 and the global loads of step s+2 are issued ahead of the matrix
// instructions of step s, one barrier per step.  mc: 8 ints per chunk (ba_plan.h); the four partial tiles land `stride`
// partial slots apart.
constexpr int kLdQ = 18;
__device__ __forceinline__ void load_quarter_regs(v2d (&v)[2], const double* __restrict__ src, size_t ld) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = threadIdx.x + 256 * it;          // 16-byte element index in the 64 x 16 quarter tile
        v[it] = *reinterpret_cast<const v2d*>(src + (size_t)(e >> 3) * ld + (e & 7) * 2);
    }
}
__device__ __forceinline__ void store_quarter_lds(double* dst, const v2d (&v)[2]) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = threadIdx.x + 256 * it;
        *reinterpret_cast<v2d*>(dst + (e >> 3) * kLdQ + (e & 7) * 2) = v[it];
    }
}

__global__ __launch_bounds__(256, 2) void k_panel2_part(CholDev c, const int* __restrict__ mc, const int* __restrict__ wg,
                                                        double* __restrict__ Wp) {
    __shared__ __attribute__((aligned(16))) double Rs[2][2][kNB * kLdQ];      // [buffer][row tile]
    __shared__ __attribute__((aligned(16))) double Cs[2][2][kNB * kLdQ];      // [buffer][column tile]
    __shared__ double yv[2][kNB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wa = wave >> 1, wb = wave & 1;
    const int o = t >> 2, part = t & 3, li = lane & 15, lk = lane >> 4;
  {
    const int* e = mc + 8 * wg[blockIdx.x];      // (one entry per workgroup: walking several made the compiler spill around the loop)
    const int i0 = e[0], i1 = e[1], k0 = e[2], k1 = e[3], q0 = e[4], q1 = e[5], out0 = e[6], stride = e[7];
    const int ia = wa ? i1 : i0, kb = wb ? k1 : k0;
    const bool active = ia >= 0 && ia >= kb;
    const bool diag = (i0 == k0);
    v4d acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) acc[m][n2] = (v4d){0.0, 0.0, 0.0, 0.0};
    const int i1r = i1 >= 0 ? i1 : i0;
    // quarter `st` (16 columns) of the operand tiles of step st: column tile q0 + st / 4
    auto load_step = [&](int st, v2d (&g0)[2], v2d (&g1)[2], v2d (&g2)[2], v2d (&g3)[2]) {
        const int j = q0 + (st >> 2), col = (st & 3) * 16;
        load_quarter_regs(g0, tile_ptr(c, i0, j) + col, c.ld); load_quarter_regs(g1, tile_ptr(c, i1r, j) + col, c.ld);
        load_quarter_regs(g2, tile_ptr(c, k0, j) + col, c.ld); load_quarter_regs(g3, tile_ptr(c, k1, j) + col, c.ld);
    };
    const int ns = 4 * (q1 - q0);
    v2d rg0[2], rg1[2], rg2[2], rg3[2];
    double yreg = 0.0;
    // prologue: step 0 into buffer 0, step 1 into the registers
    {
        load_step(0, rg0, rg1, rg2, rg3);
        if (diag && t < kNB) yreg = c.y[q0 * kNB + t];
        store_quarter_lds(Rs[0][0], rg0); store_quarter_lds(Rs[0][1], rg1);
        store_quarter_lds(Cs[0][0], rg2); store_quarter_lds(Cs[0][1], rg3);
        if (diag && t < kNB) yv[0][t] = yreg;
        if (ns > 1) {
            load_step(1, rg0, rg1, rg2, rg3);
        }
        __syncthreads();
    }
    double sv0 = 0.0, sv1 = 0.0;
    for (int s = 0; s < ns; ++s) {
        const int buf = s & 1;
        if (s + 1 < ns) {                      // operands of step s+1: registers -> the other buffer
            store_quarter_lds(Rs[buf ^ 1][0], rg0); store_quarter_lds(Rs[buf ^ 1][1], rg1);
            store_quarter_lds(Cs[buf ^ 1][0], rg2); store_quarter_lds(Cs[buf ^ 1][1], rg3);
            if (diag && ((s + 1) & 3) == 0 && t < kNB) yv[((s + 1) >> 2) & 1][t] = yreg;
        }
        if (s + 2 < ns) {                      // operands of step s+2: global -> registers
            load_step(s + 2, rg0, rg1, rg2, rg3);
            if (diag && ((s + 2) & 3) == 0 && t < kNB) yreg = c.y[(q0 + ((s + 2) >> 2)) * kNB + t];
        }
        if (diag) {
            const double* yq = yv[(s >> 2) & 1] + (s & 3) * 16 + part * 4;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                sv0 += Rs[buf][0][o * kLdQ + part * 4 + m] * yq[m];
                sv1 += Rs[buf][1][o * kLdQ + part * 4 + m] * yq[m];
            }
        }
        if (active) {
            const double* As = Rs[buf][wa];
            const double* Bs = Cs[buf][wb];
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                double a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    a[m] = As[(16 * m + li) * kLdQ + kk + lk];
                    b[m] = Bs[(16 * m + li) * kLdQ + kk + lk];
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n2 = 0; n2 < 4; ++n2) acc[m][n2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n2], acc[m][n2], 0, 0, 0);
            }
        }
        __syncthreads();                       // buffer `buf` is free for step s+2, buffer buf^1 is complete
    }
    if (active) {
        double* out = Wp + (size_t)(out0 + (2 * wa + wb) * stride) * kPartStride;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) out[(16 * m + lk + 4 * g) * kNB + 16 * n2 + li] = acc[m][n2][g];
    }
    if (diag) {        // forward substitution: sum_j L_kj y_j of both pivot rows, next to the partial tiles (k0,k0) and (k1,k1)
        sv0 += __shfl_xor(sv0, 1, kWave); sv0 += __shfl_xor(sv0, 2, kWave);
        sv1 += __shfl_xor(sv1, 1, kWave); sv1 += __shfl_xor(sv1, 2, kWave);
        if (part == 0) {
            Wp[(size_t)out0 * kPartStride + kNB * kNB + o] = sv0;
            if (i1 >= 0) Wp[(size_t)(out0 + 3 * stride) * kPartStride + kNB * kNB + o] = sv1;
        }
    }
  }
}

// Fixed-order sum of strided values with 8 loads in flight (a plain s += load loop is one L2 round trip per term).
__device__ __forceinline__ double sum_strided(const double* __restrict__ base, size_t stride, int n) {
    double s = 0.0;
    int p = 0;
    for (; p + 8 <= n; p += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(p + u) * stride];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; p < n; ++p) s += base[(size_t)p * stride];
    return s;
}

// grid (targets, 16): 256 tile elements per workgroup
// EPT tile elements per thread: 16 / EPT workgroups per target (standalone launch: 1 — short lists, as many CUs as possible; inside
// k_panel_slot, whose workgroups hold a CU each: 16 = one workgroup per target)
template <int EPT>
__device__ __forceinline__ void ll_update_reduce_body(const CholDev& c, const int bx, const int by, const int* __restrict__ rt,
                                                      const int* __restrict__ rp, const double* __restrict__ Wp) {
    const int i = rt[2 * bx], k = rt[2 * bx + 1];
    const int p0 = rp[2 * bx], p1 = rp[2 * bx + 1];
    if (EPT == 1) {
        const int e = by * 256 + threadIdx.x;
        const double s = sum_strided(Wp + (size_t)p0 * kPartStride + e, kPartStride, p1 - p0);
        tile_ptr(c, i, k)[(size_t)(e >> 6) * c.ld + (e & 63)] -= s;
    } else {
        // the thread's EPT elements advance together: EPT (x2) loads in flight per partial instead of EPT serial sums
        double s[EPT], a0[EPT];
        const double* W = Wp + (size_t)p0 * kPartStride + (size_t)by * EPT * 256 + threadIdx.x;
        double* const tik = tile_ptr(c, i, k);
#pragma unroll
        for (int u = 0; u < EPT; ++u) { s[u] = 0.0; a0[u] = tik[(size_t)(((by * EPT + u) * 256 + threadIdx.x) >> 6) * c.ld + (threadIdx.x & 63)]; }
        int p = p0;
        for (; p + 2 <= p1; p += 2) {
            double v0[EPT], v1[EPT];
#pragma unroll
            for (int u = 0; u < EPT; ++u) { v0[u] = W[(size_t)u * 256]; v1[u] = W[kPartStride + (size_t)u * 256]; }
#pragma unroll
            for (int u = 0; u < EPT; ++u) { s[u] += v0[u]; s[u] += v1[u]; }
            W += 2 * kPartStride;
        }
        if (p < p1) {
#pragma unroll
            for (int u = 0; u < EPT; ++u) s[u] += W[(size_t)u * 256];
        }
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const int e = (by * EPT + u) * 256 + threadIdx.x;
            tik[(size_t)(e >> 6) * c.ld + (e & 63)] = a0[u] - s[u];
        }
    }
    if (i == k && by == 0 && threadIdx.x < kNB) {
        const double s = sum_strided(Wp + (size_t)p0 * kPartStride + kNB * kNB + threadIdx.x, kPartStride, p1 - p0);
        c.rhs[k * kNB + threadIdx.x] -= s;
    }
}
__global__ __launch_bounds__(256) void k_ll_update_reduce(CholDev c, const int* __restrict__ rt, const int* __restrict__ rp,
                                                          const double* __restrict__ Wp) {
    ll_update_reduce_body<1>(c, blockIdx.x, blockIdx.y, rt, rp, Wp);
}

// ------------------------------------------------------------ fused level kernels
// One launch per elimination-tree level (on a split level after k_ll_update_part + k_ll_update_reduce, which leave the
// updated tiles and right-hand side in place: the lists are then empty).  One workgroup per structurally non-zero
// tile (i,k) of the level's columns k:
//   1. the update of the pivot tile, A_kk - sum_j L_kj L_kj^T, formed by EVERY workgroup of column k in the same order
//      (bit-identical), and — off-diagonal workgroups — of its own tile, A_ik - sum_j L_ij L_kj^T: products on the FP64 matrix
//      cores with the next operands prefetched.  (Summing a split level's partial tiles here as well was measured: 16-32
//      dependent L2 round trips per workgroup, 31 us per level; the 16-workgroups-per-tile reduction launch is faster.)
//   2. L_kk = chol, Linv_k (potrf_lds), again by every workgroup of the column: no tile of a level waits for another
//      workgroup, so a level costs one launch instead of update -> potrf -> trsm (three dependent launches, the pivot
//      factorisation of 12-17 us on the critical path either way);
//   3. diagonal workgroup: stores L_kk, Linv_k and the forward substitution y_k = Linv_k (rhs_k - sum_j L_kj y_j);
//      off-diagonal workgroup: L_ik = (A_ik - ...) Linv_k^T.
// Ceres solves the same system with a supernodal sparse Cholesky (ba_solver.cc:74); this is the exact solve, restated.
// FILL (first level of the tree; its lists are empty): the tiles are not read from S but composed here from the block values
// (what k_tile_fill does: one launch and a global round trip of every level-0 tile less per LM iteration; a single-tile
// system — LBA-sized calls — has no fill launch at all); workgroups >= n_factor compose the tiles of the other columns.
struct LvFill { Dev d; FillLists f; const int* fz_q; const int* rest; int n_factor; };
// late / Ql (look-ahead schedule, else nullptr): per entry the partial slots of (k,k) and (i,k) whose tiles the accumulators start
// from (the contribution of column k-2, formed by the previous launch), -1 none.
template <bool FILL>
__device__ __forceinline__ void lv_factor_body(const CholDev& c, const int b, const int* __restrict__ tiles, const int* __restrict__ dptr,
                                               const int* __restrict__ dj, const int* __restrict__ tile_cam, double* __restrict__ px,
                                               const LvFill& lf, double (*A)[kLdT], double (*Li)[kLdT], double (*Xs)[kLdT], double (*Tb)[16][17],
                                               double* yv, double* fv, const int* __restrict__ late = nullptr, const double* __restrict__ Ql = nullptr) {
    // LDS of the calling kernel: A, Li, Xs [kNB][kLdT], Tb [3][16][17], yv, fv [kNB].  Xs: off-diagonal workgroup: A_ik - update, parked
    // there while the pivot tile is factored (it used to stay in 64 registers across potrf_lds: with them the in-register 16x16
    // factorisation ran out of architectural VGPRs and copied every broadcast value through AGPRs — 4 of its 8 instructions per column
    // update; 109 KB of LDS = one workgroup per CU, the grid of a level has <= 1 per CU anyway)
    if (FILL && b >= lf.n_factor) {        // a tile of a later column: compose and store (k_tile_fill)
        const int q = lf.rest[b - lf.n_factor];
        const int ti = lf.f.tiles[2 * q], tj = lf.f.tiles[2 * q + 1];
        compose_tile<kLdT>(c, lf.d, lf.f, q, &A[0][0], yv);
        double* base = tile_ptr(c, ti, tj);
        for (int e = threadIdx.x; e < kNB * kNB; e += 256) {
            const int r = e >> 6, col = e & 63;
            base[(size_t)r * c.ld + col] = A[r][col];
        }
        if (ti == tj && threadIdx.x < kNB) c.rhs[ti * kNB + threadIdx.x] = yv[threadIdx.x];
        return;
    }
    const int i = tiles[2 * b], k = tiles[2 * b + 1];
    const bool diag = (i == k);
    XBA_STAMP(1, 0);
    const int nb = (c.tile_rows[k] + 15) >> 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int r0 = (wave >> 1) * 32, c0 = (wave & 1) * 32;
    const int o = t >> 2, part = t & 3;
    const size_t ld = c.ld;
    const double* Skk = tile_ptr(c, k, k);
    double* Sik = tile_ptr(c, i, k);
    // the assembled tiles, in the accumulator layout of tile_abt_mfma (requested now, used after the update phase)
    v4d skk[2][2], sik[2][2], akk[2][2], aik[2][2];
    if (FILL) {
        // off-diagonal tile first (composed in Li's storage, kept in registers), then the pivot tile straight into A
        if (!diag) compose_tile<kLdT>(c, lf.d, lf.f, lf.fz_q[2 * b + 1], &Li[0][0], nullptr);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    sik[m][n2][g] = diag ? 0.0 : Li[r0 + 16 * m + lk + 4 * g][c0 + 16 * n2 + li];
                    skk[m][n2][g] = 0.0; akk[m][n2][g] = 0.0; aik[m][n2][g] = 0.0;
                }
        __syncthreads();
        compose_tile<kLdT>(c, lf.d, lf.f, lf.fz_q[2 * b], &A[0][0], yv);      // yv: the 64 rhs rows of tile k
    } else {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = r0 + 16 * m + lk + 4 * g, col = c0 + 16 * n2 + li;
                skk[m][n2][g] = Skk[(size_t)r * ld + col];
                sik[m][n2][g] = diag ? 0.0 : Sik[(size_t)r * ld + col];
                akk[m][n2][g] = 0.0; aik[m][n2][g] = 0.0;
            }
    }
    const int late_kk = (!FILL && late) ? late[2 * b] : -1, late_ik = (!FILL && late) ? late[2 * b + 1] : -1;
    if (late_kk >= 0) {            // (requested with the assembled tiles: no round trip of its own)
        const double* Q = Ql + (size_t)late_kk * kPartStride;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) akk[m][n2][g] = Q[(r0 + 16 * m + lk + 4 * g) * kNB + c0 + 16 * n2 + li];
    }
    if (late_ik >= 0) {
        const double* Q = Ql + (size_t)late_ik * kPartStride;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int g = 0; g < 4; ++g) aik[m][n2][g] = Q[(r0 + 16 * m + lk + 4 * g) * kNB + c0 + 16 * n2 + li];
    }
    double fsum = 0.0;              // (diagonal workgroup) this thread's share of sum_j L_kj y_j, row o
    if (t < kNB) fv[t] = (diag && late_kk >= 0) ? Ql[(size_t)late_kk * kPartStride + kNB * kNB + t] : 0.0;
    if (!FILL) {
        double* As = &A[0][0]; double* Bs = &Li[0][0];
        const int qa = dptr[b], qb = dptr[b + 1];
        double2 ra[8], rb[8];
        if (qa < qb) {
            const int jj = dj[qa]; const bool own = jj >= 0; const int j = own ? jj : ~jj;
            load_tile_regs(rb, tile_ptr(c, k, j), ld);
            if (!diag && own) load_tile_regs(ra, tile_ptr(c, i, j), ld);
        }
        for (int q = qa; q < qb; ++q) {
            const int jj = dj[q]; const bool own = jj >= 0; const int j = own ? jj : ~jj;
            __syncthreads();                       // the previous products no longer read LDS
            store_tile_lds(Bs, rb);
            if (!diag && own) store_tile_lds(As, ra);
            if (diag && t < kNB) yv[t] = c.y[j * kNB + t];
            __syncthreads();
            if (q + 1 < qb) {                      // next contribution: loads in flight during the MFMAs below
                const int jn = dj[q + 1]; const bool ownn = jn >= 0; const int j2 = ownn ? jn : ~jn;
                load_tile_regs(rb, tile_ptr(c, k, j2), ld);
                if (!diag && ownn) load_tile_regs(ra, tile_ptr(c, i, j2), ld);
            }
            if (diag) {
#pragma unroll
                for (int m = 0; m < 16; ++m) fsum += Bs[o * kLdT + part * 16 + m] * yv[part * 16 + m];
            }
            tile_abt_mfma(Bs, Bs, akk);
            if (!diag && own) tile_abt_mfma(As, Bs, aik);
        }
        __syncthreads();
        if (diag) {
            fsum += __shfl_xor(fsum, 1, kWave);
            fsum += __shfl_xor(fsum, 2, kWave);
            if (part == 0) fv[o] += fsum;
        }
    }
    XBA_STAMP(1, 1);
    // the pivot tile: lower triangle of A_kk - update, identity on the padding rows of Linv
    // (FILL: A already holds the composed pivot tile — lower triangle, zeros above — and yv its right-hand side)
    double rhs_k = 0.0;
    if (FILL && t < kNB) rhs_k = yv[t];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int r = r0 + 16 * m + lk + 4 * g, col = c0 + 16 * n2 + li;
                if (!FILL) A[r][col] = (col <= r) ? skk[m][n2][g] - akk[m][n2][g] : 0.0;
                Li[r][col] = (r == col && r >= 16 * nb) ? 1.0 : 0.0;
                if (!diag) Xs[r][col] = sik[m][n2][g] - aik[m][n2][g];
            }
    __syncthreads();
    XBA_STAMP(1, 2);
    potrf_lds(A, Li, Tb, nb);
    XBA_STAMP(1, 12);
    if (diag) {
        // Only Linv_k is stored: nothing reads the factor of a pivot tile again (updates, substitutions and the backward
        // pass use the off-diagonal tiles and Linv), and S(k,k) must keep its assembled value while other workgroups of the
        // column may still be reading it.
        double* lo = c.Linv + (size_t)k * kNB * kNB;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int e = t + 256 * it; const int r = e >> 5, col = (e & 31) * 2;
            reinterpret_cast<double2*>(lo)[e] = make_double2((col <= r) ? Li[r][col] : 0.0, (col + 1 <= r) ? Li[r][col + 1] : 0.0);
        }
        if (t < kNB) yv[t] = (FILL ? rhs_k : c.rhs[k * kNB + t]) - fv[t];
        __syncthreads();
        double sacc = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) sacc += Li[o][part * 16 + m] * yv[part * 16 + m];
        sacc += __shfl_xor(sacc, 1, kWave);
        sacc += __shfl_xor(sacc, 2, kWave);
        if (part == 0) c.y[k * kNB + o] = sacc;
        if (px) {
            // last level of the tree: no column has tiles below it, so the backward substitution of the level is x_k = Linv_k^T y_k,
            // formed here (one launch and one kernel boundary less per solve); solution also in camera order, as k_lv_bwd writes it
            __syncthreads();
            if (part == 0) yv[o] = sacc;
            __syncthreads();
            double s2 = 0.0;
#pragma unroll
            for (int m = 0; m < 16; ++m) s2 += Li[part * 16 + m][o] * yv[part * 16 + m];
            s2 += __shfl_xor(s2, 1, kWave);
            s2 += __shfl_xor(s2, 2, kWave);
            if (part == 0) {
                c.x[k * kNB + o] = s2;
                const int cam = (o < c.cw * c.cpt) ? tile_cam[k * c.cpt + o / c.cw] : -1;
                if (cam >= 0) px[c.cw * (size_t)cam + o % c.cw] = s2;
            }
        }
        XBA_STAMP(1, 13);
        return;
    }
    // off-diagonal tile: L_ik = X Linv_k^T with X = A_ik - update (in Xs since before the factorisation)
    v4d acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) acc[m][n2] = (v4d){0.0, 0.0, 0.0, 0.0};
    tile_abt_mfma(&Xs[0][0], &Li[0][0], acc);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int g = 0; g < 4; ++g) Sik[(size_t)(r0 + 16 * m + lk + 4 * g) * ld + c0 + 16 * n2 + li] = acc[m][n2][g];
}
template <bool FILL>
__global__ __launch_bounds__(256) void k_lv_factor(CholDev c, const int* __restrict__ tiles, const int* __restrict__ dptr,
                                                   const int* __restrict__ dj, const int* __restrict__ tile_cam, double* __restrict__ px,
                                                   LvFill lf) {
    __shared__ double A[kNB][kLdT];
    __shared__ double Li[kNB][kLdT];
    __shared__ double Xs[kNB][kLdT];
    __shared__ double Tb[3][16][17];
    __shared__ double yv[kNB], fv[kNB];
    lv_factor_body<FILL>(c, blockIdx.x, tiles, dptr, dj, tile_cam, px, lf, A, Li, Xs, Tb, yv, fv);
}

// Look-ahead panel schedule (ba_plan.h: lookahead; level = column): ONE launch per column s.  Workgroups [0, n_factor): the
// fused factor kernel of column s (its list holds the contribution of column s-1, everything older than s-2 has been
// subtracted in place); then 16 per target: the fixed-order sum of the partial products of column s+1 into its tiles (written
// by the previous launch); one per tile of column s+1: its "late partial", the contribution of column s-1, which the next
// launch's factor workgroups start their accumulators from (one product instead of two on the critical path of a column);
// the rest: the partial products of column s+2 over the columns < s (one chunk each).  The
// parts touch disjoint data (columns s / s+1 / the partial buffer of the other parity), so the kernel boundary is the only
// synchronisation: the per-column chain update -> sum -> factor of the plain panel schedule becomes factor alone, with the
// other two riding on the CUs the factor kernel leaves idle.  The factor workgroups come first: they are the critical path
// and are dispatched first.  109 KB of LDS (the chunks overlay the factor tiles): one workgroup per CU.
//   Measured and not adopted (round 3): the same schedule on two streams — every event record / wait on the main stream cost
//   a 12-14 us bubble between dependent kernels (49 us per column at config U against 56 without look-ahead; this kernel: 39);
//   a 75 KB variant for two workgroups per CU (chunk operands overlaid on the factor tiles, the off-diagonal tile parked in
//   its own global storage, accumulators started from the negated tile to stay below 256 registers) — no gain at config T
//   (the column is bound by the factor workgroups, not by the chunks), slower at U and on the level schedule of L.
struct SlotArgs {
    const int* fz_tile; const int* fz_dptr; const int* fz_late; const double* Ql; int n_factor;   // column s (pointers offset to the level)
    const int* sp_rt; const int* sp_rp; int n_reduce; const double* Wr;      // column s+1: targets, partial ranges, their buffer
    const int* md_tgt; const int* md_q; int n_late; double* Wq;              // column s+1: late partials (the contribution of column s-1)
    const int* sp_tgt; const int* sp_q; int n_part; double* Wp;              // column s+2: chunks and their buffer
};
__global__ __launch_bounds__(256) void k_panel_slot(CholDev c, SlotArgs a, const int* __restrict__ dj, const int* __restrict__ cj,
                                                    const int* __restrict__ mcj, const int* __restrict__ tile_cam) {
    // order: the factor workgroups (critical path), the late partials (the next launch starts from them), the chunks (long), the sums (short)
    __shared__ __attribute__((aligned(16))) double T3[3][kNB][kLdT];      // A | Li | Xs of the factor workgroups; the chunks' operand buffers overlay them
    __shared__ double Tb[3][16][17];
    __shared__ double yv[2 * kNB], fv[kNB];
    static_assert(4 * kNB * kLdH <= 3 * kNB * kLdT, "the chunk operands fit the factor tiles");
    const int b = blockIdx.x;
    if (b < a.n_factor) {
        lv_factor_body<false>(c, b, a.fz_tile, a.fz_dptr, dj, tile_cam, nullptr, LvFill{}, T3[0], T3[1], T3[2], Tb, yv, fv, a.fz_late, a.Ql);
    } else if (b < a.n_factor + a.n_late) {
        ll_update_part_body(c, b - a.n_factor, a.md_tgt, a.md_q, mcj, a.Wq, &T3[0][0][0], yv);
    } else if (b < a.n_factor + a.n_late + a.n_part) {
        ll_update_part_body(c, b - a.n_factor - a.n_late, a.sp_tgt, a.sp_q, cj, a.Wp, &T3[0][0][0], yv);
    } else {
        ll_update_reduce_body<16>(c, b - a.n_factor - a.n_late - a.n_part, 0, a.sp_rt, a.sp_rp, a.Wr);
    }
}

// Backward substitution, one level per launch: x_k = Linv_k^T (y_k - sum_{i in col(k)} L_ik^T x_i).  Every term of the sum
// is requested at once (the column list is staged in LDS first); the solution is also written in camera order.  (Turning it
// into the candidate cameras here as well — quaternion plus on one lane per camera — put 4 us on the critical path of each
// of the five levels; that work now rides in trailing workgroups of k_backsub.)
__global__ __launch_bounds__(256) void k_lv_bwd(CholDev c, const int* __restrict__ klist, const int* __restrict__ cptr,
                                                const int* __restrict__ ci, const int* __restrict__ tile_cam, double* __restrict__ px) {
    __shared__ double acc[kNB];
    __shared__ int ids[64];
    const int k = klist[blockIdx.x];
    const int t = threadIdx.x, o = t >> 2, part = t & 3;
    const int q0 = cptr[blockIdx.x], q1 = cptr[blockIdx.x + 1];
    // what does not depend on the other tiles is requested first: this thread's 16 entries of Linv_k^T and y_k
    const double* Lk = c.Linv + (size_t)k * kNB * kNB;
    double lk[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) lk[m] = Lk[(part * 16 + m) * kNB + o];
    const double yk = c.y[k * kNB + o];
    // (round 5) the list is walked from its LAST entry — the ancestor nearest the root, solved first — down to the parent: the order in
    // which the one-launch form (k_lv_bwd_all) finds its operands ready; both forms add the same terms in the same order
    double s = 0.0;
    for (int qe = q1; qe > q0; qe -= 64) {
        const int qb = max(q0, qe - 64), n = qe - qb;
        __syncthreads();
        if (t < n) ids[t] = ci[qb + t];
        __syncthreads();
#pragma unroll 4
        for (int e = n - 1; e >= 0; --e) {
            const int i = ids[e];
            const double* M = tile_ptr(c, i, k) + (size_t)(part * 16) * c.ld + o;
            const double* xv = c.x + i * kNB + part * 16;
#pragma unroll
            for (int m = 0; m < 16; ++m) s += M[(size_t)m * c.ld] * xv[m];
        }
    }
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    if (part == 0) acc[o] = yk - s;
    __syncthreads();
    double s2 = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s2 += lk[m] * acc[part * 16 + m];
    s2 += __shfl_xor(s2, 1, kWave);
    s2 += __shfl_xor(s2, 2, kWave);
    if (part == 0) {
        c.x[k * kNB + o] = s2;
        const int cam = (o < c.cw * c.cpt) ? tile_cam[k * c.cpt + o / c.cw] : -1;
        if (cam >= 0) px[c.cw * (size_t)cam + o % c.cw] = s2;    // the solution in camera order (what k_sol_gather did)
    }
}

// ---- backward substitution of a DEEP level schedule (round 4): one launch per level, the tiles of a column shared out
// A dissected photo collection (ba_plan.h: ordering 3) has ~170 levels of a few columns with up to 70 tiles each: one workgroup per
// column (k_lv_bwd) walks 2 MB of tiles alone — 207 us per level —, the push form of the panel schedules (k_bwd2) needs one launch
// per two COLUMNS (380 launches, 5.4 ms), and the one-launch form above has most of its 760 workgroups polling.  Here a column's
// list is cut into chunks of kBwdChunk tiles, one workgroup each; a chunk stores its 64 partial sums write-through and takes a
// ticket on the column's counter, and the LAST chunk to arrive (hand-off recipe R1 of the hardware guide: payload stored
// write-through and drained, one relaxed agent-scope arrival per workgroup, acquire fence on the reading side — nobody waits)
// adds the partials in chunk order and forms x_k = Linv_k^T (y_k - sum): deterministic.  The counter is left at zero.
// (Ordering, for the reader who looks for a release: the payload is written by agent-scope atomic stores — `global_store ... sc1`,
//  performed at the device's coherence point, not parked in the XCD's L2 —, `s_waitcnt vmcnt(0)` + the workgroup barrier put every
//  one of them before the ticket, which is an agent-scope read-modify-write at the same point; the reader's acquire fence + agent-scope
//  loads complete the pair.  A __ATOMIC_RELEASE on the ticket would state the same in the HIP memory model and costs a
//  `buffer_wbl2` — a write-back of the whole L2 — per arriving workgroup; this is the guide's recipe R1, which names the instructions.)
constexpr int kBwdChunk = 4;          // (8: two dependent rounds of tile loads per workgroup, 32 us per level at config T)
__global__ __launch_bounds__(256) void k_lv_bwd_chunk(CholDev c, const int* __restrict__ klist, const int* __restrict__ ci, const int* __restrict__ tile_cam,
                                                      double* __restrict__ px, const int4* __restrict__ chunks, double* part_buf, unsigned* counter) {
    __shared__ double acc[kNB];
    __shared__ bool last;
    const int4 ch = chunks[blockIdx.x];                  // entry of the column in klist | list range [q0, q1) | chunk index + (chunks of the column << 16)
    const int k = klist[ch.x], q0 = ch.y, q1 = ch.z, idx = ch.w & 0xffff, nch = ch.w >> 16;
    const int t = threadIdx.x, o = t >> 2, part = t & 3;
    // (what only the last arrival uses is requested by everyone, up front: it would otherwise be two more dependent round trips at the end)
    const double* Lk = c.Linv + (size_t)k * kNB * kNB;
    double lk[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) lk[m] = Lk[(part * 16 + m) * kNB + o];
    const double yk = (t < kNB) ? c.y[k * kNB + t] : 0.0;
    double s = 0.0;
    for (int qb = q0; qb < q1; qb += 4) {
        const int n = min(4, q1 - qb);
        double M[4][16], xv[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < n) {
                const int i = ci[qb + u];
                const double* Mp = tile_ptr(c, i, k) + (size_t)(part * 16) * c.ld + o;
                const double* xp = c.x + i * kNB + part * 16;
#pragma unroll
                for (int m = 0; m < 16; ++m) { M[u][m] = Mp[(size_t)m * c.ld]; xv[u][m] = xp[m]; }
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < n) {
#pragma unroll
                for (int m = 0; m < 16; ++m) s += M[u][m] * xv[u][m];
            }
    }
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    if (nch > 1) {
        double* mine = part_buf + (size_t)blockIdx.x * kNB;
        if (part == 0) __hip_atomic_store(mine + o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) last = (__hip_atomic_fetch_add(counter + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)nch);
        __syncthreads();
        if (!last) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (t < kNB) {
            const double* first = part_buf + (size_t)(blockIdx.x - idx) * kNB + t;      // the chunks of a column are consecutive workgroups
            double sum = 0.0;
            for (int j = 0; j < nch; ++j) sum += __hip_atomic_load(first + (size_t)j * kNB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[t] = yk - sum;
        }
        if (t == 0) __hip_atomic_store(counter + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if (part == 0) acc[o] = c.y[k * kNB + o] - s;
    }
    __syncthreads();
    double s2 = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s2 += lk[m] * acc[part * 16 + m];
    s2 += __shfl_xor(s2, 1, kWave);
    s2 += __shfl_xor(s2, 2, kWave);
    if (part == 0) {
        c.x[k * kNB + o] = s2;
        const int cam = (o < c.cw * c.cpt) ? tile_cam[k * c.cpt + o / c.cw] : -1;
        if (cam >= 0) px[c.cw * (size_t)cam + o % c.cw] = s2;
    }
}

// ---- the whole backward substitution of a level schedule in ONE launch (round 4)
// One launch per level (k_lv_bwd above) costs 8-18 us per level — a kernel boundary, three or four dependent global round trips
// (column list -> tile addresses -> tiles -> result) and a grid of a few workgroups — for 64 values per column; a band with
// missed detections (config R) or loop closures (config LP) has 8-9 levels: 100 us of every LM iteration.  What travels
// between the levels is tiny (x_i: 64 doubles per column), and everything else a column needs — Linv_k^T, y_k, its tiles L_ik —
// was written by EARLIER kernels.  So: one workgroup per column, all levels in one launch; a workgroup requests its
// operands at once, then waits for the x_i of its ancestors, which arrive as DATA-TAGGED granules (8 bytes = {tag = epoch of
// this launch, 32 bits of the value}; the hand-off recipe R2 of cdna_hip_programming.md Guideline 16: agent-scope relaxed
// stores / loads of single aligned 8-byte words, the data is the flag, no fences) — a hop costs ~1-2 us instead of a launch.
//  * order[] lists the columns by level, root side first, and a workgroup takes its entry with a ticket (atomic counter), so
//    every workgroup only ever waits for workgroups that already run: no assumption about dispatch order or residency;
//  * tags: epoch counts the launches of the context (never 0; the granule buffer is zeroed when it is allocated), the ticket
//    counter is never reset (base = tickets handed out by earlier launches, modulo 2^32) — no memset per launch;
//  * every spin is bounded: after spin_max (kBwdSpinMax) polls the workgroup raises *err and the scalar slot S_BWD_ERR — which reaches
//    the host with the NEXT hand-over of the scalar block, i.e. before the LM controller takes another decision on a garbage x
//    (fetch_scalars() turns it into XRSFM_BA_EINTERNAL; round 6) — and goes on with what it has, so the launch
//    always terminates; wait_epoch != epoch is the test hook that makes every wait time out (XRSFM_BA_DEBUG_BWD_TIMEOUT);
//  * columns of the LAST level (col_final) were solved inside their k_lv_factor launch: their x is read from c.x;
//  * the sums run in the order of k_lv_bwd (list order, then row order): bit-identical solution (XRSFM_BA_BWD_ALL=0 is the A/B).
typedef __attribute__((address_space(1))) unsigned long long xba_gu64;
constexpr int kBwdPre = 4;                    // ancestors per round: one per wave polls, their tile values sit in registers
constexpr unsigned kBwdSpinMax = 1u << 21;
__global__ __launch_bounds__(256) void k_lv_bwd_all(CholDev c, const int* __restrict__ klist, const int* __restrict__ cptr, const int* __restrict__ ci,
                                                    const int* __restrict__ tile_cam, double* __restrict__ px, const int* __restrict__ order,
                                                    const unsigned char* __restrict__ col_final, unsigned long long* gx, unsigned* counter,
                                                    unsigned base, unsigned epoch, unsigned* err, double* err_scal, unsigned spin_max,
                                                    unsigned wait_epoch) {
    __shared__ double xs[kBwdPre][kNB];
    __shared__ double acc[kNB];
    __shared__ int s_b;
    const int t = threadIdx.x, o = t >> 2, part = t & 3, lane = t & 63, wave = t >> 6;
    if (t == 0) s_b = (int)(atomicAdd(counter, 1u) - base);
    __syncthreads();
    const int e = order[s_b];
    const int k = klist[e];
    const int q0 = cptr[e], q1 = cptr[e + 1];
    const double* Lk = c.Linv + (size_t)k * kNB * kNB;
    double lk[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) lk[m] = Lk[(part * 16 + m) * kNB + o];
    const double yk = c.y[k * kNB + o];
    // The column's list in ascending row order names its parent first and the ancestor nearest the root last; the root side is solved
    // first.  Walking the list forwards made every workgroup wait for its PARENT in its first round and run the other rounds — a tile-
    // load round trip each, 18 of them for a 70-tile column of a dissected photo collection — only afterwards: the per-level hop was
    // rounds x 2 us (20 ms per solve at config T, round 4).  Backwards, the rounds of the far ancestors run while the near ones are
    // still being solved and only the last round waits (round 5; k_lv_bwd walks the same way: bit-identical).
    double s = 0.0;
    for (int qe = q1; qe > q0; qe -= kBwdPre) {
        const int qb = max(q0, qe - kBwdPre), n = qe - qb;
        double M[kBwdPre][16];
#pragma unroll
        for (int u = 0; u < kBwdPre; ++u)
            if (u < n) {
                const double* Mp = tile_ptr(c, ci[qb + u], k) + (size_t)(part * 16) * c.ld + o;
#pragma unroll
                for (int m = 0; m < 16; ++m) M[u][m] = Mp[(size_t)m * c.ld];
            }
        if (wave < n) {                                  // wave w fetches x of ancestor w of this round
            const int i = ci[qb + wave];
            double xv;
            if (col_final[i]) xv = c.x[i * kNB + lane];
            else {
                xba_gu64* g = (xba_gu64*)(gx + (size_t)i * (2 * kNB) + 2 * lane);
                unsigned long long a0 = 0, a1 = 0;
                for (unsigned spins = 0;; ++spins) {
                    a0 = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    a1 = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ok = (unsigned)(a0 >> 32) == wait_epoch && (unsigned)(a1 >> 32) == wait_epoch;
                    if (__all(ok)) break;                // (wave-uniform exit)
                    if (spins >= spin_max) { if (lane == 0) { atomicOr(err, 1u); __hip_atomic_store(err_scal, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                xv = __hiloint2double((int)(unsigned)a1, (int)(unsigned)a0);
            }
            xs[wave][lane] = xv;
        }
        __syncthreads();
#pragma unroll
        for (int u = kBwdPre - 1; u >= 0; --u)
            if (u < n) {
#pragma unroll
                for (int m = 0; m < 16; ++m) s += M[u][m] * xs[u][part * 16 + m];
            }
        __syncthreads();                                 // xs is reused by the next round
    }
    s += __shfl_xor(s, 1, kWave);
    s += __shfl_xor(s, 2, kWave);
    if (part == 0) acc[o] = yk - s;
    __syncthreads();
    double s2 = 0.0;
#pragma unroll
    for (int m = 0; m < 16; ++m) s2 += lk[m] * acc[part * 16 + m];
    s2 += __shfl_xor(s2, 1, kWave);
    s2 += __shfl_xor(s2, 2, kWave);
    if (part == 0) {
        xba_gu64* g = (xba_gu64*)(gx + (size_t)k * (2 * kNB) + 2 * o);
        const unsigned long long tag = (unsigned long long)epoch << 32;
        __hip_atomic_store(g, tag | (unsigned)__double2loint(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g + 1, tag | (unsigned)__double2hiint(s2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.x[k * kNB + o] = s2;
        const int cam = (o < c.cw * c.cpt) ? tile_cam[k * c.cpt + o / c.cw] : -1;
        if (cam >= 0) px[c.cw * (size_t)cam + o % c.cw] = s2;    // the solution in camera order
    }
}

__global__ void k_zero_vec(double* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace xba
