// Device-side packing (round 4): the track-major 64-slot tiles + camera-major positions of ba_pack.h, built ON THE GPU from the
// caller's raw arrays.  pack_problem() (ba_pack.h) stays the specification and the fallback: this path produces the SAME arrays
// element for element (tests/test_gpu_pack.py compares every array of both), it only replaces the host's sorts and gathers —
// 27-30 ms of a one-shot global BA of 2 M observations on 16 host threads, the largest part of what BASolver::GBA / KGBA pay
// around the solve (/root/reference/src/optimization/ba_solver.cc:596-607 builds its ceres::Problem per call the same way) — by
// radix sorts (rocPRIM device primitives: set-up plumbing, not the hot path) and a handful of kernels:
//   A  (point, camera, input index) sort of the observations            = the host's CSR by point + per-track camera sort
//   B  track boundaries, lengths, the list of active points
//   C  LSD multi-key sort of the tracks by their camera tuple (4 cameras per 64-bit key)     = the host's tuple sort
//   D  groups of equal tuples -> "big group" starts (a group that fills a tile starts on a tile boundary)
//   E  placement (one sequential pass over the tracks: host, 1 byte per track down, 4 bytes per track up)
//   F  slot fill (cam / point / observation index / u / v), points in packed order
//   G  per-tile analysis: longest track, regular tiles, Gram tiles (distinct cameras, camera index per slot, pair cells)
//   H  camera-major positions of the writers (stable sort by camera), twice (linearisation / S assembly)
// Taken when the problem has no track longer than 64 observations, fewer than 65 535 cameras, 6-wide camera blocks and enough
// observations (32 768) to pay for ~25 launches and three round trips (XRSFM_BA_DEVICE_PACK=0 / 1 forces host / device).
// Measured on MI355X (tools/pack_crossover.py, tools/pack_phases.py): config 4 (2 M observations) xrsfm_ba_create 31 -> 3.7 ms
// (upload + observation sort 1.4, tracks + tuple sort 0.5, placement on the host 0.95, slots / tiles / camera-major 0.7), Cholesky
// set-up 5.7 -> 1.0 ms (device_keys below); 48 k observations 1.5 -> 1.2 ms; below 30 k the host packing is faster.
#pragma once
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "ba_pack.h"

namespace xba {
namespace devpack {

constexpr int kThreads = 256;
constexpr int kTupCams = 4;                         // cameras per 64-bit sort key (fields of bits(camera id + 1) <= 16 bits)

// status word: bit 0 index out of range, bit 1 a track longer than 64 observations (host path instead)
// observation key: (point << cshift) | camera, cshift = bits of a camera id (the sort then runs over cshift + bits(point) bits only)
__global__ void k_keys(const int* __restrict__ obs_cam, const int* __restrict__ obs_pt, int n_obs, int n_cams, int n_pts, int cshift,
                       unsigned long long* __restrict__ key, int* __restrict__ val, unsigned* __restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_obs) return;
    const int c = obs_cam[i], j = obs_pt[i];
    if (c < 0 || c >= n_cams || j < 0 || j >= n_pts) { atomicOr(status, 1u); key[i] = ~0ull; val[i] = i; return; }
    key[i] = ((unsigned long long)(unsigned)j << cshift) | (unsigned)c;
    val[i] = i;
}
// head[r] = 1 where a track starts in the sorted observation list
__global__ void k_heads(const unsigned long long* __restrict__ key, int n_obs, int cshift, int* __restrict__ head) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_obs) return;
    head[r] = (r == 0 || (key[r] >> cshift) != (key[r - 1] >> cshift)) ? 1 : 0;
}
// trk[r] = inclusive scan of head - 1 (track rank of observation r); ptr[t] = first observation of track t; pt_of[t] = its point
__global__ void k_track_ptr(const unsigned long long* __restrict__ key, const int* __restrict__ head, const int* __restrict__ trk_incl, int n_obs, int cshift,
                            int* __restrict__ ptr, int* __restrict__ pt_of) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_obs) return;
    if (head[r]) { const int t = trk_incl[r] - 1; ptr[t] = r; pt_of[t] = (int)(key[r] >> cshift); }
}
__global__ void k_track_len(const int* __restrict__ ptr, int n_trk, int n_obs, int* __restrict__ len, int* __restrict__ maxlen, unsigned* __restrict__ status) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = t < n_trk ? ((t + 1 < n_trk) ? ptr[t + 1] : n_obs) - ptr[t] : 0;
    if (t < n_trk) len[t] = l;
    if (l > 64) atomicOr(status, 2u);
    int m = l;                                   // one atomic per wave, not per track (half a million atomics on one word: 90 us)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(maxlen, m);
}
// sort key of camera group g (cameras 4g .. 4g+3 of the tuple, id + 1, 0 = beyond the end) of the track at sorted position r
__global__ void k_tuple_key(const unsigned long long* __restrict__ obs_key, const int* __restrict__ ptr, const int* __restrict__ len,
                            const int* __restrict__ perm, int n_trk, int g, unsigned long long cmask, int fbits, unsigned long long* __restrict__ key) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_trk) return;
    const int t = perm[r], l = len[t], p0 = ptr[t];
    unsigned long long k = 0;
#pragma unroll
    for (int q = 0; q < kTupCams; ++q) {
        const int e = kTupCams * g + q;
        const unsigned long long c = (e < l) ? (obs_key[p0 + e] & cmask) + 1ull : 0ull;
        k |= c << (fbits * (kTupCams - 1 - q));
    }
    key[r] = k;
}
__global__ void k_iota(int* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i; }
// gs[r] = 1 if the track at sorted position r starts a new group of equal tuples
__global__ void k_group_start(const unsigned long long* __restrict__ obs_key, const int* __restrict__ ptr, const int* __restrict__ len,
                              const int* __restrict__ perm, int n_trk, unsigned long long cmask, int* __restrict__ gs) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_trk) return;
    if (r == 0) { gs[0] = 1; return; }
    const int a = perm[r - 1], b = perm[r];
    const int la = len[a], lb = len[b];
    bool same = la == lb;
    for (int q = 0; same && q < la; ++q) same = (obs_key[ptr[a] + q] & cmask) == (obs_key[ptr[b] + q] & cmask);
    gs[r] = same ? 0 : 1;
}
// gid_incl = inclusive scan of gs; gstart[g] = sorted position of the first track of group g (+ sentinel n_trk)
__global__ void k_group_pos(const int* __restrict__ gs, const int* __restrict__ gid_incl, int n_trk, int* __restrict__ gstart) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_trk) return;
    if (r == n_trk) { gstart[gid_incl[n_trk - 1]] = n_trk; return; }
    if (gs[r]) gstart[gid_incl[r] - 1] = r;
}
// one byte per track in sorted order for the host's placement pass: length | big-group-start << 7
__global__ void k_track_byte(const int* __restrict__ len, const int* __restrict__ perm, const int* __restrict__ gs, const int* __restrict__ gid_incl,
                             const int* __restrict__ gstart, int n_trk, unsigned char* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_trk) return;
    const int l = len[perm[r]];
    int big = 0;
    if (gs[r]) { const int g = gid_incl[r] - 1; const long long n = gstart[g + 1] - gstart[g]; big = (l <= 64 && n * l >= 64) ? 1 : 0; }
    out[r] = (unsigned char)(l | (big << 7));
}
__global__ void k_inv_perm(const int* __restrict__ perm, int n, int* __restrict__ inv) { const int r = blockIdx.x * blockDim.x + threadIdx.x; if (r < n) inv[perm[r]] = r; }
// slots of one observation (sorted position r): tile layout of the track's packed rank
__global__ void k_fill_slots(const unsigned long long* __restrict__ obs_key, const int* __restrict__ obs_idx, const int* __restrict__ trk_incl,
                             const int* __restrict__ ptr, const int* __restrict__ inv, const int* __restrict__ trk_start, const double* __restrict__ obs_uv,
                             int n_obs, unsigned long long cmask, int* __restrict__ slot_cam, int* __restrict__ slot_pt, int* __restrict__ slot_obs, double* __restrict__ slot_u,
                             double* __restrict__ slot_v) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_obs) return;
    const int t = trk_incl[r] - 1, pj = inv[t];
    const int s = trk_start[pj] + (r - ptr[t]);
    const int i = obs_idx[r];
    slot_cam[s] = (int)(obs_key[r] & cmask); slot_pt[s] = pj; slot_obs[s] = i;
    slot_u[s] = obs_uv[2 * (size_t)i]; slot_v[s] = obs_uv[2 * (size_t)i + 1];
}
__global__ void k_points(const int* __restrict__ perm, const int* __restrict__ pt_of, const double* __restrict__ points, const unsigned char* __restrict__ point_const,
                         int n_trk, int* __restrict__ pt_orig, double* __restrict__ P, unsigned char* __restrict__ pt_const) {
    const int pj = blockIdx.x * blockDim.x + threadIdx.x;
    if (pj >= n_trk) return;
    const int j = pt_of[perm[pj]];
    pt_orig[pj] = j;
    P[3 * (size_t)pj] = points[3 * (size_t)j]; P[3 * (size_t)pj + 1] = points[3 * (size_t)j + 1]; P[3 * (size_t)pj + 2] = points[3 * (size_t)j + 2];
    pt_const[pj] = (point_const && point_const[j]) ? 1 : 0;
}

// ---- per-tile analysis: one wave per tile, lane = slot (ba_pack.h: "maxlen + regular tiles", "gram: cameras of tiles")
// out: tile_maxlen, tile_stride, tile_ncam (before the big-tile demotion), slot_cidx, cells2[tile] = C^2 (0 if not a Gram tile),
// tile_big[tile] = 1 if its LDS need exceeds the small class
__global__ __launch_bounds__(256) void k_tiles(const int* __restrict__ slot_cam, const int* __restrict__ slot_pt, int n_tiles, int gram_max_cams, int gram_cw,
                                               int* __restrict__ tile_maxlen, int* __restrict__ tile_stride, int* __restrict__ tile_ncam,
                                               unsigned char* __restrict__ slot_cidx, int* __restrict__ cells2, int* __restrict__ tile_big) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int s = 64 * t + lane;
    const int cam = slot_cam[s], pt = slot_pt[s];
    // valid prefix: the host loops stop at the first slot without an observation (a tile's observations are contiguous from slot 0)
    const unsigned long long vm = __ballot(cam >= 0);
    const int nvalid = (vm == ~0ull) ? 64 : __ffsll((long long)~vm) - 1;
    const bool valid = lane < nvalid;
    const int prev_pt = __shfl_up(pt, 1, 64);
    const bool head = valid && (lane == 0 || prev_pt != pt);
    const unsigned long long hm = __ballot(head);
    // longest run
    int best = 1;
    {
        // run length ending at each lane = lane - (position of the last head at or before lane) + 1
        const unsigned long long upto = hm & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
        const int last_head = upto ? 63 - __clzll((long long)upto) : 0;
        int run = valid ? lane - last_head + 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) run = max(run, __shfl_xor(run, off, 64));
        best = max(1, run);
    }
    int stride = 0;
    if (nvalid > 0) {
        // first track length L
        const unsigned long long rest = hm & ~1ull;
        const int L = rest ? __ffsll((long long)rest) - 1 : nvalid;
        if (L <= 32 && L < 64) {
            const int r = lane % L;
            const int cam_r = __shfl(cam, r, 64);
            bool ok = true;
            if (valid && lane >= L) {
                ok = cam == cam_r;
                ok = ok && ((r > 0) ? (pt == prev_pt) : (pt != prev_pt));
            }
            const bool regular = __all(ok || !valid);
            if (regular && nvalid % L == 0 && nvalid >= 2 * L && gram_cw == 6) stride = L;
        }
    }
    // Gram tile: distinct cameras, ascending
    int C = 0, cidx = 255, ntrk = __popcll(hm);
    if (nvalid > 0) {
        bool first = valid;                 // first occurrence of its camera among the valid lanes
        int less = 0;                       // distinct cameras smaller than mine
        for (int l = 0; l < nvalid; ++l) {
            const int cl = __shfl(cam, l, 64);
            if (valid && l < lane && cl == cam) first = false;
        }
        const unsigned long long fm = __ballot(first);
        C = __popcll(fm);
        for (int l = 0; l < nvalid; ++l) {
            const int cl = __shfl(cam, l, 64);
            if ((fm >> l) & 1ull) less += (valid && cl < cam) ? 1 : 0;
        }
        cidx = valid ? less : 255;
    }
    int passes = 1;
    const bool gram = nvalid > 0 && C >= 2 && C <= gram_max_cams && gram_lds_need(C, ntrk, &passes, gram_cw) <= kGramMaxLds;
    if (lane == 0) {
        tile_maxlen[t] = best; tile_stride[t] = stride; tile_ncam[t] = gram ? C : 0;
        cells2[t] = gram ? C * C : 0;
        tile_big[t] = (gram && gram_lds_need(C, ntrk, &passes, gram_cw) > kGramSmallLds) ? 1 : 0;
    }
    slot_cidx[s] = (unsigned char)((gram && valid) ? cidx : 255);
}
// tile_gt_off (exclusive scan of cells2) -> -1 for tiles without a table; the pair cells of every Gram tile (before the demotion, like the host)
__global__ __launch_bounds__(256) void k_gram_cells(const int* __restrict__ slot_cam, const int* __restrict__ slot_pt, const unsigned char* __restrict__ slot_cidx,
                                                    const int* __restrict__ tile_ncam, const int* __restrict__ gt_scan, int n_tiles, int* __restrict__ tile_gt_off,
                                                    unsigned char* __restrict__ gt_cell) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int C = tile_ncam[t];
    if (lane == 0) tile_gt_off[t] = C > 0 ? gt_scan[t] : -1;
    if (C <= 0) return;
    const int s = 64 * t + lane;
    const int cam = slot_cam[s], pt = slot_pt[s], ci = slot_cidx[s];
    unsigned char* cell = gt_cell + gt_scan[t];
    for (int d = 1; d < 64; ++d) {                      // pairs inside a track (cameras ascend in a track)
        const int pt2 = __shfl_down(pt, d, 64), cam2 = __shfl_down(cam, d, 64), ci2 = __shfl_down(ci, d, 64);
        if (cam >= 0 && lane + d < 64 && cam2 >= 0 && pt2 == pt) cell[ci * C + ci2] = 1;
    }
}
// big tiles demoted to the per-pair path when they are too few for a launch of their own (ba_pack.h); then the writer flags
__global__ __launch_bounds__(256) void k_demote_flags(const int* __restrict__ slot_cam, int n_tiles, int demote, const int* __restrict__ tile_big,
                                                      const int* __restrict__ tile_stride, int* __restrict__ tile_ncam, int* __restrict__ tile_gt_off,
                                                      unsigned char* __restrict__ slot_cidx, int n_cams, unsigned* __restrict__ key_all, unsigned* __restrict__ key_gram) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int s = 64 * t + lane;
    int C = tile_ncam[t];
    if (demote && C > 0 && tile_big[t]) {
        C = 0;
        slot_cidx[s] = 255;
        if (lane == 0) { tile_ncam[t] = 0; tile_gt_off[t] = -1; }
    }
    const int cam = slot_cam[s];
    const int L = tile_stride[t];
    const bool w = cam >= 0 && (L == 0 || lane < L);
    bool g = w;
    if (C > 0) {                                            // (wave-uniform)
        const int ci_v = cam >= 0 ? (int)slot_cidx[s] : -1;
        bool first = cam >= 0;                              // first lane of its camera index
        for (int l = 0; l < 64; ++l) { const int cl = __shfl(ci_v, l, 64); if (l < lane && cl == ci_v) first = false; }
        g = first;
    }
    key_all[s] = w ? (unsigned)cam : (unsigned)n_cams;
    key_gram[s] = g ? (unsigned)cam : (unsigned)n_cams;
}
__global__ void k_count_cams(const unsigned* __restrict__ key, int n, int n_cams, int* __restrict__ cnt) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n && key[s] < (unsigned)n_cams) atomicAdd(cnt + key[s], 1);
}
__global__ void k_campos(const unsigned* __restrict__ sorted_key, const int* __restrict__ sorted_slot, int n, int n_cams, int* __restrict__ campos) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) campos[sorted_slot[r]] = sorted_key[r] < (unsigned)n_cams ? r : -1;
}

#define XBA_DP_HIP(x) do { if ((x) != hipSuccess) return XRSFM_BA_ENODEV; } while (0)

struct Result {
    // device arrays (from the allocator passed in); Packed gets the small host-side pieces
    int *slot_cam = nullptr, *slot_pt = nullptr, *slot_obs = nullptr, *slot_campos = nullptr, *slot_campos_g = nullptr;
    double *slot_u = nullptr, *slot_v = nullptr, *P = nullptr;
    int *tile_stride = nullptr, *tile_maxlen = nullptr, *tile_ncam = nullptr, *tile_gt_off = nullptr, *cam_ptr = nullptr, *cam_ptr_g = nullptr, *pt_orig = nullptr;
    unsigned char *slot_cidx = nullptr, *gt_cell = nullptr, *pt_const = nullptr;
};

// rocPRIM sorts / scans on the packing stream with one temporary buffer that grows on demand (device_pack and device_keys)
template <typename TmpAlloc>
struct PrimOps {
    hipStream_t st; TmpAlloc tmp;
    void* d_tmp = nullptr; size_t tb = 0;
    bool need_tmp(size_t bytes) { if (bytes <= tb) return true; d_tmp = tmp(bytes); tb = d_tmp ? bytes : 0; return d_tmp != nullptr; }
    template <typename K> int sort_pairs(K* kin, K* kout, int* vin, int* vout, size_t n, unsigned bits) {
        size_t q = 0;
        if (rocprim::radix_sort_pairs(nullptr, q, kin, kout, vin, vout, n, 0u, bits, st) != hipSuccess) return XRSFM_BA_ENODEV;
        if (!need_tmp(q)) return XRSFM_BA_ENOMEM;
        size_t t = tb;
        return rocprim::radix_sort_pairs(d_tmp, t, kin, kout, vin, vout, n, 0u, bits, st) == hipSuccess ? 0 : XRSFM_BA_ENODEV;
    }
    int scan_incl(int* in, int* out, size_t n) {
        size_t q = 0;
        if (rocprim::inclusive_scan(nullptr, q, in, out, n, rocprim::plus<int>(), st) != hipSuccess) return XRSFM_BA_ENODEV;
        if (!need_tmp(q)) return XRSFM_BA_ENOMEM;
        size_t t = tb;
        return rocprim::inclusive_scan(d_tmp, t, in, out, n, rocprim::plus<int>(), st) == hipSuccess ? 0 : XRSFM_BA_ENODEV;
    }
    int scan_excl(int* in, int* out, size_t n) {
        size_t q = 0;
        if (rocprim::exclusive_scan(nullptr, q, in, out, 0, n, rocprim::plus<int>(), st) != hipSuccess) return XRSFM_BA_ENODEV;
        if (!need_tmp(q)) return XRSFM_BA_ENOMEM;
        size_t t = tb;
        return rocprim::exclusive_scan(d_tmp, t, in, out, 0, n, rocprim::plus<int>(), st) == hipSuccess ? 0 : XRSFM_BA_ENODEV;
    }
};

// keep(bytes) -> device memory that lives as long as the context; scratch(bytes) -> device memory the caller releases after the call
// (both nullptr when out of memory).  Returns XRSFM_BA_OK, an error, or +1: "take the host path" (a track longer than 64 observations).
template <typename Keep, typename Scratch>
inline int device_pack(const xrsfm_ba_problem& p, hipStream_t st, Keep&& keep, Scratch&& scratch_alloc, Packed& o, Result& R) {
    typedef unsigned long long u64;
    const int No = p.n_obs, Np = p.n_points, Nc = p.n_cams;
    PhaseTimer timer("devpack");
    auto tmp = [&](size_t bytes) -> void* { return scratch_alloc(bytes ? bytes : 8); };
    auto alloc = [&](size_t bytes) -> void* { return keep(bytes ? bytes : 8); };
    PrimOps<decltype(tmp)> prim{st, tmp};
    auto sort64 = [&](u64* kin, u64* kout, int* vin, int* vout, size_t n, unsigned bits) { return prim.sort_pairs(kin, kout, vin, vout, n, bits); };
    auto sort32 = [&](unsigned* kin, unsigned* kout, int* vin, int* vout, size_t n, unsigned bits) { return prim.sort_pairs(kin, kout, vin, vout, n, bits); };
    auto scan_incl = [&](int* in, int* out, size_t n) { return prim.scan_incl(in, out, n); };
    auto scan_excl = [&](int* in, int* out, size_t n) { return prim.scan_excl(in, out, n); };
    int e = 0;
#define XBA_TMP(T, name, n) T* name = static_cast<T*>(tmp(sizeof(T) * (size_t)(n))); if (!name) return XRSFM_BA_ENOMEM
#define XBA_KEEP(T, name, n) name = static_cast<T*>(alloc(sizeof(T) * (size_t)((n) > 0 ? (n) : 1))); if (!name) return XRSFM_BA_ENOMEM
#define XBA_DO(x) do { if ((e = (x))) return e; } while (0)
    int cshift = 1; while ((1ll << cshift) < (long long)Nc) ++cshift;                      // bits of a camera id
    int fbits = 1; while ((1ll << fbits) <= (long long)Nc) ++fbits;                        // bits of camera id + 1 (<= 16)
    int pt_bits = 1; while ((1ll << pt_bits) < (long long)Np) ++pt_bits;
    const u64 cmask = (1ull << cshift) - 1ull;
    const int nbo = (No + kThreads - 1) / kThreads;
    // raw inputs
    XBA_TMP(int, d_cam, No + 1); XBA_TMP(int, d_pt, No + 1); XBA_TMP(double, d_uv, 2 * (size_t)No + 2); XBA_TMP(double, d_points, 3 * (size_t)Np + 3);
    if (No > 0) {
        XBA_DP_HIP(hipMemcpyAsync(d_cam, p.obs_cam, sizeof(int) * (size_t)No, hipMemcpyHostToDevice, st));
        XBA_DP_HIP(hipMemcpyAsync(d_pt, p.obs_pt, sizeof(int) * (size_t)No, hipMemcpyHostToDevice, st));
        XBA_DP_HIP(hipMemcpyAsync(d_uv, p.obs_uv, sizeof(double) * 2 * (size_t)No, hipMemcpyHostToDevice, st));
    }
    if (Np > 0) XBA_DP_HIP(hipMemcpyAsync(d_points, p.points, sizeof(double) * 3 * (size_t)Np, hipMemcpyHostToDevice, st));
    unsigned char* d_pconst = nullptr;
    if (p.point_const && Np > 0) { XBA_TMP(unsigned char, pc_, Np); d_pconst = pc_; XBA_DP_HIP(hipMemcpyAsync(d_pconst, p.point_const, (size_t)Np, hipMemcpyHostToDevice, st)); }
    XBA_TMP(unsigned, d_status, 4);
    XBA_DP_HIP(hipMemsetAsync(d_status, 0, 16, st));
    // A: sort by (point, camera), stable in the input order
    XBA_TMP(u64, key_a, No + 1); XBA_TMP(u64, key_b, No + 1); XBA_TMP(int, val_a, No + 1); XBA_TMP(int, val_b, No + 1);
    XBA_TMP(int, head, No + 1); XBA_TMP(int, trk_incl, No + 1);
    int n_trk = 0;
    unsigned status[4] = {0, 0, 0, 0};
    if (No > 0) {
        hipLaunchKernelGGL(k_keys, dim3(nbo), dim3(kThreads), 0, st, d_cam, d_pt, No, Nc, Np, cshift, key_a, val_a, d_status);
        XBA_DO(sort64(key_a, key_b, val_a, val_b, (size_t)No, (unsigned)(cshift + pt_bits)));
    }
    u64* okey = key_b; int* oidx = val_b;          // sorted observations: (point << cshift | camera), input index
    // B: tracks
    if (No > 0) {
        hipLaunchKernelGGL(k_heads, dim3(nbo), dim3(kThreads), 0, st, okey, No, cshift, head);
        XBA_DO(scan_incl(head, trk_incl, (size_t)No));
        XBA_DP_HIP(hipMemcpyAsync(&n_trk, trk_incl + (No - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    }
    XBA_DP_HIP(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipStreamSynchronize(st));
    if (status[0] & 1u) return XRSFM_BA_EINVAL;
    timer.mark("upload + observation sort");
    const int nbt = (n_trk + kThreads - 1) / kThreads;
    XBA_TMP(int, ptr, n_trk + 1); XBA_TMP(int, pt_of, n_trk + 1); XBA_TMP(int, len, n_trk + 1);
    if (n_trk > 0) {
        hipLaunchKernelGGL(k_track_ptr, dim3(nbo), dim3(kThreads), 0, st, okey, head, trk_incl, No, cshift, ptr, pt_of);
        hipLaunchKernelGGL(k_track_len, dim3(nbt), dim3(kThreads), 0, st, ptr, n_trk, No, len, (int*)(d_status + 2), d_status);
    }
    XBA_DP_HIP(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipStreamSynchronize(st));
    if (status[0] & 2u) return 1;                         // long tracks: host path
    const int maxlen = (int)status[2];
    // C: tuple sort, least significant camera group first
    XBA_TMP(int, perm_a, n_trk + 1); XBA_TMP(int, perm_b, n_trk + 1);
    XBA_TMP(u64, tkey_a, n_trk + 1); XBA_TMP(u64, tkey_b, n_trk + 1);
    if (n_trk > 0) hipLaunchKernelGGL(k_iota, dim3(nbt), dim3(kThreads), 0, st, perm_a, n_trk);
    const int groups = (maxlen + kTupCams - 1) / kTupCams;
    for (int g = groups - 1; g >= 0 && n_trk > 0; --g) {
        hipLaunchKernelGGL(k_tuple_key, dim3(nbt), dim3(kThreads), 0, st, okey, ptr, len, perm_a, n_trk, g, cmask, fbits, tkey_a);
        XBA_DO(sort64(tkey_a, tkey_b, perm_a, perm_b, (size_t)n_trk, (unsigned)(fbits * kTupCams)));
        std::swap(perm_a, perm_b);
    }
    int* perm = perm_a;                                   // packed rank -> track (tracks are in ascending point order)
    // D: groups of equal tuples
    XBA_TMP(int, gs, n_trk + 1); XBA_TMP(int, gid, n_trk + 1); XBA_TMP(int, gstart, n_trk + 2); XBA_TMP(unsigned char, tbyte, n_trk + 1);
    std::vector<unsigned char> hbyte((size_t)n_trk);
    if (n_trk > 0) {
        hipLaunchKernelGGL(k_group_start, dim3(nbt), dim3(kThreads), 0, st, okey, ptr, len, perm, n_trk, cmask, gs);
        XBA_DO(scan_incl(gs, gid, (size_t)n_trk));
        hipLaunchKernelGGL(k_group_pos, dim3((n_trk + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, gs, gid, n_trk, gstart);
        hipLaunchKernelGGL(k_track_byte, dim3(nbt), dim3(kThreads), 0, st, len, perm, gs, gid, gstart, n_trk, tbyte);
        XBA_DP_HIP(hipMemcpyAsync(hbyte.data(), tbyte, (size_t)n_trk, hipMemcpyDeviceToHost, st));
    }
    XBA_KEEP(int, R.pt_orig, n_trk); XBA_KEEP(double, R.P, 3 * (size_t)n_trk); XBA_KEEP(unsigned char, R.pt_const, n_trk);
    o.pt_orig.resize((size_t)n_trk);
    if (n_trk > 0) {
        hipLaunchKernelGGL(k_points, dim3(nbt), dim3(kThreads), 0, st, perm, pt_of, d_points, d_pconst, n_trk, R.pt_orig, R.P, R.pt_const);
        XBA_DP_HIP(hipMemcpyAsync(o.pt_orig.data(), R.pt_orig, sizeof(int) * (size_t)n_trk, hipMemcpyDeviceToHost, st));
    }
    XBA_DP_HIP(hipStreamSynchronize(st));
    timer.mark("tracks + tuple sort");
    // E: placement (ba_pack.h: "placement"), host
    o.n_cams = Nc; o.n_pts = n_trk; o.n_obs = No;
    o.items.clear(); o.items.reserve(2 * ((size_t)No / 48 + 16));
    std::vector<int> trk_start((size_t)n_trk);
    {
        long long pos = 0;
        int cur_tile_start = -1;
        auto pad = [&]() { pos = (pos + 63) & ~63LL; };
        for (int pj = 0; pj < n_trk; ++pj) {
            const int l = hbyte[pj] & 127, big = hbyte[pj] >> 7;
            const int used = (int)(pos % 64);
            if (cur_tile_start < 0 || used + l > 64 || used == 0 || big) {
                pad();
                cur_tile_start = (int)(pos / 64);
                o.items.push_back(cur_tile_start); o.items.push_back(1);
            }
            if (pos > INT32_MAX - 128) return XRSFM_BA_EINVAL;
            trk_start[pj] = (int)pos;
            pos += l;
        }
        pad();
        if (pos > INT32_MAX) return XRSFM_BA_EINVAL;
        o.n_slots = (int)pos;
    }
    o.n_tiles = o.n_slots / 64;
    if (n_trk == 0 && o.n_slots > 0) return XRSFM_BA_EINVAL;
    o.pt_const.assign((size_t)n_trk, 0);
    o.n_var_p = 0;
    for (int pj = 0; pj < n_trk; ++pj) { const unsigned char cst = (p.point_const && p.point_const[o.pt_orig[pj]]) ? 1 : 0; o.pt_const[pj] = cst; o.n_var_p += !cst; }
    timer.mark("placement (host)");
    const int ns = o.n_slots, nt = o.n_tiles;
    const int nbs = (ns + kThreads - 1) / kThreads, nbw = (nt + 3) / 4;
    // F: slots
    XBA_TMP(int, d_trk_start, n_trk + 1); XBA_TMP(int, inv, n_trk + 1);
    if (n_trk > 0) XBA_DP_HIP(hipMemcpyAsync(d_trk_start, trk_start.data(), sizeof(int) * (size_t)n_trk, hipMemcpyHostToDevice, st));
    XBA_KEEP(int, R.slot_cam, ns); XBA_KEEP(int, R.slot_pt, ns); XBA_KEEP(int, R.slot_obs, ns); XBA_KEEP(double, R.slot_u, ns); XBA_KEEP(double, R.slot_v, ns);
    if (ns > 0) {
        XBA_DP_HIP(hipMemsetAsync(R.slot_cam, 0xff, sizeof(int) * (size_t)ns, st)); XBA_DP_HIP(hipMemsetAsync(R.slot_pt, 0xff, sizeof(int) * (size_t)ns, st));
        XBA_DP_HIP(hipMemsetAsync(R.slot_obs, 0xff, sizeof(int) * (size_t)ns, st));
        XBA_DP_HIP(hipMemsetAsync(R.slot_u, 0, sizeof(double) * (size_t)ns, st)); XBA_DP_HIP(hipMemsetAsync(R.slot_v, 0, sizeof(double) * (size_t)ns, st));
    }
    if (n_trk > 0) {
        hipLaunchKernelGGL(k_inv_perm, dim3(nbt), dim3(kThreads), 0, st, perm, n_trk, inv);
        hipLaunchKernelGGL(k_fill_slots, dim3(nbo), dim3(kThreads), 0, st, okey, oidx, trk_incl, ptr, inv, d_trk_start, d_uv, No, cmask, R.slot_cam, R.slot_pt, R.slot_obs,
                           R.slot_u, R.slot_v);
    }
    // G: tiles
    XBA_KEEP(int, R.tile_maxlen, nt); XBA_KEEP(int, R.tile_stride, nt); XBA_KEEP(int, R.tile_ncam, nt); XBA_KEEP(int, R.tile_gt_off, nt);
    XBA_KEEP(unsigned char, R.slot_cidx, ns);
    XBA_TMP(int, cells2, nt + 1); XBA_TMP(int, gt_scan, nt + 1); XBA_TMP(int, tile_big, nt + 1); XBA_TMP(int, big_scan, nt + 1);
    int tail[2] = {0, 0}, n_big = 0;
    if (nt > 0) {
        hipLaunchKernelGGL(k_tiles, dim3(nbw), dim3(kThreads), 0, st, R.slot_cam, R.slot_pt, nt, kGramMaxCams, 6, R.tile_maxlen, R.tile_stride, R.tile_ncam, R.slot_cidx,
                           cells2, tile_big);
        XBA_DO(scan_excl(cells2, gt_scan, (size_t)nt));
        XBA_DO(scan_incl(tile_big, big_scan, (size_t)nt));
        XBA_DP_HIP(hipMemcpyAsync(&tail[0], gt_scan + (nt - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        XBA_DP_HIP(hipMemcpyAsync(&tail[1], cells2 + (nt - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        XBA_DP_HIP(hipMemcpyAsync(&n_big, big_scan + (nt - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        XBA_DP_HIP(hipStreamSynchronize(st));
    }
    o.n_gt_cells = tail[0] + tail[1];
    XBA_KEEP(unsigned char, R.gt_cell, o.n_gt_cells);
    if (o.n_gt_cells > 0) XBA_DP_HIP(hipMemsetAsync(R.gt_cell, 0, (size_t)o.n_gt_cells, st));
    XBA_TMP(unsigned, key_all, ns + 1); XBA_TMP(unsigned, key_gram, ns + 1); XBA_TMP(unsigned, skey, ns + 1); XBA_TMP(int, sl_a, ns + 1); XBA_TMP(int, sl_b, ns + 1);
    XBA_KEEP(int, R.slot_campos, ns); XBA_KEEP(int, R.slot_campos_g, ns); XBA_KEEP(int, R.cam_ptr, Nc + 1); XBA_KEEP(int, R.cam_ptr_g, Nc + 1);
    XBA_TMP(int, cnt, Nc + 2);
    const int demote = (n_big > 0 && (long long)n_big * 20 <= (long long)nt) ? 1 : 0;
    if (nt > 0) {
        hipLaunchKernelGGL(k_gram_cells, dim3(nbw), dim3(kThreads), 0, st, R.slot_cam, R.slot_pt, R.slot_cidx, R.tile_ncam, gt_scan, nt, R.tile_gt_off, R.gt_cell);
        hipLaunchKernelGGL(k_demote_flags, dim3(nbw), dim3(kThreads), 0, st, R.slot_cam, nt, demote, tile_big, R.tile_stride, R.tile_ncam, R.tile_gt_off, R.slot_cidx, Nc,
                           key_all, key_gram);
    }
    // H: camera-major positions (stable sort of the writers by camera)
    int cbits = 1; while ((1ll << cbits) <= (long long)Nc) ++cbits;          // keys 0 .. Nc
    o.cam_ptr.assign((size_t)Nc + 1, 0); o.cam_ptr_g.assign((size_t)Nc + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
        unsigned* key = pass == 0 ? key_all : key_gram;
        int* campos = pass == 0 ? R.slot_campos : R.slot_campos_g;
        int* cam_ptr = pass == 0 ? R.cam_ptr : R.cam_ptr_g;
        XBA_DP_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * ((size_t)Nc + 2), st));
        if (ns > 0) {
            hipLaunchKernelGGL(k_iota, dim3(nbs), dim3(kThreads), 0, st, sl_a, ns);
            hipLaunchKernelGGL(k_count_cams, dim3(nbs), dim3(kThreads), 0, st, key, ns, Nc, cnt);
            XBA_DO(sort32(key, skey, sl_a, sl_b, (size_t)ns, (unsigned)cbits));
            hipLaunchKernelGGL(k_campos, dim3(nbs), dim3(kThreads), 0, st, skey, sl_b, ns, Nc, campos);
        }
        XBA_DO(scan_excl(cnt, cam_ptr, (size_t)Nc + 1));
        XBA_DP_HIP(hipMemcpyAsync((pass == 0 ? o.cam_ptr : o.cam_ptr_g).data(), cam_ptr, sizeof(int) * ((size_t)Nc + 1), hipMemcpyDeviceToHost, st));
    }
    XBA_DP_HIP(hipStreamSynchronize(st));
    XBA_DP_HIP(hipGetLastError());
    o.n_cam_entries = o.cam_ptr[Nc]; o.n_cam_entries_g = o.cam_ptr_g[Nc];
    // counts (ba_pack.h: "counts")
    o.n_var_q = o.n_var_t = 0;
    for (int c = 0; c < Nc; ++c) {
        if (o.cam_ptr[c + 1] <= o.cam_ptr[c]) continue;
        const unsigned cc = p.cam_const ? p.cam_const[c] : 0u;
        if (!(cc & XRSFM_BA_CONST_Q)) o.n_var_q++;
        if (!(cc & XRSFM_BA_CONST_T)) o.n_var_t++;
    }
    timer.mark("slots + tiles + camera-major");
#undef XBA_TMP
#undef XBA_KEEP
#undef XBA_DO
    return XRSFM_BA_OK;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Pair keys of the Cholesky path on the device (ba_plan.h: chol_local_keys + the "blocks + destinations" part of the plan):
// slot_pair_ptr, the sorted (block key, writer index) list -> blocks (row, column camera), blk_ptr, pair_dst, and the launch
// buckets of the S assembly (pairs_items).  The host plan then only needs the block list (a few thousand pairs).
// Single rank, local pattern; every tile is a single-tile item (device-packed contexts have no long tracks).

// per slot: pairs that start at it (later slots of the same track); duplicate-camera check; per-tile key counts
__global__ __launch_bounds__(256) void k_pair_counts(const int* __restrict__ slot_cam, const int* __restrict__ slot_pt, const int* __restrict__ tile_ncam,
                                                     const int* __restrict__ tile_gt_off, const unsigned char* __restrict__ gt_cell, int n_tiles,
                                                     int* __restrict__ cnt, int* __restrict__ nk_slot, int* __restrict__ nk_tile, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int s = 64 * t + lane;
    const int cam = slot_cam[s], pt = slot_pt[s];
    const bool valid = cam >= 0;
    const int next_pt = __shfl_down(pt, 1, 64), next_cam = __shfl_down(cam, 1, 64);
    const bool last = !valid || lane == 63 || next_cam < 0 || next_pt != pt;          // last slot of its track
    if (valid && !last && next_cam <= cam) atomicOr(status, 4u);                        // two observations of one track in the same frame
    const unsigned long long lm = __ballot(last && valid);
    // slots until the end of the own track: position of the first "last" flag at or after this lane
    const unsigned long long from = lm >> lane;
    const int run = (valid && from) ? __ffsll((long long)from) - 1 : 0;
    cnt[s] = run;
    const int C = tile_ncam[t];
    nk_slot[s] = (C > 0) ? 0 : run;
    int cells = 0;
    if (C > 0) {
        const unsigned char* cell = gt_cell + tile_gt_off[t];
        for (int e = lane; e < C * C; e += 64) { const int a = e / C, b = e - a * C; if (b > a && cell[e]) ++cells; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cells += __shfl_xor(cells, off, 64);
    }
    if (lane == 0) nk_tile[t] = cells;
}
__global__ __launch_bounds__(256) void k_pair_keys(const int* __restrict__ slot_cam, const unsigned char* __restrict__ slot_cidx, const int* __restrict__ tile_ncam,
                                                   const int* __restrict__ tile_gt_off, const unsigned char* __restrict__ gt_cell, int n_tiles,
                                                   const int* __restrict__ cnt, const int* __restrict__ spp, const int* __restrict__ koff, const int* __restrict__ goff,
                                                   int k1_total, int n_obs_pairs, int cshift, unsigned long long* __restrict__ key, int* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const int s = 64 * t + lane;
    const int cam = slot_cam[s];
    const int C = tile_ncam[t];
    if (C <= 0) {
        const int n = cnt[s];
        for (int dd = 1; dd <= n; ++dd) {
            const int cb = slot_cam[s + dd];
            key[koff[s] + dd - 1] = ((unsigned long long)(unsigned)cb << cshift) | (unsigned)cam;
            val[koff[s] + dd - 1] = spp[s] + dd - 1;
        }
        return;
    }
    // ascending distinct cameras of the tile: cams[cidx] = camera
    __shared__ int cams_s[4][16];
    int* cams = cams_s[threadIdx.x >> 6];
    if (cam >= 0) cams[slot_cidx[s]] = cam;           // (equal values from every lane of a camera)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const unsigned char* cell = gt_cell + tile_gt_off[t];
    int base = k1_total + goff[t];
    for (int e0 = 0; e0 < C * C; e0 += 64) {
        const int e = e0 + lane;
        const int a = e / C, b = e - a * C;
        const bool on = e < C * C && b > a && cell[e];
        const unsigned long long m = __ballot(on);
        if (on) {
            const int r = base + __popcll(m & ((1ull << lane) - 1ull));
            key[r] = ((unsigned long long)(unsigned)cams[b] << cshift) | (unsigned)cams[a];
            val[r] = n_obs_pairs + tile_gt_off[t] + e;
        }
        base += __popcll(m);
    }
}
__global__ void k_key_heads(const unsigned long long* __restrict__ key, int n, int* __restrict__ head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
__global__ void k_blocks(const unsigned long long* __restrict__ key, const int* __restrict__ val, const int* __restrict__ head, const int* __restrict__ bid_incl,
                         int n, int cshift, int* __restrict__ blk_ptr, int* __restrict__ blk_rc, int* __restrict__ pair_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { blk_ptr[n > 0 ? bid_incl[n - 1] : 0] = n; return; }
    pair_dst[val[i]] = i;
    if (head[i]) {
        const int b = bid_incl[i] - 1;
        blk_ptr[b] = i;
        blk_rc[2 * b] = (int)(key[i] >> cshift); blk_rc[2 * b + 1] = (int)(key[i] & ((1ull << cshift) - 1ull));
    }
}
// launch bucket of every tile (ba_plan.h: 2 * (NI - 1) + LDS class for a Gram tile, 8 = per-pair path) + per-bucket LDS need
__global__ __launch_bounds__(256) void k_buckets(const int* __restrict__ slot_cam, const int* __restrict__ slot_pt, const int* __restrict__ tile_ncam, int n_tiles,
                                                 int cw, unsigned* __restrict__ bkey, int* __restrict__ bcount, int* __restrict__ bshm) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int s = 64 * min(t, n_tiles - 1) + lane;
    const int cam = slot_cam[s], pt = slot_pt[s];
    const int prev = __shfl_up(pt, 1, 64);
    const int ntrk = __popcll(__ballot(cam >= 0 && (lane == 0 || prev != pt)));
    // counters privatised per workgroup (tens of thousands of atomics on one word cost 370 us at config 4)
    __shared__ int s_cnt[9], s_shm[8];
    if (threadIdx.x < 9) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < 8) s_shm[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0 && t < n_tiles) {
        const int C = tile_ncam[t];
        int b = 8;
        if (C > 0) {
            int passes = 1;
            const int base = 64 * 15 * 8;
            const int need = max(base, gram_lds_need(C, ntrk, &passes, cw));
            b = 2 * ((cw * C + 15) / 16 - 1) + (need <= kGramSmallLds ? 0 : 1);
            atomicMax(&s_shm[b], need);
        }
        bkey[t] = (unsigned)b;
        atomicAdd(&s_cnt[b], 1);
    }
    __syncthreads();
    if (threadIdx.x < 9 && s_cnt[threadIdx.x] > 0) atomicAdd(bcount + threadIdx.x, s_cnt[threadIdx.x]);
    if (threadIdx.x < 8 && s_shm[threadIdx.x] > 0) atomicMax(bshm + threadIdx.x, s_shm[threadIdx.x]);
}

struct KeysResult {
    int *spp = nullptr, *pair_dst = nullptr, *blk_ptr = nullptr, *blk_rc = nullptr, *pairs_items = nullptr;     // device (kept)
    int n_pairs = 0, n_writes = 0, n_pair_writes = 0, n_blocks = 0;        // n_pair_writes: entries that are per-pair blocks (the others: Gram cells)
    std::vector<int> blk_rc_host;
    int gram_n[8] = {0}, n_other = 0; size_t gram_shm[8] = {0};
    bool duplicate = false;
};

template <typename Keep, typename Scratch>
inline int device_keys(const Packed& o, const int* slot_cam, const int* slot_pt, const unsigned char* slot_cidx, const int* tile_ncam, const int* tile_gt_off,
                       const unsigned char* gt_cell, hipStream_t st, Keep&& keep, Scratch&& scratch_alloc, KeysResult& K) {
    typedef unsigned long long u64;
    const int ns = o.n_slots, nt = o.n_tiles, Nc = o.n_cams;
    PhaseTimer timer("devkeys");
    auto tmp = [&](size_t bytes) -> void* { return scratch_alloc(bytes ? bytes : 8); };
    auto alloc = [&](size_t bytes) -> void* { return keep(bytes ? bytes : 8); };
    PrimOps<decltype(tmp)> prim{st, tmp};
    auto sort64 = [&](u64* kin, u64* kout, int* vin, int* vout, size_t n, unsigned bits) { return prim.sort_pairs(kin, kout, vin, vout, n, bits); };
    auto sort32 = [&](unsigned* kin, unsigned* kout, int* vin, int* vout, size_t n, unsigned bits) { return prim.sort_pairs(kin, kout, vin, vout, n, bits); };
    auto scan_incl = [&](int* in, int* out, size_t n) { return prim.scan_incl(in, out, n); };
    auto scan_excl = [&](int* in, int* out, size_t n) { return prim.scan_excl(in, out, n); };
    int e = 0;
#define XBA_TMP(T, name, n) T* name = static_cast<T*>(tmp(sizeof(T) * (size_t)(n))); if (!name) return XRSFM_BA_ENOMEM
#define XBA_KEEP(T, name, n) name = static_cast<T*>(alloc(sizeof(T) * (size_t)((n) > 0 ? (n) : 1))); if (!name) return XRSFM_BA_ENOMEM
#define XBA_DO(x) do { if ((e = (x))) return e; } while (0)
    int cshift = 1; while ((1ll << cshift) < (long long)Nc) ++cshift;
    const int nbw = (nt + 3) / 4;
    XBA_TMP(unsigned, d_status, 4);
    XBA_DP_HIP(hipMemsetAsync(d_status, 0, 16, st));
    XBA_TMP(int, cnt, ns + 2); XBA_TMP(int, nk_slot, ns + 2); XBA_TMP(int, nk_tile, nt + 2); XBA_TMP(int, koff, ns + 2); XBA_TMP(int, goff, nt + 2);
    XBA_KEEP(int, K.spp, ns + 1);
    XBA_DP_HIP(hipMemsetAsync(cnt + ns, 0, sizeof(int) * 2, st)); XBA_DP_HIP(hipMemsetAsync(nk_slot + ns, 0, sizeof(int) * 2, st));
    XBA_DP_HIP(hipMemsetAsync(nk_tile + nt, 0, sizeof(int) * 2, st));
    if (nt > 0) hipLaunchKernelGGL(k_pair_counts, dim3(nbw), dim3(kThreads), 0, st, slot_cam, slot_pt, tile_ncam, tile_gt_off, gt_cell, nt, cnt, nk_slot, nk_tile, d_status);
    XBA_DO(scan_excl(cnt, K.spp, (size_t)ns + 1));              // spp[ns] = number of observation pairs
    XBA_DO(scan_excl(nk_slot, koff, (size_t)ns + 1));
    XBA_DO(scan_excl(nk_tile, goff, (size_t)nt + 1));
    int totals[3] = {0, 0, 0};
    unsigned status[4] = {0, 0, 0, 0};
    XBA_DP_HIP(hipMemcpyAsync(&totals[0], K.spp + ns, sizeof(int), hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipMemcpyAsync(&totals[1], koff + ns, sizeof(int), hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipMemcpyAsync(&totals[2], goff + nt, sizeof(int), hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipMemcpyAsync(status, d_status, 16, hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipStreamSynchronize(st));
    if (status[0] & 4u) { K.duplicate = true; return 0; }
    const int n_obs_pairs = totals[0], k1 = totals[1], nw = totals[1] + totals[2];
    if ((long long)n_obs_pairs + o.n_gt_cells > INT32_MAX) return XRSFM_BA_EINVAL;
    K.n_pairs = n_obs_pairs + o.n_gt_cells; K.n_writes = nw; K.n_pair_writes = k1;
    XBA_TMP(u64, key_a, nw + 1); XBA_TMP(u64, key_b, nw + 1); XBA_TMP(int, val_a, nw + 1); XBA_TMP(int, val_b, nw + 1);
    XBA_TMP(int, head, nw + 1); XBA_TMP(int, bid, nw + 1);
    if (nt > 0) hipLaunchKernelGGL(k_pair_keys, dim3(nbw), dim3(kThreads), 0, st, slot_cam, slot_cidx, tile_ncam, tile_gt_off, gt_cell, nt, cnt, K.spp, koff, goff, k1,
                                   n_obs_pairs, cshift, key_a, val_a);
    int n_blocks = 0;
    if (nw > 0) {
        XBA_DO(sort64(key_a, key_b, val_a, val_b, (size_t)nw, (unsigned)(2 * cshift)));
        hipLaunchKernelGGL(k_key_heads, dim3((nw + kThreads - 1) / kThreads), dim3(kThreads), 0, st, key_b, nw, head);
        XBA_DO(scan_incl(head, bid, (size_t)nw));
        XBA_DP_HIP(hipMemcpyAsync(&n_blocks, bid + (nw - 1), sizeof(int), hipMemcpyDeviceToHost, st));
        XBA_DP_HIP(hipStreamSynchronize(st));
    }
    K.n_blocks = n_blocks;
    XBA_KEEP(int, K.pair_dst, K.n_pairs); XBA_KEEP(int, K.blk_ptr, n_blocks + 1); XBA_KEEP(int, K.blk_rc, 2 * n_blocks);
    if (K.n_pairs > 0) XBA_DP_HIP(hipMemsetAsync(K.pair_dst, 0xff, sizeof(int) * (size_t)K.n_pairs, st));
    hipLaunchKernelGGL(k_blocks, dim3((nw + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, st, key_b, val_b, head, bid, nw, cshift, K.blk_ptr, K.blk_rc, K.pair_dst);
    K.blk_rc_host.resize(2 * (size_t)n_blocks);
    if (n_blocks > 0) XBA_DP_HIP(hipMemcpyAsync(K.blk_rc_host.data(), K.blk_rc, sizeof(int) * 2 * (size_t)n_blocks, hipMemcpyDeviceToHost, st));
    // launch buckets of the S assembly
    XBA_TMP(unsigned, bkey, nt + 1); XBA_TMP(unsigned, bkey_s, nt + 1); XBA_TMP(int, tl_a, nt + 1); XBA_TMP(int, bcount, 32);
    XBA_KEEP(int, K.pairs_items, nt);
    XBA_DP_HIP(hipMemsetAsync(bcount, 0, sizeof(int) * 32, st));
    int hb[32] = {0};
    if (nt > 0) {
        hipLaunchKernelGGL(k_buckets, dim3(nbw), dim3(kThreads), 0, st, slot_cam, slot_pt, tile_ncam, nt, 6, bkey, bcount, bcount + 16);
        hipLaunchKernelGGL(k_iota, dim3((nt + kThreads - 1) / kThreads), dim3(kThreads), 0, st, tl_a, nt);
        XBA_DO(sort32(bkey, bkey_s, tl_a, K.pairs_items, (size_t)nt, 4u));
    }
    XBA_DP_HIP(hipMemcpyAsync(hb, bcount, sizeof(int) * 32, hipMemcpyDeviceToHost, st));
    XBA_DP_HIP(hipStreamSynchronize(st));
    XBA_DP_HIP(hipGetLastError());
    const size_t base = (size_t)64 * 15 * sizeof(double);
    for (int b = 0; b < 8; ++b) { K.gram_n[b] = hb[b]; K.gram_shm[b] = std::max(base, (size_t)hb[16 + b]); }
    K.n_other = hb[8];
    timer.mark("pair keys + blocks + buckets");
#undef XBA_TMP
#undef XBA_KEEP
#undef XBA_DO
    return XRSFM_BA_OK;
}

}  // namespace devpack
}  // namespace xba
