// Host-side packing: caller SoA (include/xrsfm_ba.h) -> track-major 64-slot tiles
// + camera-major positions.  Replaces the reference's per-observation
// problem.AddResidualBlock loop (/root/reference/src/optimization/ba_solver.cc:336-349).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <numeric>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/xrsfm_ba.h"

namespace xba {

constexpr int kGramMaxCams = 10;         // 60 operand rows: 4 MFMA row tiles
constexpr int kGramMaxLds = 20 * 1024;   // upper bound of a tile's staged operand (bytes; since round 6 every Gram tile fits kGramSmallLds in enough passes)
#ifndef XBA_GRAM_PAD
#define XBA_GRAM_PAD 1
#endif
constexpr int kGramPad = XBA_GRAM_PAD;    // padding columns of a staged operand row (LDS bank spread)
constexpr int kGramSmallLds = 10240;     // LDS class boundary of the S-assembly launches: 160 KB / 16 workgroups

// LDS bytes of a Gram tile with C cameras and T tracks: operand [6C][3T'+pad] + the C x C destination table, where the tracks
// are staged in ONE pass (T' = T) if that fits the small class and otherwise in the smallest number of passes of T' = ceil(T/p)
// tracks that does (round 6; until round 5: at most two, and what still did not fit formed a second LDS class with a launch of its
// own).  The accumulators are kept across the passes — a pass boundary only inserts zero columns, so the blocks do not depend on
// the pass count by a bit.  The kernel (k_schur_pairs) evaluates the same rule.  Every tile fits: the largest operand of one
// track per pass is 63 rows x 5 columns (bal9 mode) = 2.5 KB.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int gram_lds_need(int C, int T, int* passes, int cw = 6) {     // cw: operand rows per camera (9 in bal9 mode, ba_wide.h)
    // destination table with a fixed row stride (ba_chol.h: kGramTabLd) + the schedule of 4x4 result blocks of a small tile (ba_chol.h:
    // gram_tile4, at most 6 instructions x 4 blocks x 2 bytes)
    constexpr int tab = (kGramMaxCams + 1) * (kGramMaxCams + 1) * 4 + 48;
    int p = 1, need = 0;
    for (;; ++p) {
        const int Th = (T + p - 1) / p;
        need = cw * C * ((((3 * Th + 3) & ~3)) + kGramPad) * 8 + tab;
        if (need <= kGramSmallLds || Th <= 1) break;        // (the smallest p of a given T' = ceil(T / p): the last pass is never empty)
    }
    *passes = p;
    return need;
}
constexpr int kGramMaxCamsWide = 7;      // bal9 mode: 7 cameras x 9 rows = 63 operand rows

// Host-side helper: run fn(begin, end) over [0, n) on up to 8 threads (16 for the largest loops) (large problems only; the packing of config L is ~100 ms
// of single-thread work otherwise).  The pieces are disjoint, so the result does not depend on the thread count.
template <typename F>
inline void pack_parallel_for(long long n, F&& fn, long long min_n = 200000) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (n < min_n || hw < 2) ? 1 : (int)std::min<unsigned>(n >= 5 * min_n ? 16u : 8u, hw);
    if (nt == 1) { fn(0LL, n); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nt);      // an exception in a worker (std::bad_alloc) is rethrown by the caller's thread
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { try { fn(n * t / nt, n * (t + 1) / nt); } catch (...) { err[t] = std::current_exception(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

// XRSFM_BA_PACK_TIMING=1: phase times of the host-side set-up (packing, Cholesky plan) on stderr (developer aid)
struct PhaseTimer {
    const char* tag;
    bool on;
    std::chrono::steady_clock::time_point prev;
    explicit PhaseTimer(const char* t) : tag(t), on(std::getenv("XRSFM_BA_PACK_TIMING") != nullptr), prev(std::chrono::steady_clock::now()) {}
    void mark(const char* what) {
        if (!on) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[%s] %-28s %7.2f ms\n", tag, what, 1e3 * std::chrono::duration<double>(t - prev).count());
        prev = t;
    }
};

// Vector whose resize() leaves new elements uninitialised: the large per-slot arrays are written exactly once, in parallel,
// so the pages are first touched by the threads that fill them instead of being zeroed by one thread beforehand.
template <typename T>
struct RawAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = RawAlloc<U>; };
    RawAlloc() = default;
    template <typename U> RawAlloc(const RawAlloc<U>&) {}
    template <typename U> void construct(U* q) noexcept { ::new ((void*)q) U; }
    template <typename U, typename A0, typename... A> void construct(U* q, A0&& a0, A&&... a) { ::new ((void*)q) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
template <typename T> using RawVec = std::vector<T, RawAlloc<T>>;

// Fixed partition of [0, n) into up to 8 pieces whose boundaries are multiples of `align` (one piece for small n), and a runner
// that hands every piece its index: for two-pass algorithms (count, then place) that need the same partition twice.
inline std::vector<long long> pack_cuts(long long n, long long min_n, long long align) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (n < min_n || hw < 2) ? 1 : (int)std::min<unsigned>(n >= 4 * min_n ? 16u : 8u, hw);      // (the result never depends on the piece count)
    std::vector<long long> cut(nt + 1);
    for (int t = 0; t <= nt; ++t) cut[t] = (t == nt) ? n : (n * t / nt) / align * align;
    return cut;
}
template <typename F>
inline void pack_parallel_chunks(const std::vector<long long>& cut, F&& fn) {
    const int nc = (int)cut.size() - 1;
    if (nc == 1) { fn(0, cut[0], cut[1]); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(nc);
    for (int t = 0; t < nc; ++t) th.emplace_back([&, t] { try { fn(t, cut[t], cut[t + 1]); } catch (...) { err[t] = std::current_exception(); } });
    for (auto& x : th) x.join();
    for (auto& e : err) if (e) std::rethrow_exception(e);
}

struct Packed {
    int n_cams = 0, n_pts = 0, n_obs = 0, n_tiles = 0, n_slots = 0;
    std::vector<int> pt_orig;        // packed point -> caller point index
    RawVec<int> slot_cam, slot_pt, slot_obs, slot_campos;   // slot_obs: caller obs index (-1 pad)
    RawVec<double> slot_u, slot_v;
    std::vector<int> items;          // pairs {first_tile, n_tiles}
    std::vector<int> cam_ptr;        // [n_cams+1] into the camera-major scatter buffer (entries, not observations)
    std::vector<int> tile_maxlen;    // [n_tiles] longest track in the tile (<= 64)
    std::vector<int> tile_stride;    // [n_tiles] L > 0: every track of the tile has the same L cameras ("regular" tile)
    int n_cam_entries = 0;
    // The same for the S assembly (k_schur_pairs -> k_chol_segsum): a Gram tile sums the diagonal-block / rhs terms of every
    // camera over its tracks first, so only the first lane of each distinct camera writes an entry.
    std::vector<int> cam_ptr_g;
    RawVec<int> slot_campos_g;
    int n_cam_entries_g = 0;
    // "Gram tiles" (S assembly): a single tile whose tracks see at most kGramMaxCams distinct cameras.  Its camera-pair blocks
    // come out of ONE Gram product V V^T, V = [6 x distinct camera] x [3 x track] (zero where a track does not see a camera),
    // already summed over the tracks; a regular tile is the dense special case.
    std::vector<int> tile_ncam;          // [n_tiles] distinct cameras C (0: not a Gram tile)
    std::vector<int> tile_gt_off;        // [n_tiles] offset of the tile's C x C destination table (cell (a,b), a < b: block of the pair)
    std::vector<unsigned char> slot_cidx;// [n_slots] index of the slot's camera in the tile's ascending camera list
    std::vector<unsigned char> gt_cell;  // [n_gt_cells] 1: some track of the tile sees both cameras of the cell
    int n_gt_cells = 0;
    std::vector<unsigned char> pt_const;
    int n_var_q = 0, n_var_t = 0, n_var_p = 0;
};

// Cameras whose intrinsics block is VARIABLE (cam_const bit 2, value 4: bal9 mode — 9-wide camera blocks; the reference always
// holds intrinsics constant, ba_solver.cc:602-606): only the extension model 5 {f, k1, k2}, one intrinsics entry per such camera.
constexpr unsigned kCamIntrVariable = 4u;
constexpr int kModelBal = 5;

// wide (bal9 mode): no regular-tile pre-reductions (the 9-wide linearisation / back-substitution work per observation); Gram
// tiles of up to kGramMaxCamsWide cameras (round 4: k9_pairs_gram — the S assembly of 9-wide blocks through the same Gram
// product and per-camera pre-reduction as the 6-wide path).
// argument checks of a problem (everything but the observation indices, which the packing checks as it reads them)
inline int pack_validate(const xrsfm_ba_problem& p, bool wide) {
    if (p.n_cams < 0 || p.n_points < 0 || p.n_obs < 0 || p.n_intr < 0) return XRSFM_BA_EINVAL;
    if (p.n_obs > 0 && (!p.obs_cam || !p.obs_pt || !p.obs_uv)) return XRSFM_BA_EINVAL;
    if (p.n_cams > 0 && (!p.cam_q || !p.cam_t || !p.cam_intr)) return XRSFM_BA_EINVAL;
    if (p.n_points > 0 && !p.points) return XRSFM_BA_EINVAL;
    if (p.n_intr > 0 && (!p.intr_model || !p.intr_params)) return XRSFM_BA_EINVAL;
    for (int c = 0; c < p.n_cams; ++c) {
        const int ii = p.cam_intr[c];
        if (ii < 0 || ii >= p.n_intr) return XRSFM_BA_EINVAL;
        const int m = p.intr_model[ii];
        if (m < 0 || m > kModelBal) return XRSFM_BA_EINVAL;
        if (p.cam_const && (p.cam_const[c] & kCamIntrVariable) && m != kModelBal) return XRSFM_BA_EINVAL;
    }
    if (wide) {      // a variable intrinsics entry belongs to exactly one camera
        std::vector<int> users(p.n_intr, 0);
        for (int c = 0; c < p.n_cams; ++c) users[p.cam_intr[c]]++;
        for (int c = 0; c < p.n_cams; ++c)
            if (p.cam_const && (p.cam_const[c] & kCamIntrVariable) && users[p.cam_intr[c]] != 1) return XRSFM_BA_EINVAL;
    }
    return XRSFM_BA_OK;
}

inline int pack_problem(const xrsfm_ba_problem& p, Packed& o, bool wide = false) {
    if (int ev = pack_validate(p, wide)) return ev;
    const int No = p.n_obs, Np = p.n_points, Nc = p.n_cams;
    PhaseTimer timer("pack");
    auto mark = [&](const char* what) { timer.mark(what); };
    // CSR by caller point (observation indices of every track, in input order): a stable counting sort.  Large inputs in
    // parallel: every piece of the observation list counts per point, a pass over points x pieces turns the counts into the
    // pieces' start positions inside each track, every piece places its observations.
    std::vector<int> cnt(Np + 1, 0), ptr(Np + 1, 0);
    RawVec<int> csr(No);
    const std::vector<long long> ocut = pack_cuts(No, 1000000, 1);
    const int och = (int)ocut.size() - 1;
    if (och > 1 && (size_t)och * (size_t)Np <= ((size_t)64 << 20)) {
        std::vector<std::vector<int>> lc(och);
        std::vector<char> bad(och, 0);
        pack_parallel_chunks(ocut, [&](int t, long long i0, long long i1) {
            lc[t].assign(Np, 0);
            for (long long i = i0; i < i1; ++i) {
                const int c = p.obs_cam[i], j = p.obs_pt[i];
                if (c < 0 || c >= Nc || j < 0 || j >= Np) { bad[t] = 1; return; }
                lc[t][j]++;
            }
        });
        for (int t = 0; t < och; ++t) if (bad[t]) return XRSFM_BA_EINVAL;
        pack_parallel_for(Np, [&](long long j0, long long j1) {
            for (long long j = j0; j < j1; ++j) {
                int run = 0;
                for (int t = 0; t < och; ++t) { const int v = lc[t][j]; lc[t][j] = run; run += v; }
                cnt[j + 1] = run;
            }
        }, 100000);
        for (int j = 0; j < Np; ++j) ptr[j + 1] = ptr[j] + cnt[j + 1];
        pack_parallel_chunks(ocut, [&](int t, long long i0, long long i1) {
            for (long long i = i0; i < i1; ++i) { const int j = p.obs_pt[i]; csr[ptr[j] + lc[t][j]++] = (int)i; }
        });
    } else {
        for (int i = 0; i < No; ++i) {
            const int c = p.obs_cam[i], j = p.obs_pt[i];
            if (c < 0 || c >= Nc || j < 0 || j >= Np) return XRSFM_BA_EINVAL;
            cnt[j + 1]++;
        }
        for (int j = 0; j < Np; ++j) ptr[j + 1] = ptr[j] + cnt[j + 1];
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (int i = 0; i < No; ++i) csr[fill[p.obs_pt[i]]++] = i;
    }
    mark("csr by point");
    // observations of every track ordered by camera
    pack_parallel_for(Np, [&](long long j0, long long j1) {
        for (long long j = j0; j < j1; ++j) {
            int* b = csr.data() + ptr[j];
            const int len = ptr[j + 1] - ptr[j];
            if (len <= 32) {                        // stable insertion sort: frame-major input is already in order
                for (int x = 1; x < len; ++x) {
                    const int v = b[x], cv = p.obs_cam[v];
                    int y = x - 1;
                    while (y >= 0 && p.obs_cam[b[y]] > cv) { b[y + 1] = b[y]; --y; }
                    b[y + 1] = v;
                }
            } else {
                std::stable_sort(b, b + len, [&](int a, int c) { return p.obs_cam[a] < p.obs_cam[c]; });
            }
        }
    });
    // cameras of every track in track order (one sequential array: the tuple keys, the tuple comparisons and the slot fill read it
    // instead of chasing obs_cam through the CSR, three dependent random reads per element)
    RawVec<int> tcam(No);
    pack_parallel_for(No, [&](long long i0, long long i1) { for (long long i = i0; i < i1; ++i) tcam[i] = p.obs_cam[csr[i]]; }, 200000);
    mark("per-track camera sort");
    // active points: short tracks sorted by their camera tuple (tracks seeing the same cameras become neighbours:
    // locality of the camera gathers, and whole tiles that share one tuple can be pre-reduced in the wave),
    // long tracks (> 64 obs) at the end
    std::vector<int> order;
    order.reserve(Np);
    for (int j = 0; j < Np; ++j)
        if (cnt[j + 1] > 0) order.push_back(j);
    auto tuple_less = [&](int a, int b) {
        const int la = cnt[a + 1], lb = cnt[b + 1];
        const int n = std::min(la, lb);
        for (int k = 0; k < n; ++k) {
            const int ca = tcam[ptr[a] + k], cb = tcam[ptr[b] + k];
            if (ca != cb) return ca < cb;
        }
        return la < lb;
    };
    std::vector<unsigned long long> sort_keys;      // keys of `order` after the sort (empty: comparator sort)
    int key_cams = 0;
    {
        // The comparator walks two tuples through three indirections per element; sorting 64-bit keys built from the
        // first cameras (4 x 15 bits, or 3 x 20 bits for more than 32k cameras; +1 so that a shorter tuple sorts before its
        // extensions; long tracks last) and falling back to the comparator only inside groups of equal keys whose tuples
        // are longer than the key is several times faster on a million tracks.
        struct KI { unsigned long long key; int idx; };
        std::vector<KI> ki(order.size());
        const int kbits = (Nc + 1 < (1 << 15)) ? 15 : 20, kcams = (kbits == 15) ? 4 : 3;
        const bool wide = Nc + 1 >= (1 << 20);         // camera ids do not fit the key: plain comparator sort
        pack_parallel_for((long long)order.size(), [&](long long n0, long long n1) {
            for (long long n = n0; n < n1; ++n) {
                const int j = order[n];
                const int len = cnt[j + 1];
                unsigned long long key = (len > 64) ? (1ull << 63) : 0ull;
                for (int q = 0; q < kcams; ++q) {
                    const unsigned long long c = (q < len) ? (unsigned long long)tcam[ptr[j] + q] + 1ull : 0ull;
                    key |= c << (kbits * (kcams - 1 - q));
                }
                ki[n] = {key, j};
            }
        });
        mark("  tuple: keys");
        auto full_less = [&](int a, int b) {
            const bool la = cnt[a + 1] > 64, lb = cnt[b + 1] > 64;
            if (la != lb) return lb;
            return tuple_less(a, b);
        };
        if (wide) {
            std::stable_sort(order.begin(), order.end(), full_less);
        } else {
            // (key, idx) order.  Large inputs: ki is in ascending idx order, so a stable LSD radix sort over the key fields
            // yields it; one pass per camera field (kbits wide) and a last one for the long-track flag
            int fbits0 = 1;                                           // bits of a field value (camera id + 1 <= Nc)
            while (fbits0 < kbits && (1ll << fbits0) <= (long long)Nc) ++fbits0;
            int ibits = 1;
            while ((1ll << ibits) < (long long)Np) ++ibits;
            if (ki.size() >= 2048 && kcams * fbits0 + 1 + ibits <= 64) {
                // (round 4) Key and point index share ONE 64-bit word: [long-track flag | kcams camera fields of fbits0 bits | index], so the
                // stable LSD radix sort moves 8-byte records instead of 16-byte ones and only over the bits that are in use — config 4:
                // 4 x 10 + 1 key bits, 19 index bits; the same order as the 16-byte sort (equal keys keep ascending index).
                const int kshift = ibits;
                RawVec<unsigned long long> pk(ki.size()), tmp8(ki.size());
                pack_parallel_for((long long)ki.size(), [&](long long n0, long long n1) {
                    for (long long n = n0; n < n1; ++n) {
                        const unsigned long long key = ki[n].key;
                        unsigned long long c = 0;
                        for (int q = 0; q < kcams; ++q) c |= ((key >> (kbits * (kcams - 1 - q))) & ((1ull << kbits) - 1ull)) << (fbits0 * (kcams - 1 - q));
                        c |= (key >> 63) << (kcams * fbits0);
                        pk[n] = (c << kshift) | (unsigned long long)(unsigned)ki[n].idx;
                    }
                });
                const std::vector<long long> cut = pack_cuts((long long)pk.size(), 100000, 1);
                const int nch = (int)cut.size() - 1;
                std::vector<std::vector<unsigned>> hist(nch);
                const int total_bits = kcams * fbits0 + 1;
                for (int done = 0; done < total_bits; done += 11) {
                    const int bits = std::min(11, total_bits - done), shift = kshift + done;
                    const size_t nb = (size_t)1 << bits;
                    const unsigned long long mask = nb - 1;
                    pack_parallel_chunks(cut, [&](int t, long long n0, long long n1) {
                        hist[t].assign(nb, 0u);
                        for (long long n = n0; n < n1; ++n) hist[t][(pk[n] >> shift) & mask]++;
                    });
                    unsigned run = 0;
                    bool one_bucket = false;
                    for (size_t b2 = 0; b2 < nb; ++b2) {
                        unsigned in_bucket = 0;
                        for (int t = 0; t < nch; ++t) { const unsigned v = hist[t][b2]; hist[t][b2] = run; run += v; in_bucket += v; }
                        one_bucket |= in_bucket == pk.size();
                    }
                    if (one_bucket) continue;
                    pack_parallel_chunks(cut, [&](int t, long long n0, long long n1) {
                        for (long long n = n0; n < n1; ++n) tmp8[hist[t][(pk[n] >> shift) & mask]++] = pk[n];
                    });
                    pk.swap(tmp8);
                }
                const unsigned long long imask = (1ull << ibits) - 1ull;
                pack_parallel_for((long long)ki.size(), [&](long long n0, long long n1) {
                    for (long long n = n0; n < n1; ++n) {
                        const unsigned long long c = pk[n] >> kshift;
                        unsigned long long key = (c >> (kcams * fbits0)) << 63;
                        for (int q = 0; q < kcams; ++q) key |= ((c >> (fbits0 * (kcams - 1 - q))) & ((1ull << fbits0) - 1ull)) << (kbits * (kcams - 1 - q));
                        ki[n] = {key, (int)(pk[n] & imask)};
                    }
                });
            } else
            if (ki.size() < 2048) {                      // tiny calls: the bucket arrays would cost more than the sort
                std::sort(ki.begin(), ki.end(), [](const KI& a, const KI& b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; });
            } else {
                // parallel stable LSD radix sort: every piece of the input counts its digits, a pass over digits x pieces turns
                // the counts into start positions, every piece scatters its elements in input order.  Digits of at most 11
                // bits (bucket tables stay in L1/L2), and only the bits a camera id + 1 can occupy in each field.
                std::vector<KI> tmp(ki.size());
                const std::vector<long long> cut = pack_cuts((long long)ki.size(), 100000, 1);
                const int nch = (int)cut.size() - 1;
                std::vector<std::vector<unsigned>> hist(nch);
                auto pass = [&](int shift, int bits) {
                    const size_t nb = (size_t)1 << bits;
                    const unsigned long long mask = nb - 1;
                    pack_parallel_chunks(cut, [&](int t, long long n0, long long n1) {
                        hist[t].assign(nb, 0u);
                        for (long long n = n0; n < n1; ++n) hist[t][(ki[n].key >> shift) & mask]++;
                    });
                    unsigned run = 0;
                    bool one_bucket = false;
                    for (size_t b2 = 0; b2 < nb; ++b2) {
                        unsigned in_bucket = 0;
                        for (int t = 0; t < nch; ++t) { const unsigned v = hist[t][b2]; hist[t][b2] = run; run += v; in_bucket += v; }
                        one_bucket |= in_bucket == ki.size();
                    }
                    if (one_bucket) return;                              // every key has the same digit: nothing to move
                    pack_parallel_chunks(cut, [&](int t, long long n0, long long n1) {
                        for (long long n = n0; n < n1; ++n) tmp[hist[t][(ki[n].key >> shift) & mask]++] = ki[n];
                    });
                    ki.swap(tmp);
                };
                int fbits = 1;                                           // bits of a field value (camera id + 1 <= Nc)
                while (fbits < kbits && (1ll << fbits) <= (long long)Nc) ++fbits;
                for (int q = kcams - 1; q >= 0; --q)
                    for (int done = 0; done < fbits; done += 11) pass(kbits * (kcams - 1 - q) + done, std::min(11, fbits - done));
                pass(63, 1);
            }
            mark("  tuple: radix");
            sort_keys.resize(ki.size()); key_cams = kcams;
            pack_parallel_for((long long)ki.size(), [&](long long n0, long long n1) {
                for (long long n = n0; n < n1; ++n) { order[n] = ki[n].idx; sort_keys[n] = ki[n].key; }
            });
            for (size_t b = 0; b < ki.size();) {       // equal leading cameras: finish with the full comparison (stable)
                size_t e = b + 1;
                while (e < ki.size() && ki[e].key == ki[b].key) ++e;
                if (e - b > 1) {
                    bool longer = false;                // tuples that extend beyond the key can still differ
                    for (size_t n = b; n < e && !longer; ++n) longer = cnt[order[n] + 1] > kcams;
                    if (longer) std::stable_sort(order.begin() + b, order.begin() + e, full_less);
                }
                b = e;
            }
        }
    }
    mark("tuple sort");
    o.n_cams = Nc; o.n_pts = (int)order.size(); o.n_obs = No;
    o.pt_orig = order;
    o.pt_const.assign(o.n_pts, 0);
    o.items.clear();
    o.items.reserve(2 * ((size_t)No / 48 + 16));
    // A group of tracks with one camera tuple that fills at least a tile by itself starts on a tile boundary: its tiles are
    // then all regular (one dense Gram product each, small LDS footprint) instead of the first one mixing two tuples.
    std::vector<char> big_group_start(o.n_pts, 0);
    {
        // same[n]: track n has the same tuple as track n-1 (sorted order: one direction of the comparison suffices)
        std::vector<char> same(o.n_pts, 0);
        pack_parallel_for(o.n_pts, [&](long long n0, long long n1) {
            for (long long n = std::max<long long>(n0, 1); n < n1; ++n) {
                const int a = order[n - 1], b = order[n];
                // tuples no longer than the key are equal exactly when their keys are (the key holds every camera + 1)
                if (!sort_keys.empty() && cnt[a + 1] <= key_cams && cnt[b + 1] <= key_cams) same[n] = sort_keys[n - 1] == sort_keys[n];
                else same[n] = !tuple_less(a, b);
            }
        });
        for (int b = 0; b < o.n_pts;) {
            int e = b + 1;
            while (e < o.n_pts && same[e]) ++e;
            const int len = cnt[order[b] + 1];
            if (len <= 64 && (long long)(e - b) * len >= 64) big_group_start[b] = 1;
            b = e;
        }
    }
    mark("group starts");
    // Placement first (one cheap pass over the tracks: where every track starts, which tiles open), then the slots are
    // filled in parallel.
    std::vector<long long> trk_start(o.n_pts);
    {
        long long pos = 0;            // slots laid out so far (incl. padding)
        int cur_tile_start = -1;      // tile index of the open short tile, -1 if none
        auto pad = [&]() { pos = (pos + 63) & ~63LL; };
        for (int pj = 0; pj < o.n_pts; ++pj) {
            const int len = cnt[order[pj] + 1];
            if (len <= 64) {
                const int used = (int)(pos % 64);
                if (cur_tile_start < 0 || used + len > 64 || used == 0 || big_group_start[pj]) {
                    pad();
                    cur_tile_start = (int)(pos / 64);
                    o.items.push_back(cur_tile_start); o.items.push_back(1);
                }
            } else {
                pad();
                cur_tile_start = -1;
                o.items.push_back((int)(pos / 64)); o.items.push_back((len + 63) / 64);
            }
            trk_start[pj] = pos;
            pos += len;
            if (len > 64) pad();
        }
        pad();
        if (pos > INT32_MAX) return XRSFM_BA_EINVAL;
        o.n_slots = (int)pos;
    }
    o.n_tiles = o.n_slots / 64;
    o.slot_cam.resize(o.n_slots); o.slot_pt.resize(o.n_slots); o.slot_obs.resize(o.n_slots);
    o.slot_u.resize(o.n_slots); o.slot_v.resize(o.n_slots);
    mark("placement + alloc");
    pack_parallel_for(o.n_pts, [&](long long p0, long long p1) {
        for (long long pj = p0; pj < p1; ++pj) {
            const int j = order[pj];
            const int len = cnt[j + 1];
            o.pt_const[pj] = (p.point_const && p.point_const[j]) ? 1 : 0;
            const int* obs_b = csr.data() + ptr[j];
            const long long s0 = trk_start[pj];
            for (int q = 0; q < len; ++q) {
                const int i = obs_b[q];
                o.slot_cam[s0 + q] = tcam[ptr[j] + q]; o.slot_pt[s0 + q] = (int)pj; o.slot_obs[s0 + q] = i;
                o.slot_u[s0 + q] = p.obs_uv[2 * (size_t)i]; o.slot_v[s0 + q] = p.obs_uv[2 * (size_t)i + 1];
            }
            // padding up to the next track (or the end): owned by this track's thread as well
            const long long s1 = (pj + 1 < o.n_pts) ? trk_start[pj + 1] : (long long)o.n_slots;
            for (long long s2 = s0 + len; s2 < s1; ++s2) { o.slot_cam[s2] = -1; o.slot_pt[s2] = -1; o.slot_obs[s2] = -1; o.slot_u[s2] = 0.0; o.slot_v[s2] = 0.0; }
        }
    }, 50000);
    if (o.n_pts == 0 && o.n_slots > 0) return XRSFM_BA_EINVAL;       // (no tracks means no slots)
    mark("slots + uv");
    // per tile: longest track run; regular tiles = >= 2 tracks, all with the same camera tuple of length L <= 32
    o.tile_maxlen.assign(o.n_tiles, 1);
    o.tile_stride.assign(o.n_tiles, 0);
    pack_parallel_for(o.n_tiles, [&](long long t0, long long t1) {
        for (long long t = t0; t < t1; ++t) {
            const long long b0 = 64 * t;
            int run = 0, best = 1;
            for (int q = 0; q < 64; ++q) {
                const long long s2 = b0 + q;
                if (o.slot_cam[s2] < 0) break;
                run = (q > 0 && o.slot_pt[s2] == o.slot_pt[s2 - 1]) ? run + 1 : 1;
                best = std::max(best, run);
            }
            o.tile_maxlen[t] = best;
            if (o.slot_cam[b0] < 0) continue;
            int L = 1;
            while (L < 64 && o.slot_cam[b0 + L] >= 0 && o.slot_pt[b0 + L] == o.slot_pt[b0]) ++L;
            if (L > 32 || L >= 64) continue;
            bool regular = true;
            int nvalid = L;
            for (long long s2 = b0 + L; s2 < b0 + 64 && o.slot_cam[s2] >= 0; ++s2, ++nvalid) {
                const int r = (int)((s2 - b0) % L);
                if (o.slot_cam[s2] != o.slot_cam[b0 + r]) { regular = false; break; }
                if (r > 0 ? o.slot_pt[s2] != o.slot_pt[s2 - 1] : o.slot_pt[s2] == o.slot_pt[s2 - 1]) { regular = false; break; }
            }
            if (regular && nvalid % L == 0 && nvalid >= 2 * L && !wide) o.tile_stride[t] = L;
        }
    }, 4000);
    for (size_t it = 0; it + 1 < o.items.size(); it += 2)          // long items are never regular
        if (o.items[it + 1] > 1)
            for (int t = o.items[it]; t < o.items[it] + o.items[it + 1]; ++t) o.tile_stride[t] = 0;
    mark("maxlen + regular tiles");
    // Gram tiles
    o.tile_ncam.assign(o.n_tiles, 0); o.tile_gt_off.assign(o.n_tiles, -1);
    o.slot_cidx.assign(o.n_slots, 255); o.gt_cell.clear();
    {
        std::vector<char> single(o.n_tiles, 0);
        for (size_t it = 0; it + 1 < o.items.size(); it += 2)
            if (o.items[it + 1] == 1) single[o.items[it]] = 1;
        const int gram_max_cams = wide ? kGramMaxCamsWide : kGramMaxCams, gram_cw = wide ? 9 : 6;
        std::vector<int> tile_cams((size_t)o.n_tiles * kGramMaxCams, -1);       // ascending distinct cameras of the Gram tiles
        pack_parallel_for(o.n_tiles, [&](long long t0, long long t1) {
            int cams[64];
            for (long long t = t0; t < t1; ++t) {
                if (!single[t]) continue;
                const long long b0 = 64 * t;
                int nc = 0, ntrk = 0;
                for (int q = 0; q < 64 && o.slot_cam[b0 + q] >= 0; ++q) {
                    cams[nc++] = o.slot_cam[b0 + q];
                    ntrk += (q == 0 || o.slot_pt[b0 + q] != o.slot_pt[b0 + q - 1]);
                }
                if (nc == 0) continue;
                std::sort(cams, cams + nc);
                const int C = (int)(std::unique(cams, cams + nc) - cams);
                int passes = 1;
                if (C < 2 || C > gram_max_cams || gram_lds_need(C, ntrk, &passes, gram_cw) > kGramMaxLds) continue;
                o.tile_ncam[t] = C;
                for (int q = 0; q < C; ++q) tile_cams[(size_t)t * kGramMaxCams + q] = cams[q];
                for (int q = 0; q < 64 && o.slot_cam[b0 + q] >= 0; ++q)
                    o.slot_cidx[b0 + q] = (unsigned char)(std::lower_bound(cams, cams + C, o.slot_cam[b0 + q]) - cams);
            }
        }, 4000);
        mark("  gram: cameras of tiles");
        {
            size_t off = 0;
            for (int t = 0; t < o.n_tiles; ++t)
                if (o.tile_ncam[t] > 0) { o.tile_gt_off[t] = (int)off; off += (size_t)o.tile_ncam[t] * o.tile_ncam[t]; }
            o.gt_cell.assign(off, 0);
        }
        pack_parallel_for(o.n_tiles, [&](long long t0, long long t1) {
            for (long long t = t0; t < t1; ++t) {
                const int C = o.tile_ncam[t];
                if (C <= 0) continue;
                const long long b0 = 64 * t;
                unsigned char* cell = o.gt_cell.data() + o.tile_gt_off[t];
                for (int q = 0; q < 64 && o.slot_cam[b0 + q] >= 0; ++q)         // pairs inside a track (cameras ascend in a track)
                    for (int q2 = q + 1; q2 < 64 && o.slot_cam[b0 + q2] >= 0 && o.slot_pt[b0 + q2] == o.slot_pt[b0 + q]; ++q2)
                        cell[o.slot_cidx[b0 + q] * C + o.slot_cidx[b0 + q2]] = 1;
            }
        }, 4000);
        mark("  gram: cells");
        // The S-assembly kernel runs once per LDS class (<= 10 KB: 16 workgroups per CU; larger).  A handful of large tiles
        // is not worth a second launch (its duration is one tile's latency, ~15 us): they take the per-pair path instead.
        auto lds_need = [&](int t) {
            int ntrk = 0, passes = 1;
            for (int q = 0; q < 64 && o.slot_cam[64 * t + q] >= 0; ++q) ntrk += (q == 0 || o.slot_pt[64 * t + q] != o.slot_pt[64 * t + q - 1]);
            return (size_t)gram_lds_need(o.tile_ncam[t], ntrk, &passes, gram_cw);
        };
        int n_big = 0;
        for (int t = 0; t < o.n_tiles; ++t) n_big += (o.tile_ncam[t] > 0 && lds_need(t) > (size_t)kGramSmallLds);
        if (n_big > 0 && n_big * 20 <= o.n_tiles) {
            for (int t = 0; t < o.n_tiles; ++t)
                if (o.tile_ncam[t] > 0 && lds_need(t) > (size_t)kGramSmallLds) {
                    for (int q = 0; q < 64; ++q) o.slot_cidx[64 * t + q] = 255;
                    o.tile_ncam[t] = 0; o.tile_gt_off[t] = -1;        // (its table cells stay allocated, unused)
                }
        }
        o.n_gt_cells = (int)o.gt_cell.size();
    }
    mark("gram tiles");
    // camera-major positions of the lanes that write a camera-side partial: every valid lane of an irregular tile,
    // the first track (lanes < L) of a regular one.  Within a camera: slot order.
    // Stable counting sort by camera, in parallel: every piece of the slot range counts its flagged slots per camera, a serial
    // pass over cameras x pieces turns the counts into start positions, then every piece numbers its own slots.
    RawVec<char> w_all(o.n_slots), w_gram(o.n_slots);
    pack_parallel_for(o.n_tiles, [&](long long t0, long long t1) {
        for (long long t = t0; t < t1; ++t) {
            const int L = o.tile_stride[t], C = o.tile_ncam[t];
            bool seen[kGramMaxCams] = {false};
            for (int q = 0; q < 64; ++q) {
                const long long s2 = 64 * t + q;
                const char w = o.slot_cam[s2] >= 0 && (L == 0 || q < L);
                w_all[s2] = w;
                if (C <= 0) { w_gram[s2] = w; continue; }
                char g = 0;
                if (o.slot_cam[s2] >= 0) { const int ci = o.slot_cidx[s2]; if (!seen[ci]) { seen[ci] = true; g = 1; } }
                w_gram[s2] = g;
            }
        }
    }, 4000);
    auto camera_major = [&](const RawVec<char>& flag, std::vector<int>& cam_ptr, RawVec<int>& campos) {
        const std::vector<long long> cut = pack_cuts(o.n_slots, 400000, 64);
        const int nch = (int)cut.size() - 1;
        std::vector<std::vector<int>> local(nch);
        pack_parallel_chunks(cut, [&](int t, long long s0, long long s1) {
            local[t].assign(Nc, 0);
            for (long long s2 = s0; s2 < s1; ++s2) if (flag[s2]) local[t][o.slot_cam[s2]]++;
        });
        cam_ptr.assign(Nc + 1, 0);
        int run = 0;
        for (int c = 0; c < Nc; ++c) {
            cam_ptr[c] = run;
            for (int t = 0; t < nch; ++t) { const int v = local[t][c]; local[t][c] = run; run += v; }
        }
        cam_ptr[Nc] = run;
        campos.resize(o.n_slots);
        pack_parallel_chunks(cut, [&](int t, long long s0, long long s1) {
            for (long long s2 = s0; s2 < s1; ++s2) campos[s2] = flag[s2] ? local[t][o.slot_cam[s2]]++ : -1;
        });
    };
    camera_major(w_all, o.cam_ptr, o.slot_campos);
    o.n_cam_entries = o.cam_ptr[Nc];
    camera_major(w_gram, o.cam_ptr_g, o.slot_campos_g);
    o.n_cam_entries_g = o.cam_ptr_g[Nc];
    mark("camera-major maps");
    std::vector<char> cam_seen(Nc, 0);
    for (int c = 0; c < Nc; ++c) cam_seen[c] = o.cam_ptr[c + 1] > o.cam_ptr[c];      // (every observed camera owns at least one camera-major entry)
    // effective parameter count (num_effective_parameters_reduced)
    o.n_var_q = o.n_var_t = o.n_var_p = 0;
    for (int c = 0; c < Nc; ++c) {
        if (!cam_seen[c]) continue;
        const unsigned cc = p.cam_const ? p.cam_const[c] : 0u;
        if (!(cc & XRSFM_BA_CONST_Q)) o.n_var_q++;
        if (!(cc & XRSFM_BA_CONST_T)) o.n_var_t++;
    }
    {
        const std::vector<long long> pc = pack_cuts(o.n_pts, 200000, 1);
        std::vector<int> part((int)pc.size() - 1, 0);
        pack_parallel_chunks(pc, [&](int t, long long p0, long long p1) { int n = 0; for (long long pj = p0; pj < p1; ++pj) n += !o.pt_const[pj]; part[t] = n; });
        for (int v : part) o.n_var_p += v;
    }
    mark("counts");
    return XRSFM_BA_OK;
}

}  // namespace xba
