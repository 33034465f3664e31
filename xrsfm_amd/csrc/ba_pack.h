// Host-side packing: caller SoA (include/xrsfm_ba.h) -> track-major 64-slot tiles
// + camera-major positions.  Replaces the reference's per-observation
// problem.AddResidualBlock loop (/root/reference/src/optimization/ba_solver.cc:336-349).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/xrsfm_ba.h"

namespace xba {

constexpr int kGramMaxCams = 10;         // 60 operand rows: 4 MFMA row tiles
constexpr int kGramMaxLds = 20 * 1024;   // staged operand of one tile (bytes); larger tiles use the per-pair path
constexpr int kGramSmallLds = 10240;     // LDS class boundary of the S-assembly launches: 160 KB / 16 workgroups

// LDS bytes of a Gram tile with C cameras and T tracks: operand [6C][3T'+pad] + the C x C destination table, where the tracks
// are staged in ONE pass (T' = T) if that fits the small class and otherwise in TWO passes of T' = ceil(T/2) tracks with the
// accumulators kept across the passes.  The kernel (k_schur_pairs) evaluates the same rule.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int gram_lds_need(int C, int T, int* passes) {
    const int one = 6 * C * ((((3 * T + 3) & ~3)) + 2) * 8 + C * C * 4;
    if (one <= kGramSmallLds) { *passes = 1; return one; }
    const int Th = (T + 1) / 2;
    *passes = 2;
    return 6 * C * ((((3 * Th + 3) & ~3)) + 2) * 8 + C * C * 4;
}

// Host-side helper: run fn(begin, end) over [0, n) on up to 8 threads (large problems only; the packing of config L is ~100 ms
// of single-thread work otherwise).  The pieces are disjoint, so the result does not depend on the thread count.
template <typename F>
inline void pack_parallel_for(long long n, F&& fn, long long min_n = 200000) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (n < min_n || hw < 2) ? 1 : (int)std::min<unsigned>(8u, hw);
    if (nt == 1) { fn(0LL, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { fn(n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

struct Packed {
    int n_cams = 0, n_pts = 0, n_obs = 0, n_tiles = 0, n_slots = 0;
    std::vector<int> pt_orig;        // packed point -> caller point index
    std::vector<int> slot_cam, slot_pt, slot_campos, slot_obs;  // slot_obs: caller obs index (-1 pad)
    std::vector<double> slot_u, slot_v;
    std::vector<int> items;          // pairs {first_tile, n_tiles}
    std::vector<int> cam_ptr;        // [n_cams+1] into the camera-major scatter buffer (entries, not observations)
    std::vector<int> tile_maxlen;    // [n_tiles] longest track in the tile (<= 64)
    std::vector<int> tile_stride;    // [n_tiles] L > 0: every track of the tile has the same L cameras ("regular" tile)
    int n_cam_entries = 0;
    // The same for the S assembly (k_schur_pairs -> k_chol_segsum): a Gram tile sums the diagonal-block / rhs terms of every
    // camera over its tracks first, so only the first lane of each distinct camera writes an entry.
    std::vector<int> cam_ptr_g, slot_campos_g;
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

inline int pack_problem(const xrsfm_ba_problem& p, Packed& o) {
    if (p.n_cams < 0 || p.n_points < 0 || p.n_obs < 0 || p.n_intr < 0) return XRSFM_BA_EINVAL;
    if (p.n_obs > 0 && (!p.obs_cam || !p.obs_pt || !p.obs_uv)) return XRSFM_BA_EINVAL;
    if (p.n_cams > 0 && (!p.cam_q || !p.cam_t || !p.cam_intr)) return XRSFM_BA_EINVAL;
    if (p.n_points > 0 && !p.points) return XRSFM_BA_EINVAL;
    if (p.n_intr > 0 && (!p.intr_model || !p.intr_params)) return XRSFM_BA_EINVAL;
    for (int c = 0; c < p.n_cams; ++c) {
        const int ii = p.cam_intr[c];
        if (ii < 0 || ii >= p.n_intr) return XRSFM_BA_EINVAL;
        const int m = p.intr_model[ii];
        if (m < 0 || m > 4) return XRSFM_BA_EINVAL;
    }
    const int No = p.n_obs, Np = p.n_points, Nc = p.n_cams;
    // XRSFM_BA_PACK_TIMING=1: phase times of the host-side packing on stderr (developer aid)
    const bool timing = std::getenv("XRSFM_BA_PACK_TIMING") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[pack] %-28s %7.2f ms\n", what, 1e3 * std::chrono::duration<double>(t - t_prev).count());
        t_prev = t;
    };
    std::vector<int> cnt(Np + 1, 0), mincam(Np, INT32_MAX);
    for (int i = 0; i < No; ++i) {
        const int c = p.obs_cam[i], j = p.obs_pt[i];
        if (c < 0 || c >= Nc || j < 0 || j >= Np) return XRSFM_BA_EINVAL;
        cnt[j + 1]++;
        mincam[j] = std::min(mincam[j], c);
    }
    // CSR by caller point
    std::vector<int> ptr(Np + 1, 0);
    for (int j = 0; j < Np; ++j) ptr[j + 1] = ptr[j] + cnt[j + 1];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1), csr(No);
    for (int i = 0; i < No; ++i) csr[fill[p.obs_pt[i]]++] = i;
    mark("csr by point");
    // observations of every track ordered by camera
    pack_parallel_for(Np, [&](long long j0, long long j1) {
        for (long long j = j0; j < j1; ++j)
            std::stable_sort(csr.begin() + ptr[j], csr.begin() + ptr[j + 1], [&](int a, int b) { return p.obs_cam[a] < p.obs_cam[b]; });
    });
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
            const int ca = p.obs_cam[csr[ptr[a] + k]], cb = p.obs_cam[csr[ptr[b] + k]];
            if (ca != cb) return ca < cb;
        }
        return la < lb;
    };
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
                    const unsigned long long c = (q < len) ? (unsigned long long)p.obs_cam[csr[ptr[j] + q]] + 1ull : 0ull;
                    key |= c << (kbits * (kcams - 1 - q));
                }
                ki[n] = {key, j};
            }
        });
        auto full_less = [&](int a, int b) {
            const bool la = cnt[a + 1] > 64, lb = cnt[b + 1] > 64;
            if (la != lb) return lb;
            return tuple_less(a, b);
        };
        if (wide) {
            std::stable_sort(order.begin(), order.end(), full_less);
        } else {
            const auto ki_less = [](const KI& a, const KI& b) { return a.key != b.key ? a.key < b.key : a.idx < b.idx; };
            if (ki.size() < 400000) {
                std::sort(ki.begin(), ki.end(), ki_less);
            } else {                                    // 8 sorted runs in parallel, then three rounds of pairwise merges (a total order: unique result)
                const size_t nk = ki.size();
                size_t cut[9];
                for (int q = 0; q <= 8; ++q) cut[q] = nk * q / 8;
                pack_parallel_for(8 * 200000LL, [&](long long a, long long b) {
                    for (long long q = a / 200000; q < b / 200000; ++q) std::sort(ki.begin() + cut[q], ki.begin() + cut[q + 1], ki_less);
                });
                for (int width = 1; width < 8; width *= 2) {
                    std::vector<std::thread> th;
                    for (int q = 0; q + width < 8; q += 2 * width)
                        th.emplace_back([&, q, width] { std::inplace_merge(ki.begin() + cut[q], ki.begin() + cut[q + width], ki.begin() + cut[std::min(q + 2 * width, 8)], ki_less); });
                    for (auto& x : th) x.join();
                }
            }
            for (size_t n = 0; n < ki.size(); ++n) order[n] = ki[n].idx;
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
    o.items.clear(); o.slot_cam.clear(); o.slot_pt.clear(); o.slot_obs.clear();
    {
        const size_t cap = (size_t)No + (size_t)No / 8 + 4096;      // slots incl. tile padding (grows if a pathological input needs more)
        o.slot_cam.reserve(cap); o.slot_pt.reserve(cap); o.slot_obs.reserve(cap); o.items.reserve(2 * (cap / 64 + 1));
    }
    auto pad_tile = [&]() {
        while (o.slot_cam.size() % 64) { o.slot_cam.push_back(-1); o.slot_pt.push_back(-1); o.slot_obs.push_back(-1); }
    };
    // A group of tracks with one camera tuple that fills at least a tile by itself starts on a tile boundary: its tiles are
    // then all regular (one dense Gram product each, small LDS footprint) instead of the first one mixing two tuples.
    std::vector<char> big_group_start(o.n_pts, 0);
    {
        // same[n]: track n has the same tuple as track n-1 (sorted order: one direction of the comparison suffices)
        std::vector<char> same(o.n_pts, 0);
        pack_parallel_for(o.n_pts, [&](long long n0, long long n1) {
            for (long long n = std::max<long long>(n0, 1); n < n1; ++n) same[n] = !tuple_less(order[n - 1], order[n]);
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
    int cur_tile_start = -1;  // tile index of the open short tile, -1 if none
    for (int pj = 0; pj < o.n_pts; ++pj) {
        const int j = order[pj];
        const int len = cnt[j + 1];
        o.pt_const[pj] = (p.point_const && p.point_const[j]) ? 1 : 0;
        const int* obs_b = csr.data() + ptr[j];
        if (len <= 64) {
            const int used = (int)(o.slot_cam.size() % 64);
            if (cur_tile_start < 0 || used + len > 64 || used == 0 || big_group_start[pj]) {
                pad_tile();
                cur_tile_start = (int)(o.slot_cam.size() / 64);
                o.items.push_back(cur_tile_start); o.items.push_back(1);
            }
        } else {
            pad_tile();
            cur_tile_start = -1;
            o.items.push_back((int)(o.slot_cam.size() / 64)); o.items.push_back((len + 63) / 64);
        }
        for (int q = 0; q < len; ++q) { const int i = obs_b[q]; o.slot_cam.push_back(p.obs_cam[i]); o.slot_pt.push_back(pj); o.slot_obs.push_back(i); }
        if (len > 64) pad_tile();
    }
    pad_tile();
    o.n_slots = (int)o.slot_cam.size();
    o.n_tiles = o.n_slots / 64;
    o.slot_u.assign(o.n_slots, 0.0); o.slot_v.assign(o.n_slots, 0.0);
    pack_parallel_for(o.n_slots, [&](long long s0, long long s1) {
        for (long long s = s0; s < s1; ++s)
            if (o.slot_obs[s] >= 0) { o.slot_u[s] = p.obs_uv[2 * (size_t)o.slot_obs[s]]; o.slot_v[s] = p.obs_uv[2 * (size_t)o.slot_obs[s] + 1]; }
    });
    mark("slots + uv");
    o.tile_maxlen.assign(o.n_tiles, 1);
    for (int t = 0; t < o.n_tiles; ++t) {
        int run = 0, best = 1;
        for (int q = 0; q < 64; ++q) {
            const int s2 = 64 * t + q;
            if (o.slot_cam[s2] < 0) break;
            run = (q > 0 && o.slot_pt[s2] == o.slot_pt[s2 - 1]) ? run + 1 : 1;
            best = std::max(best, run);
        }
        o.tile_maxlen[t] = best;
    }
    // regular tiles: >= 2 tracks, all with the same camera tuple of length L <= 32
    o.tile_stride.assign(o.n_tiles, 0);
    for (int t = 0; t < o.n_tiles; ++t) {
        const int b0 = 64 * t;
        if (o.slot_cam[b0] < 0) continue;
        int L = 1;
        while (L < 64 && o.slot_cam[b0 + L] >= 0 && o.slot_pt[b0 + L] == o.slot_pt[b0]) ++L;
        if (L > 32 || L >= 64) continue;
        bool regular = true;
        int nvalid = L;
        for (int s2 = b0 + L; s2 < b0 + 64 && o.slot_cam[s2] >= 0; ++s2, ++nvalid) {
            const int r = (s2 - b0) % L;
            if (o.slot_cam[s2] != o.slot_cam[b0 + r]) { regular = false; break; }
            if (r > 0 ? o.slot_pt[s2] != o.slot_pt[s2 - 1] : o.slot_pt[s2] == o.slot_pt[s2 - 1]) { regular = false; break; }
        }
        if (regular && nvalid % L == 0 && nvalid >= 2 * L) o.tile_stride[t] = L;
    }
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
                if (C < 2 || C > kGramMaxCams || gram_lds_need(C, ntrk, &passes) > kGramMaxLds) continue;
                o.tile_ncam[t] = C;
                for (int q = 0; q < C; ++q) tile_cams[(size_t)t * kGramMaxCams + q] = cams[q];
                for (int q = 0; q < 64 && o.slot_cam[b0 + q] >= 0; ++q)
                    o.slot_cidx[b0 + q] = (unsigned char)(std::lower_bound(cams, cams + C, o.slot_cam[b0 + q]) - cams);
            }
        }, 4000);
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
        // The S-assembly kernel runs once per LDS class (<= 10 KB: 16 workgroups per CU; larger).  A handful of large tiles
        // is not worth a second launch (its duration is one tile's latency, ~15 us): they take the per-pair path instead.
        auto lds_need = [&](int t) {
            int ntrk = 0, passes = 1;
            for (int q = 0; q < 64 && o.slot_cam[64 * t + q] >= 0; ++q) ntrk += (q == 0 || o.slot_pt[64 * t + q] != o.slot_pt[64 * t + q - 1]);
            return (size_t)gram_lds_need(o.tile_ncam[t], ntrk, &passes);
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
    auto writes = [&](int s2) { const int L = o.tile_stride[s2 / 64]; return o.slot_cam[s2] >= 0 && (L == 0 || (s2 % 64) < L); };
    o.cam_ptr.assign(Nc + 1, 0);
    for (int s2 = 0; s2 < o.n_slots; ++s2)
        if (writes(s2)) o.cam_ptr[o.slot_cam[s2] + 1]++;
    for (int c = 0; c < Nc; ++c) o.cam_ptr[c + 1] += o.cam_ptr[c];
    o.n_cam_entries = o.cam_ptr[Nc];
    std::vector<int> cf(o.cam_ptr.begin(), o.cam_ptr.end() - 1);
    o.slot_campos.assign(o.n_slots, -1);
    for (int s2 = 0; s2 < o.n_slots; ++s2)
        if (writes(s2)) o.slot_campos[s2] = cf[o.slot_cam[s2]]++;
    {
        std::vector<char> wg(o.n_slots, 0);
        for (int t = 0; t < o.n_tiles; ++t) {
            const int C = o.tile_ncam[t];
            if (C <= 0) { for (int q = 0; q < 64; ++q) wg[64 * t + q] = writes(64 * t + q); continue; }
            bool seen[kGramMaxCams] = {false};
            for (int q = 0; q < 64 && o.slot_cam[64 * t + q] >= 0; ++q) {
                const int ci = o.slot_cidx[64 * t + q];
                if (!seen[ci]) { seen[ci] = true; wg[64 * t + q] = 1; }
            }
        }
        o.cam_ptr_g.assign(Nc + 1, 0);
        for (int s2 = 0; s2 < o.n_slots; ++s2)
            if (wg[s2]) o.cam_ptr_g[o.slot_cam[s2] + 1]++;
        for (int c = 0; c < Nc; ++c) o.cam_ptr_g[c + 1] += o.cam_ptr_g[c];
        o.n_cam_entries_g = o.cam_ptr_g[Nc];
        std::vector<int> cg(o.cam_ptr_g.begin(), o.cam_ptr_g.end() - 1);
        o.slot_campos_g.assign(o.n_slots, -1);
        for (int s2 = 0; s2 < o.n_slots; ++s2)
            if (wg[s2]) o.slot_campos_g[s2] = cg[o.slot_cam[s2]]++;
    }
    mark("camera-major maps");
    std::vector<char> cam_seen(Nc, 0);
    for (int s2 = 0; s2 < o.n_slots; ++s2)
        if (o.slot_cam[s2] >= 0) cam_seen[o.slot_cam[s2]] = 1;
    // effective parameter count (num_effective_parameters_reduced)
    o.n_var_q = o.n_var_t = o.n_var_p = 0;
    for (int c = 0; c < Nc; ++c) {
        if (!cam_seen[c]) continue;
        const unsigned cc = p.cam_const ? p.cam_const[c] : 0u;
        if (!(cc & XRSFM_BA_CONST_Q)) o.n_var_q++;
        if (!(cc & XRSFM_BA_CONST_T)) o.n_var_t++;
    }
    for (int pj = 0; pj < o.n_pts; ++pj)
        if (!o.pt_const[pj]) o.n_var_p++;
    mark("counts");
    return XRSFM_BA_OK;
}

}  // namespace xba
