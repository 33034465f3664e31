// Host-side plan of the explicit reduced-camera solve (no device calls: testable without a GPU through
// xrsfm_ba_debug_chol_plan): camera-pair blocks and their scatter destinations, elimination order of the cameras,
// symbolic factorisation of the 64x64 tile pattern, elimination-tree levels and the per-level work lists.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

#include "ba_pack.h"

namespace xba {

constexpr int kPlanTile = 64;        // tile size (must equal kNB of ba_chol.h)
constexpr int kPlanMaxTiles = 4096;  // tile columns a plan may have beyond the dense limit (T x T tile maps stay small: 262 000 unknowns)
constexpr int kCamsPerTile = 10;     // 10 cameras = 60 rows per tile + 4 identity padding rows
constexpr int kCamsPerTileWide = 7;  // bal9 mode (9 unknowns per camera): 7 cameras = 63 rows per tile + 1 identity padding row

struct CholPlan {
    int n = 0, n_pad = 0, T = 0, n_blocks = 0, n_pairs = 0, n_writes = 0, n_tiles_nz = 0, n_levels = 0;
    int n_pair_writes = 0;           // block entries that are per-pair blocks (n_writes - the Gram cells in use)
    int cam_width = 6, cams_per_tile = kCamsPerTile;     // unknowns per camera (9 in bal9 mode) and cameras per 64-row tile
    int ordering = 0;                // 0 natural, 1 multi-way nested dissection of a band / ring, 3 nested dissection of an unordered camera graph, 2 reverse Cuthill-McKee (unordered
                                     // collections with viewpoint clusters: banded fill instead of a dense factor)
    long long tile_products = 0;     // 64x64x64 tile products of one factorisation (symbolic count; 2 * 64^3 flop each)
    int n_hubs = 0, band = 0;
    bool use_levels = false, panel_ll = false;
    // look-ahead panel schedule (panel schedule of a pure chain: one column per level): the contributions j < k - 2 of column k
    // reach it through partial products (formed in the launch that factors column k - 2) + their fixed-order sum (in the launch
    // of column k - 1), and the fused factor kernel adds j = k - 2, k - 1 itself: one launch per column (k_panel_slot,
    // ba_chol.h) instead of the chain update -> sum -> factor (72 us per column at config T)
    bool lookahead = false;
    // k_schur_pairs is launched once per LDS class: items whose staged operand fits 10 KB (16 workgroups per CU) and the rest
    size_t pairs_shm = 0, pairs_shm_big = 0;
    std::vector<int> pairs_items;    // Gram tiles of the small class | Gram tiles of the big class (tile indices) | other items (item indices)
    int n_pairs_small = 0, n_pairs_big = 0, n_pairs_other = 0;
    int gram_n[8] = {0};             // Gram tiles per launch bucket 2 * (NI - 1) + (0 small | 1 big LDS class), in pairs_items order
    size_t gram_shm[8] = {0};        // dynamic LDS of each bucket's launch
    std::vector<int> spp, pair_dst, blk_ptr, blk_rc, cam_off, tile_rows, tiles_nz;
    std::vector<int> cols_flat, cols_off;                                  // per tile column its row tiles j < k (push-form backward substitution)
    // ... two columns per launch (k_bwd2): pair p = columns (T-1-2p, T-2-2p); per pair the union of their row tiles j < the lower
    // column as (j, flags: bit 0 = L(k,j) non-zero, bit 1 = L(k-1,j)), and whether L(k,k-1) is non-zero
    std::vector<int> bw2_ent, bw2_off, bw2_link;
    std::vector<int> lv_k, lv_tgt, lv_cptr, lv_cj, lv_bptr, lv_bi;         // level schedule
    std::vector<int> lv_k_off, lv_tgt_off;
    // thin upper levels (few targets with long contribution lists): the lists are cut into chunks, one workgroup per
    // chunk writes a partial tile, a second launch adds the partials of a target in list order
    std::vector<int> sp_tgt, sp_q;      // per chunk: (i,k) and the [q0,q1) range in lv_cj
    std::vector<int> sp_rt, sp_rp;      // per split target: (i,k) and its [p0,p1) range of partials (level-relative)
    std::vector<int> sp_chunk_off, sp_rt_off;   // per level (size n_levels+1)
    int sp_max_chunks = 0;
    // (round 6) look-ahead on the LEVEL schedule of a dissected collection (config 5's shape: 174 levels, each update -> sum -> factor
    // with the chip idle through the last two): the contributions of a level-l target are cut into EARLY ones — columns of levels
    // <= l - 1 - la_depth, final long before level l is reached — and LATE ones; the early chunks of level l run on a second stream
    // while the main stream is still at the levels l - la_depth .. l - 1, the late chunks (a few per cent of the products) follow the
    // previous level's factor kernel, and the fixed-order sum adds both.  Per level the chunk list holds the early chunks first
    // (sp_e_cnt of them); a chunk's partial slot is sp_slot (a target's slots stay one contiguous range: early, then late).
    int la_depth = 0;                   // 0: off (every chunk in one launch, list order)
    std::vector<int> sp_slot;           // per chunk: partial slot (level-relative)
    std::vector<int> sp_e_cnt;          // per level: early chunks
    // panel schedule, dense part of the pattern: an even number of consecutive single-column levels (columns K0, K0+1, ...) share
    // one update launch of 128x128 macro tiles (k_panel2_part: rows (i0,i1) x columns (k0,k1), contributions j in [q0,q1) of
    // [0,K0)); the later levels of the panel then only add the panel's own earlier columns inside the fused factor kernel.
    //   per entry 8 ints: i0, i1 (-1: none), k0, k1, q0, q1, first partial index, partial stride between its four tiles
    //   mp_wg: per level W+1 offsets into the entries (workgroup w walks entries [mp_wg[w], mp_wg[w+1])); mp_off: per level
    //   offsets into mp_wg
    std::vector<int> mp_chunk, mp_wg, mp_off;
    // fill lists: per structurally non-zero tile (tiles_nz order) its 6x6 blocks: entry >= 0 off-diagonal block id,
    // entry < 0 the diagonal block of camera -(entry+1)
    std::vector<int> tf_ptr, tf_ent;
    // fused level kernel (k_lv_factor, ba_chol.h): one workgroup per structurally non-zero tile (i,k) of a level's columns;
    // it forms the update of the DIAGONAL tile (k,k) itself (every workgroup of a column repeats that sum and the
    // factorisation of the 64x64 pivot tile bit for bit, so no tile of a level waits for another one), and of its own tile.
    //   direct level: fz_dptr/fz_dj = the row tiles j < k of column k's diagonal; entry j if (i,j) is non-zero as well
    //                 (it then also contributes to tile (i,k)), ~j otherwise;
    //   split level:  empty lists (k_ll_update_part + k_ll_update_reduce have updated the tiles and the right-hand side in place).
    std::vector<int> fz_tile, fz_dptr, fz_dj, fz_off;
    // look-ahead schedule: the contribution of column k - 2 to column k is formed in the launch of column k - 1 — one single-product
    // chunk per tile (md_tgt: (i,k); md_q: its entry in md_cj), written to partial slot = its index within the level — and the
    // factor kernel of column k starts its accumulators from it: fz_late, per fused-kernel entry the slot of (k,k) and of (i,k), -1 none
    std::vector<int> tile_map;          // [T][T] packed tile storage: index of tile (i,k) in tiles_nz order, n_tiles_nz (= the shared zero tile) outside the pattern
    std::vector<int> md_tgt, md_q, md_cj, md_off, fz_late;      // (md_q indexes md_cj: the CSR over lv_cj must stay contiguous)
    int md_max = 0;
    std::vector<int> fz_q;              // per fused-kernel entry: index in tiles_nz of (k,k) and of (i,k) (fill lists tf_ptr / tf_ent)
    std::vector<int> fill_rest;         // tiles_nz indices of the tiles whose column is not in level 0
    std::vector<int> tile_cam;          // [T][kCamsPerTile] camera in slot q of tile t, -1 = none (backward kernel: candidate cameras)
};

struct PairKey {                 // ((cam_b << 32) | cam_a, pair index); trivially constructible: the lists below are never zero-filled
    unsigned long long first; int second;
    bool operator<(const PairKey& o) const { return first != o.first ? first < o.first : second < o.second; }
};
typedef RawVec<PairKey> PairKeys;   // sorted by (key, index)

// A track with two observations in the same camera (the reference guards against it, pnp.cc:84, but the map format allows it;
// Ceres simply adds both residuals): its camera pair (a,a) is a diagonal block, which the pair-block assembly does not
// produce.  Internal code: AUTO takes the implicit-Schur PCG path, which treats every observation on its own; an explicit
// CHOLESKY request is refused with XRSFM_BA_EINVAL.
constexpr int kErrDuplicateObs = -100;
constexpr int kErrPlanCheck = -101;      // XRSFM_BA_PLAN_CHECK: the schedule does not cover the factorisation (internal error)

// Pairs (a = slot, b = slot + dd in the same track) -> block (cam_b, cam_a), cam_b > cam_a.  `keyed` lists everything that
// WRITES a partial block, as (block key, index into pair_dst):
//   * a Gram tile (ba_pack.h) writes one partial per camera pair that some track of the tile sees together; its index is
//     n_obs_pairs + the cell of the tile's C x C table;
//   * every other tile writes one partial per observation pair; its index is the pair index spp[s] + dd - 1.
inline int chol_local_keys(const Packed& k, std::vector<int>& spp, PairKeys& keyed) {
    const int ns = k.n_slots;
    // Two parallel passes over fixed pieces of the slot range (a track may continue into the next piece: long items):
    //   A  spp[s + 1] <- number of later slots of the same track (= pairs that start at s), piece totals
    //   B  prefix sums in place, validation, and the keys of the per-pair path
    const std::vector<long long> cut = pack_cuts(ns, 400000, 64);
    const int nch = (int)cut.size() - 1;
    spp.resize((size_t)ns + 1);
    spp[0] = 0;
    std::vector<long long> total(nch, 0), nkeys(nch, 0);
    pack_parallel_chunks(cut, [&](int t, long long s0, long long s1) {
        int run = 0;
        if (s1 < ns && k.slot_cam[s1] >= 0)
            for (long long s = s1 + 1; s < ns && k.slot_cam[s] >= 0 && k.slot_pt[s] == k.slot_pt[s1]; ++s) ++run;
        long long sum = 0, keys = 0;
        for (long long s = s1 - 1; s >= s0; --s) {
            if (k.slot_cam[s] < 0) run = 0;
            else run = (s + 1 < ns && k.slot_cam[s + 1] >= 0 && k.slot_pt[s + 1] == k.slot_pt[s]) ? run + 1 : 0;
            spp[s + 1] = run;
            sum += run;
            if (k.tile_ncam[s / 64] <= 0) keys += run;       // (a Gram tile writes per camera pair of the tile, below)
        }
        total[t] = sum; nkeys[t] = keys;
    });
    std::vector<long long> base(nch + 1, 0), kbase(nch + 1, 0);
    for (int t = 0; t < nch; ++t) { base[t + 1] = base[t] + total[t]; kbase[t + 1] = kbase[t] + nkeys[t]; }
    if (base[nch] > INT32_MAX) return XRSFM_BA_EINVAL;
    const int n_obs_pairs = (int)base[nch];
    // Gram tiles: one key per co-visible camera pair of the tile (counted first, so that every piece writes its keys in place:
    // an unordered collection with long tracks has tens of millions of keys, and neither per-piece lists nor a zero-filled
    // result are affordable — config T: 61.8 M keys, 1 GB)
    const std::vector<long long> tcut = pack_cuts(k.n_tiles, 8000, 1);
    const int ntc = (int)tcut.size() - 1;
    std::vector<long long> gkeys(ntc, 0);
    pack_parallel_chunks(tcut, [&](int t, long long t0, long long t1) {
        long long n = 0;
        for (long long tile = t0; tile < t1; ++tile) {
            const int C = k.tile_ncam[tile];
            if (C <= 0) continue;
            const unsigned char* cell = k.gt_cell.data() + k.tile_gt_off[tile];
            for (int a = 0; a < C; ++a)
                for (int b = a + 1; b < C; ++b) n += cell[a * C + b] != 0;
        }
        gkeys[t] = n;
    });
    std::vector<long long> gbase(ntc + 1, kbase[nch]);
    for (int t = 0; t < ntc; ++t) gbase[t + 1] = gbase[t] + gkeys[t];
    if (gbase[ntc] > INT32_MAX) return XRSFM_BA_EINVAL;
    keyed.clear();
    keyed.resize((size_t)gbase[ntc]);
    std::vector<char> bad(nch, 0);
    pack_parallel_chunks(cut, [&](int t, long long s0, long long s1) {
        int acc = (int)base[t];
        PairKey* out = keyed.data() + kbase[t];
        for (long long s = s0; s < s1; ++s) {
            const int start = acc, np = spp[s + 1];
            acc += np;
            spp[s + 1] = acc;
            if (np == 0) continue;
            const bool gram = k.tile_ncam[s / 64] > 0;
            const unsigned long long ca = (unsigned)k.slot_cam[s];
            for (int dd = 1; dd <= np; ++dd) {
                const unsigned long long cb = (unsigned)k.slot_cam[s + dd];
                if (cb <= ca) { bad[t] = 1; return; }   // two observations of one track in the same frame
                if (!gram) *out++ = PairKey{(cb << 32) | ca, start + dd - 1};
            }
        }
    });
    for (int t = 0; t < nch; ++t) if (bad[t]) return kErrDuplicateObs;
    pack_parallel_chunks(tcut, [&](int t, long long t0, long long t1) {
        int cams[64];
        PairKey* out = keyed.data() + gbase[t];
        for (long long tile = t0; tile < t1; ++tile) {
            const int C = k.tile_ncam[tile];
            if (C <= 0) continue;
            for (int q = 0; q < 64 && k.slot_cam[64 * tile + q] >= 0; ++q) cams[k.slot_cidx[64 * tile + q]] = k.slot_cam[64 * tile + q];
            const unsigned char* cell = k.gt_cell.data() + k.tile_gt_off[tile];
            for (int a = 0; a < C; ++a)
                for (int b = a + 1; b < C; ++b)
                    if (cell[a * C + b])
                        *out++ = PairKey{((unsigned long long)(unsigned)cams[b] << 32) | (unsigned)cams[a], n_obs_pairs + k.tile_gt_off[tile] + a * C + b};
        }
    });
    // (key, index) order.  The entries were appended in ascending index order (pairs by slot, then the Gram cells by tile), so a
    // stable LSD radix sort over the two camera fields of the key gives it; small lists take std::sort.  Each pass runs on up to
    // 16 threads over fixed pieces of the list: per-piece histograms, offsets by (digit, piece), pieces scattered in order —
    // the result is the serial pass's, whatever the thread count (config T: 61.8 M keys, 1.08 s single-threaded).
    if (keyed.size() < 50000) { std::sort(keyed.begin(), keyed.end()); return 0; }
    int cam_bits = 1;
    while (cam_bits < 32 && (1ll << cam_bits) < (long long)k.n_cams) ++cam_bits;
    PairKeys tmp(keyed.size());
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = (keyed.size() < 2000000 || hw < 2) ? 1 : (int)std::min<unsigned>(16u, hw);
    std::vector<long long> rc(nt + 1);
    for (int t = 0; t <= nt; ++t) rc[t] = (long long)(keyed.size() * (size_t)t / (size_t)nt);
    std::vector<std::vector<unsigned>> hist(nt);
    auto pass = [&](int shift, int bits) {
        const size_t nb = (size_t)1 << bits;
        const unsigned long long mask = nb - 1;
        pack_parallel_chunks(rc, [&](int t, long long i0, long long i1) {
            std::vector<unsigned>& h = hist[t];
            h.assign(nb, 0u);
            for (long long i = i0; i < i1; ++i) h[(keyed[i].first >> shift) & mask]++;
        });
        unsigned run = 0;
        for (size_t b = 0; b < nb; ++b)
            for (int t = 0; t < nt; ++t) { const unsigned v = hist[t][b]; hist[t][b] = run; run += v; }
        pack_parallel_chunks(rc, [&](int t, long long i0, long long i1) {
            std::vector<unsigned>& h = hist[t];
            for (long long i = i0; i < i1; ++i) { const auto& e = keyed[i]; tmp[h[(e.first >> shift) & mask]++] = e; }
        });
        keyed.swap(tmp);
    };
    for (int field = 0; field < 2; ++field)
        for (int done = 0; done < cam_bits; done += 16) pass(32 * field + done, std::min(16, cam_bits - done));
    return 0;
}

namespace plan_detail {
// camera graph: edges = the off-diagonal blocks of S
struct CamGraph {
    std::vector<int> ptr, adj;                 // neighbours by ascending (degree, id)
    int degree(int c) const { return ptr[c + 1] - ptr[c]; }
};
inline CamGraph cam_graph(int Nc, const std::vector<int>& blk_rc, int n_blocks) {
    CamGraph G;
    G.ptr.assign(Nc + 1, 0);
    for (int b = 0; b < n_blocks; ++b) { G.ptr[blk_rc[2 * b] + 1]++; G.ptr[blk_rc[2 * b + 1] + 1]++; }
    for (int c = 0; c < Nc; ++c) G.ptr[c + 1] += G.ptr[c];
    G.adj.resize(G.ptr[Nc]);
    std::vector<int> raw(G.ptr[Nc]), fill(G.ptr.begin(), G.ptr.end() - 1);
    for (int b = 0; b < n_blocks; ++b) {
        const int r = blk_rc[2 * b], c = blk_rc[2 * b + 1];
        raw[fill[r]++] = c; raw[fill[c]++] = r;
    }
    // every list by ascending (degree, id) without a comparison sort per list (millions of entries for a large collection): the
    // cameras are visited in that order and appended to their neighbours' lists
    std::vector<int> by_degree(Nc);
    for (int c = 0; c < Nc; ++c) by_degree[c] = c;
    std::sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return G.degree(a) != G.degree(b) ? G.degree(a) < G.degree(b) : a < b; });
    fill.assign(G.ptr.begin(), G.ptr.end() - 1);
    for (int v : by_degree)
        for (int q = G.ptr[v]; q < G.ptr[v + 1]; ++q) G.adj[fill[raw[q]]++] = v;
    return G;
}
// Reverse Cuthill-McKee order of the camera graph (edges = the off-diagonal blocks of S): breadth-first from a pseudo-peripheral
// camera, neighbours by ascending degree, reversed; components one after the other.  Returns cameras in elimination order.
// Deterministic (ties by camera id), O(edges log degree).
inline std::vector<int> rcm_order(const CamGraph& G) {
    const int Nc = (int)G.ptr.size() - 1;
    const std::vector<int>& ptr = G.ptr;
    const std::vector<int>& adj = G.adj;
    auto degree = [&](int c) { return ptr[c + 1] - ptr[c]; };
    std::vector<int> order; order.reserve(Nc);
    std::vector<int> lvl(Nc, -1), queue;
    std::vector<char> done(Nc, 0);
    // breadth-first search from `root` over the cameras not yet ordered; returns the last camera reached and the depth
    auto bfs = [&](int root, int* depth) {
        queue.clear(); queue.push_back(root); lvl[root] = 0;
        size_t head = 0;
        while (head < queue.size()) {
            const int u = queue[head++];
            for (int q = ptr[u]; q < ptr[u + 1]; ++q) { const int v = adj[q]; if (!done[v] && lvl[v] < 0) { lvl[v] = lvl[u] + 1; queue.push_back(v); } }
        }
        int last = queue.back();
        *depth = lvl[last];
        // among the deepest cameras take the one of smallest degree (George-Liu)
        for (size_t i = queue.size(); i-- > 0 && lvl[queue[i]] == *depth;)
            if (degree(queue[i]) < degree(last) || (degree(queue[i]) == degree(last) && queue[i] < last)) last = queue[i];
        for (int u : queue) lvl[u] = -1;
        return last;
    };
    std::vector<int> by_degree(Nc);
    for (int c = 0; c < Nc; ++c) by_degree[c] = c;
    std::sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return degree(a) != degree(b) ? degree(a) < degree(b) : a < b; });
    for (int start : by_degree) {
        if (done[start]) continue;
        int root = start, depth = -1;
        for (int iter = 0; iter < 8; ++iter) {            // pseudo-peripheral camera of this component
            int d2 = 0;
            const int far = bfs(root, &d2);
            if (d2 <= depth) break;
            depth = d2; root = far;
        }
        queue.clear(); queue.push_back(root); done[root] = 1;
        size_t head = 0;
        while (head < queue.size()) {
            const int u = queue[head++];
            order.push_back(u);
            for (int q = ptr[u]; q < ptr[u + 1]; ++q) { const int v = adj[q]; if (!done[v]) { done[v] = 1; queue.push_back(v); } }
        }
    }
    std::reverse(order.begin(), order.end());
    return order;
}

// ---- nested dissection of an unordered camera graph (round 4)
// The reverse Cuthill-McKee order of a photo collection is a CHAIN: one tile column per elimination-tree level (config 5's shape:
// 750 dependent columns per factorisation), and on a ring of landmarks the breadth-first search runs both ways round, so the band
// is twice as wide as the coupling.  George's automatic nested dissection instead: breadth-first levels from a pseudo-peripheral
// camera, the level that balances the two sides best (and is small) becomes the separator — thinned to the cameras that really
// touch the far side —, the two sides are dissected in turn, a part of <= `leaf` cameras gets its own reverse Cuthill-McKee order,
// children first, separator last.  The parts of one depth are independent: the factorisation of the same collection has ~160
// levels of several columns each instead of 750 of one, with no more (here: 15 % fewer) tile products.  Deterministic (ties by id).
struct NdState {
    const CamGraph& G;
    std::vector<int> label, lvl, queue;        // label: the part a camera belongs to; lvl: -1 outside a search
    int next_label = 1;
    int leaf;
    int pp_iters = 4;                          // searches for the pseudo-peripheral start of a part's own order (rcm_part)
    std::vector<std::vector<int>>* groups;
    NdState(const CamGraph& g, int Nc, int leaf_, std::vector<std::vector<int>>* out) : G(g), label(Nc, 0), lvl(Nc, -1), leaf(leaf_), groups(out) {}
    // breadth-first levels over the cameras of part `id` from `root`; the visiting order stays in `queue` (lvl is NOT reset)
    void bfs(int root, int id) {
        queue.clear(); queue.push_back(root); lvl[root] = 0;
        for (size_t head = 0; head < queue.size(); ++head) {
            const int u = queue[head];
            for (int q = G.ptr[u]; q < G.ptr[u + 1]; ++q) { const int v = G.adj[q]; if (label[v] == id && lvl[v] < 0) { lvl[v] = lvl[u] + 1; queue.push_back(v); } }
        }
    }
    void clear_levels() { for (int u : queue) lvl[u] = -1; }
    // reverse Cuthill-McKee order of one part (its components one after the other), appended as one group
    void rcm_part(std::vector<int> nodes, int id) {
        std::sort(nodes.begin(), nodes.end(), [&](int a, int b) { return G.degree(a) != G.degree(b) ? G.degree(a) < G.degree(b) : a < b; });
        std::vector<int> order; order.reserve(nodes.size());
        for (int start : nodes) {
            if (label[start] != id) continue;
            int root = start, depth = -1;
            for (int iter = 0; iter < pp_iters; ++iter) {
                bfs(root, id);
                int far = queue.back(); const int d2 = lvl[far];
                for (size_t i = queue.size(); i-- > 0 && lvl[queue[i]] == d2;)
                    if (G.degree(queue[i]) < G.degree(far) || (G.degree(queue[i]) == G.degree(far) && queue[i] < far)) far = queue[i];
                clear_levels();
                if (d2 <= depth) break;
                depth = d2; root = far;
            }
            bfs(root, id);
            for (int u : queue) { order.push_back(u); label[u] = -1; }      // (ordered: leaves the part)
            clear_levels();
        }
        std::reverse(order.begin(), order.end());
        groups->push_back(order);
    }
    void dissect(std::vector<int> nodes, int id) {
        if ((int)nodes.size() <= leaf) { rcm_part(std::move(nodes), id); return; }
        std::sort(nodes.begin(), nodes.end());
        // connected components of the part first (the search from its camera of smallest degree doubles as the first
        // pseudo-peripheral search below)
        int root = nodes[0];
        for (int u : nodes) if (G.degree(u) < G.degree(root)) root = u;
        {
            bfs(root, id);
            if (queue.size() < nodes.size()) {
                clear_levels();
                std::vector<std::vector<int>> comps;
                std::vector<int> ids;
                for (int start : nodes) {
                    if (label[start] != id) continue;
                    bfs(start, id);
                    const int nid = next_label++;
                    comps.emplace_back(queue); ids.push_back(nid);
                    for (int u : queue) { label[u] = nid; lvl[u] = -1; }
                }
                for (size_t q = 0; q < comps.size(); ++q) dissect(std::move(comps[q]), ids[q]);
                return;
            }
        }
        // pseudo-peripheral camera of the (connected) part
        int depth = -1;
        for (int iter = 0; iter < 3; ++iter) {
            if (iter > 0) bfs(root, id);
            int far = queue.back(); const int d2 = lvl[far];
            for (size_t i = queue.size(); i-- > 0 && lvl[queue[i]] == d2;)
                if (G.degree(queue[i]) < G.degree(far) || (G.degree(queue[i]) == G.degree(far) && queue[i] < far)) far = queue[i];
            if (d2 <= depth) break;                        // (levels of `root` stay)
            depth = d2;
            if (iter == 2) break;
            clear_levels();
            root = far;
        }
        if (lvl[root] != 0) { clear_levels(); bfs(root, id); }
        const int D = lvl[queue.back()];
        if (D < 2) { clear_levels(); rcm_part(std::move(nodes), id); return; }
        std::vector<long long> cnt(D + 1, 0);
        for (int u : queue) cnt[lvl[u]]++;
        long long below = cnt[0], best = -1; int lsep = 1;
        for (int l = 1; l < D; ++l) {
            const long long above = (long long)nodes.size() - below - cnt[l];
            const long long score = 2 * cnt[l] + (below > above ? below - above : above - below);
            if (best < 0 || score < best) { best = score; lsep = l; }
            below += cnt[l];
        }
        std::vector<int> lo, hi, sep;
        for (int u : queue) {
            if (lvl[u] < lsep) lo.push_back(u);
            else if (lvl[u] > lsep) hi.push_back(u);
            else {
                bool touches = false;
                for (int q = G.ptr[u]; q < G.ptr[u + 1] && !touches; ++q) { const int v = G.adj[q]; touches = (label[v] == id && lvl[v] == lsep + 1); }
                (touches ? sep : lo).push_back(u);
            }
        }
        clear_levels();
        const int id_lo = next_label++, id_hi = next_label++, id_sep = next_label++;
        for (int u : lo) label[u] = id_lo;
        for (int u : hi) label[u] = id_hi;
        for (int u : sep) label[u] = id_sep;
        dissect(std::move(lo), id_lo);
        dissect(std::move(hi), id_hi);
        rcm_part(std::move(sep), id_sep);
    }
};
// groups (each starts on a tile boundary) in elimination order
inline std::vector<std::vector<int>> nd_groups(const CamGraph& G, int leaf, int pp_iters = 4) {
    std::vector<std::vector<int>> groups;
    const int Nc = (int)G.ptr.size() - 1;
    NdState st(G, Nc, leaf, &groups);
    st.pp_iters = pp_iters;
    std::vector<int> all(Nc);
    for (int c = 0; c < Nc; ++c) all[c] = c;
    if (Nc > 0) st.dissect(std::move(all), 0);
    return groups;
}
// tile products and elimination-tree levels of the tile pattern for cameras laid out group by group (10 per tile, every group on a
// tile boundary); products = -1 beyond `budget`
inline long long count_group_products(int Nc, const std::vector<int>& blk_rc, int n_blocks, const std::vector<std::vector<int>>& groups,
                                      int cams_per_tile, long long budget, int* n_levels) {
    std::vector<int> tile_of(Nc, 0);
    int T = 0;
    for (const auto& g : groups) {
        for (size_t q = 0; q < g.size(); ++q) tile_of[g[q]] = T + (int)q / cams_per_tile;
        T += ((int)g.size() + cams_per_tile - 1) / cams_per_tile;
    }
    std::vector<char> nz((size_t)T * T, 0);
    for (int b = 0; b < n_blocks; ++b) {
        const int ti = tile_of[blk_rc[2 * b]], tj = tile_of[blk_rc[2 * b + 1]];
        nz[(size_t)std::max(ti, tj) * T + std::min(ti, tj)] = 1;
    }
    long long total = 0;
    std::vector<int> R;
    for (int k = 0; k < T; ++k) {
        R.clear();
        for (int i = k + 1; i < T; ++i) if (nz[(size_t)i * T + k]) R.push_back(i);
        total += (long long)R.size() * ((long long)R.size() + 1) / 2;
        if (total > budget) return -1;
        for (size_t a = 0; a < R.size(); ++a)
            for (size_t b2 = 0; b2 <= a; ++b2) nz[(size_t)R[a] * T + R[b2]] = 1;
    }
    std::vector<int> level(T, 0);
    int nl = 0;
    for (int k = 0; k < T; ++k) {
        int lv = 0;
        for (int j = 0; j < k; ++j) if (nz[(size_t)k * T + j] && level[j] + 1 > lv) lv = level[j] + 1;
        level[k] = lv; nl = std::max(nl, lv + 1);
    }
    if (n_levels) *n_levels = nl;
    return total;
}

// Symbolic factorisation of the 64x64 tile pattern for the cameras in `order` (10 per tile): number of tile products
// sum_k |R_k| (|R_k| + 1) / 2 of one factorisation, or -1 as soon as it exceeds `budget`.
inline long long count_tile_products(int Nc, const std::vector<int>& blk_rc, int n_blocks, const std::vector<int>& order, int cams_per_tile,
                                     long long budget) {
    const int T = (Nc + cams_per_tile - 1) / cams_per_tile;
    std::vector<int> tile_of(Nc);
    for (int r = 0; r < Nc; ++r) tile_of[order[r]] = r / cams_per_tile;
    std::vector<char> nz((size_t)T * T, 0);
    for (int b = 0; b < n_blocks; ++b) {
        const int ti = tile_of[blk_rc[2 * b]], tj = tile_of[blk_rc[2 * b + 1]];
        nz[(size_t)std::max(ti, tj) * T + std::min(ti, tj)] = 1;
    }
    long long total = 0;
    std::vector<int> R;
    for (int k = 0; k < T; ++k) {
        R.clear();
        for (int i = k + 1; i < T; ++i) if (nz[(size_t)i * T + k]) R.push_back(i);
        total += (long long)R.size() * ((long long)R.size() + 1) / 2;
        if (total > budget) return -1;
        for (size_t a = 0; a < R.size(); ++a)
            for (size_t b2 = 0; b2 <= a; ++b2) nz[(size_t)R[a] * T + R[b2]] = 1;
    }
    return total;
}

inline void dissect(int lo, int hi, int w, int leaf, int cap, const std::vector<int>& extra, const std::vector<int>& keep,
                    std::vector<std::vector<int>>& g) {
    if (hi <= lo) { if (!extra.empty()) g.push_back(extra); return; }
    int nsep = std::max(1, (cap - (int)extra.size()) / w);
    while (nsep > 1 && (hi - lo - nsep * w) < (nsep + 1) * leaf / 2) --nsep;
    if (hi - lo <= leaf + w) {
        std::vector<int> v;
        for (int c = lo; c < hi; ++c) v.push_back(keep[c]);
        g.push_back(v);
        if (!extra.empty()) g.push_back(extra);
        return;
    }
    const int total = hi - lo - nsep * w, part = total / (nsep + 1), rem = total % (nsep + 1);
    std::vector<int> seps;
    int cur = lo;
    for (int s = 0; s <= nsep; ++s) {
        const int len = part + (s < rem ? 1 : 0);
        dissect(cur, cur + len, w, leaf, cap, std::vector<int>(), keep, g);
        cur += len;
        if (s < nsep) { for (int c = cur; c < cur + w; ++c) seps.push_back(keep[c]); cur += w; }
    }
    seps.insert(seps.end(), extra.begin(), extra.end());
    g.push_back(seps);
}
}  // namespace plan_detail

// `pattern`: union of all ranks' camera pairs (sorted, unique) or nullptr for the local pairs.
// Limits (checked BEFORE the symbolic factorisation, whose work and memory grow with the cube of the tile count when the
// pattern fills in): more than `max_dense_unknowns` camera unknowns are only planned when the band / ring ordering applies,
// and then only while the packed tile storage of the factor (non-zero 64x64 tiles) stays within `max_tile_bytes`; XRSFM_BA_ETOOBIG otherwise.
// What the device-side key generation (ba_pack_dev.h: device_keys) hands to the plan instead of the key list: the blocks, the
// sizes, and the launch buckets of the S assembly; slot_pair_ptr / pair_dst / blk_ptr / pairs_items stay on the device.
struct PlanPrebuilt {
    int n_pairs = 0, n_writes = 0, n_pair_writes = 0;
    std::vector<int> blk_rc;
    int gram_n[8] = {0}, n_other = 0; size_t gram_shm[8] = {0};
};

inline int chol_plan_build(const Packed& k, const std::vector<int>& spp, const PairKeys& keyed,
                           const std::vector<unsigned long long>* pattern, CholPlan& P,
                           long long max_dense_unknowns = INT64_MAX, unsigned long long max_tile_bytes = UINT64_MAX, int cam_width = 6,
                           const PlanPrebuilt* pre = nullptr) {
    const int CW = cam_width, CPT = (cam_width == 6) ? kCamsPerTile : kCamsPerTileWide;
    const int Nc = k.n_cams, ns = k.n_slots;
    PhaseTimer timer("plan");
    P = CholPlan();
    if (!pre) P.spp = spp;
    P.cam_width = CW; P.cams_per_tile = CPT;
    P.n = CW * Nc;
    P.n_pairs = pre ? pre->n_pairs : spp[ns] + k.n_gt_cells;      // pair_dst: observation pairs | cells of the Gram tiles' tables
    P.n_writes = pre ? pre->n_writes : (int)keyed.size();
    if (pre) P.n_pair_writes = pre->n_pair_writes;
    else {
        const int n_obs_pairs = spp[ns];
        long long np = 0;
        for (const auto& kv : keyed) np += (kv.second < n_obs_pairs);
        P.n_pair_writes = (int)np;
    }
    std::vector<unsigned long long> blk_keys;
    if (pre) {
        if (pattern) return XRSFM_BA_EINTERNAL;                  // (device keys are local-pattern only)
        P.blk_rc = pre->blk_rc;
    } else {
    if (pattern) blk_keys = *pattern;
    else if (P.n_writes < 2000000)
        for (int i = 0; i < P.n_writes; ++i)
            if (i == 0 || keyed[i].first != keyed[i - 1].first) blk_keys.push_back(keyed[i].first);
    P.pair_dst.assign(P.n_pairs, -1);
    if (!pattern && P.n_writes >= 2000000) {
        // large key lists (tens of millions for an unordered collection with long tracks): the destination of the i-th key is i,
        // and the block boundaries are the positions where the key changes — pieces of the list in parallel, same result
        const unsigned hw = std::thread::hardware_concurrency();
        const int nt = hw < 2 ? 1 : (int)std::min<unsigned>(16u, hw);
        std::vector<long long> rc(nt + 1);
        for (int t = 0; t <= nt; ++t) rc[t] = (long long)((size_t)P.n_writes * (size_t)t / (size_t)nt);
        std::vector<int> heads(nt + 1, 0);
        pack_parallel_chunks(rc, [&](int t, long long i0, long long i1) {
            int h = 0;
            for (long long i = i0; i < i1; ++i) {
                h += (i == 0 || keyed[i].first != keyed[i - 1].first);
                P.pair_dst[keyed[i].second] = (int)i;
            }
            heads[t + 1] = h;
        });
        for (int t = 0; t < nt; ++t) heads[t + 1] += heads[t];
        const int nb = heads[nt];
        P.blk_ptr.assign((size_t)nb + 1, 0); P.blk_rc.assign(2 * (size_t)nb, 0);
        pack_parallel_chunks(rc, [&](int t, long long i0, long long i1) {
            int b = heads[t];
            for (long long i = i0; i < i1; ++i)
                if (i == 0 || keyed[i].first != keyed[i - 1].first) {
                    P.blk_ptr[b] = (int)i;
                    P.blk_rc[2 * (size_t)b] = (int)(keyed[i].first >> 32); P.blk_rc[2 * (size_t)b + 1] = (int)(keyed[i].first & 0xffffffffu);
                    ++b;
                }
        });
        P.blk_ptr[nb] = P.n_writes;
    } else {
        int i = 0;
        for (const unsigned long long key : blk_keys) {
            P.blk_ptr.push_back(i);
            P.blk_rc.push_back((int)(key >> 32)); P.blk_rc.push_back((int)(key & 0xffffffffu));
            while (i < P.n_writes && keyed[i].first == key) { P.pair_dst[keyed[i].second] = i; ++i; }
        }
        if (i != P.n_writes) return XRSFM_BA_EINVAL;    // a local pair that is missing from the supplied pattern
        P.blk_ptr.push_back(P.n_writes);
    }
    }
    const int n_blocks = P.n_blocks = pre ? (int)P.blk_rc.size() / 2 : (int)P.blk_ptr.size() - 1;
    const std::vector<int>& blk_rc = P.blk_rc;
    timer.mark("  blocks + destinations");

    // ---- elimination order.  Band width w of the camera graph (circular distance of the camera pairs).  A few
    // long-range pairs (loop closures, re-observed landmarks) must not destroy the band structure of a sequential
    // reconstruction: w covers all but 0.5 % of the pairs, the cameras at the ends of the remaining pairs become "hubs"
    // that are eliminated last (at most max(24, Nc/16) of them, else the natural order is kept); every other dependency they cause is handled exactly by the symbolic factorisation.
    int w = 0; bool wrap = false;
    std::vector<char> is_hub(Nc, 0);
    {
        std::vector<int> dist(n_blocks);
        for (int b = 0; b < n_blocks; ++b) {
            const int dlin = blk_rc[2 * b] - blk_rc[2 * b + 1];
            dist[b] = std::min(dlin, Nc - dlin);
        }
        if (n_blocks > 0) {         // the 99.5th percentile: a selection, not a sort (1.65 M blocks at config 5's shape: 100 ms of sorting)
            std::vector<int> sel = dist;
            const size_t q = (size_t)((n_blocks - 1) * 0.995);
            std::nth_element(sel.begin(), sel.begin() + q, sel.end());
            w = sel[q];
        }
        // ... or, on small problems where a handful of closures is more than 0.5 % of the pairs: the smallest band that
        // leaves at most `cap` hub cameras
        std::vector<int> reach(Nc, 0);
        for (int b = 0; b < n_blocks; ++b) {
            reach[blk_rc[2 * b]] = std::max(reach[blk_rc[2 * b]], dist[b]);
            reach[blk_rc[2 * b + 1]] = std::max(reach[blk_rc[2 * b + 1]], dist[b]);
        }
        std::sort(reach.begin(), reach.end());
        const int cap = std::max(24, Nc / 16);
        if (Nc > cap) w = std::min(w, reach[Nc - 1 - cap]);
        for (int b = 0; b < n_blocks; ++b) {
            const int dlin = blk_rc[2 * b] - blk_rc[2 * b + 1];
            if (dist[b] > w) { is_hub[blk_rc[2 * b]] = 1; is_hub[blk_rc[2 * b + 1]] = 1; }
            else if (dist[b] != dlin) wrap = true;                   // a band pair that closes the ring
        }
        for (int c = 0; c < Nc; ++c) P.n_hubs += is_hub[c];
    }
    P.band = w;
    std::vector<std::vector<int>> groups;   // each group starts on a tile boundary
    if (w >= 1 && 16 * w <= Nc && P.n_hubs <= std::max(24, Nc / 16)) {
        // multi-way nested dissection of the path/ring: a tree node cuts its range with g separators of w cameras each
        // that share ONE tile (g*w <= 10 cameras), so the elimination tree has depth log_{g+1} instead of log_2
        P.ordering = 1;
        std::vector<int> keep;
        for (int c = 0; c < Nc; ++c) if (!is_hub[c]) keep.push_back(c);
        const int nk = (int)keep.size();
        std::vector<int> root;
        const int wr = std::min(w, nk);
        if (wrap) for (int c = 0; c < wr; ++c) root.push_back(keep[c]);   // closes the ring: eliminated with the top separators
        plan_detail::dissect(wrap ? wr : 0, nk, w, 2 * CPT, CPT, root, keep, groups);
        if (P.n_hubs > 0) {
            std::vector<int> hubs;
            for (int c = 0; c < Nc; ++c) if (is_hub[c]) hubs.push_back(c);
            groups.push_back(hubs);
        }
    } else {
        P.n_hubs = 0;
        std::vector<int> all;
        for (int c = 0; c < Nc; ++c) all.push_back(c);
        // Unordered collections (BASELINE config 5, rec_1dsfm.cc:66-98: internet photos have no temporal order) are not random
        // graphs: photos cluster around landmarks, and a bandwidth-reducing order turns the dense reduced camera matrix of
        // the natural order into a band whose fill is a small fraction of it (synthetic collection of 7500 photos: 15 % of
        // the tiles, 6e11 instead of 4e13 flop per factorisation) — the exact solve Ceres' SPARSE_SCHUR performs then stays
        // affordable far beyond `max_dense_unknowns`.  Reverse Cuthill-McKee of the camera graph; taken only if the symbolic
        // factorisation says it pays (<= 60 % of the natural order's tile products, within the work budget), from 48 tile
        // columns (480 cameras) on; random visibility (no clusters) keeps the natural order / the PCG path.
        const int Tn = (Nc + CPT - 1) / CPT;
        const char* rcm_env = std::getenv("XRSFM_BA_RCM");
        // (beyond kPlanMaxTiles tile columns no order is accepted below: the probe — a Tn x Tn map and an O(Tn^2) walk — is skipped,
        //  so an oversized unordered problem is refused before any T^2 work and AUTO falls back to the PCG at once)
        if (Tn >= 48 && Tn <= kPlanMaxTiles && !(rcm_env && rcm_env[0] == '0')) {
            const long long budget = 12000000;              // tile products of one factorisation: 6.3e12 flop, ~0.25 s
            const plan_detail::CamGraph G = plan_detail::cam_graph(Nc, blk_rc, n_blocks);
            timer.mark("    camera graph");
            const std::vector<int> rcm = plan_detail::rcm_order(G);
            timer.mark("    reverse Cuthill-McKee");
            const long long pr = plan_detail::count_tile_products(Nc, blk_rc, n_blocks, rcm, CPT, budget);
            const long long pn = ((long long)CW * Nc <= max_dense_unknowns) ? plan_detail::count_tile_products(Nc, blk_rc, n_blocks, all, CPT, budget) : -1;
            // natural order not allowed (beyond the dense limit): its pattern is taken as full, Tn^3 / 6 products
            const long long pn_eff = pn >= 0 ? pn : (long long)Tn * Tn * Tn / 6;
            if (pr >= 0 && pr * 10 <= pn_eff * 6) { all = rcm; P.ordering = 2; }      // (random visibility: no gain -> natural order / PCG)
            // ... and from 96 tile columns on, nested dissection of the same graph (plan_detail::nd_groups) where its elimination tree is
            // at most half as deep as the chain and its fill no worse than 1.15 x: the level schedule then factors several columns per
            // launch (config 5's shape: 169 levels of 759 columns, 1.02 M tile products against 1.21 M).  XRSFM_BA_ND=0: keep the chain.
            const char* nd_env = std::getenv("XRSFM_BA_ND");
            if (P.ordering == 2 && Tn >= 96 && !(nd_env && nd_env[0] == '0')) {
                // (XRSFM_BA_ND_LEAF: developer aid of tools/t_sweep.py — the leaf size trades fill against the depth of the tree)
                const int leaf = std::getenv("XRSFM_BA_ND_LEAF") ? std::max(2 * CPT, std::atoi(std::getenv("XRSFM_BA_ND_LEAF"))) : std::max(200, std::min(1200, Nc / 6));   // (config 5: 1200 of 7500)
                timer.mark("    symbolic counts");
                // (the fill of a part's own order depends on which end of it the search happens to start from — 1.08 / 1.11 M tile
                //  products at config 5's shape with 1 / 4 pseudo-peripheral searches —: both are formed, the symbolic count decides)
                std::vector<std::vector<int>> nd, nd1;
                int nd_levels = 0, nd_tiles = 0, lv1 = 0;
                long long pd = -1, pd1 = -1;
                {   // (the second candidate on a thread of its own: the graph is shared read-only)
                    std::exception_ptr err;
                    std::thread other([&] {
                        try {
                            nd1 = plan_detail::nd_groups(G, leaf, 1);
                            pd1 = plan_detail::count_group_products(Nc, blk_rc, n_blocks, nd1, CPT, budget, &lv1);
                        } catch (...) { err = std::current_exception(); }
                    });
                    try {
                        nd = plan_detail::nd_groups(G, leaf, 4);
                        pd = plan_detail::count_group_products(Nc, blk_rc, n_blocks, nd, CPT, budget, &nd_levels);
                    } catch (...) { other.join(); throw; }
                    other.join();
                    if (err) std::rethrow_exception(err);
                }
                if (pd1 >= 0 && (pd < 0 || pd1 < pd)) { nd.swap(nd1); pd = pd1; nd_levels = lv1; }
                for (const auto& g : nd) nd_tiles += ((int)g.size() + CPT - 1) / CPT;
                timer.mark("    nested dissection");
                if (std::getenv("XRSFM_BA_PLAN_VERBOSE"))
                    fprintf(stderr, "[plan] reverse Cuthill-McKee: %d tile columns, %lld tile products | nested dissection (leaf %d): %zu groups, %d tile columns, %lld products, %d levels\n",
                            Tn, pr, leaf, nd.size(), nd_tiles, pd, nd_levels);
                if (nd_tiles <= kPlanMaxTiles && pd >= 0 && pd * 100 <= pr * 115 && 2 * nd_levels <= nd_tiles) { groups = std::move(nd); P.ordering = 3; }
            }
        }
        if (P.ordering != 3) groups.push_back(all);
    }
    timer.mark("  elimination order");
    P.cam_off.assign(Nc, 0);
    int T = 0;
    for (const auto& g : groups) {
        for (size_t q = 0; q < g.size(); ++q) P.cam_off[g[q]] = kPlanTile * (T + (int)q / CPT) + CW * ((int)q % CPT);
        const int nt = ((int)g.size() + CPT - 1) / CPT;
        for (int q = 0; q < nt; ++q) P.tile_rows.push_back(CW * std::min(CPT, (int)g.size() - q * CPT));
        T += nt;
    }
    if (T == 0) { T = 1; P.tile_rows.push_back(0); }
    P.T = T; P.n_pad = T * kPlanTile;
    P.tile_cam.assign((size_t)T * CPT, -1);
    for (int c = 0; c < Nc; ++c) P.tile_cam[(size_t)(P.cam_off[c] / kPlanTile) * CPT + (P.cam_off[c] % kPlanTile) / CW] = c;
    // beyond the dense limit only with an order that keeps the factor sparse, and while the T x T tile maps stay small (4096 tile
    // columns = 262 000 unknowns); the bytes of the packed tile storage are checked after the symbolic factorisation
    if ((long long)CW * Nc > max_dense_unknowns && (P.ordering == 0 || T > kPlanMaxTiles)) return XRSFM_BA_ETOOBIG;

    // ---- tile pattern + symbolic factorisation
    std::vector<char> nz((size_t)T * T, 0);
    for (int t = 0; t < T; ++t) nz[(size_t)t * T + t] = 1;
    for (int b = 0; b < n_blocks; ++b) {
        const int ti = P.cam_off[blk_rc[2 * b]] / kPlanTile, tj = P.cam_off[blk_rc[2 * b + 1]] / kPlanTile;
        nz[(size_t)std::max(ti, tj) * T + std::min(ti, tj)] = 1;
    }
    P.cols_off.assign(T + 1, 0);
    for (int kk = 0; kk < T; ++kk) {
        std::vector<int> R;
        for (int i = kk + 1; i < T; ++i) if (nz[(size_t)i * T + kk]) R.push_back(i);
        P.tile_products += (long long)R.size() * ((long long)R.size() + 1) / 2;
        for (size_t a = 0; a < R.size(); ++a)
            for (size_t b2 = 0; b2 <= a; ++b2) nz[(size_t)R[a] * T + R[b2]] = 1;
    }
    for (int kk = 0; kk < T; ++kk) {
        for (int j = 0; j < kk; ++j) if (nz[(size_t)kk * T + j]) P.cols_flat.push_back(j);
        P.cols_off[kk + 1] = (int)P.cols_flat.size();
        for (int j = 0; j <= kk; ++j) if (nz[(size_t)kk * T + j]) { P.tiles_nz.push_back(kk); P.tiles_nz.push_back(j); }
    }
    P.bw2_off.assign(1, 0);
    for (int kk = T - 1; kk >= 1; kk -= 2) {
        for (int j = 0; j < kk - 1; ++j) {
            const int f = (nz[(size_t)kk * T + j] ? 1 : 0) | (nz[(size_t)(kk - 1) * T + j] ? 2 : 0);
            if (f) { P.bw2_ent.push_back(j); P.bw2_ent.push_back(f); }
        }
        P.bw2_off.push_back((int)P.bw2_ent.size() / 2);
        P.bw2_link.push_back(nz[(size_t)kk * T + kk - 1] ? 1 : 0);
    }
    P.n_tiles_nz = (int)P.tiles_nz.size() / 2;
    timer.mark("  symbolic factorisation");
    std::vector<int> tile_id((size_t)T * T, -1);
    for (int q = 0; q < P.n_tiles_nz; ++q) tile_id[(size_t)P.tiles_nz[2 * q] * T + P.tiles_nz[2 * q + 1]] = q;
    if ((long long)CW * Nc > max_dense_unknowns &&
        ((unsigned long long)P.n_tiles_nz + 1) * (unsigned long long)(kPlanTile * kPlanTile + kPlanTile) * sizeof(double) > max_tile_bytes) return XRSFM_BA_ETOOBIG;
    P.tile_map.resize((size_t)T * T);
    for (size_t e = 0; e < P.tile_map.size(); ++e) P.tile_map[e] = tile_id[e] >= 0 ? tile_id[e] : P.n_tiles_nz;
    {
        std::vector<int> cnt(P.n_tiles_nz + 1, 0);
        auto tile_of_block = [&](int b) {
            const int ti = P.cam_off[blk_rc[2 * b]] / kPlanTile, tj = P.cam_off[blk_rc[2 * b + 1]] / kPlanTile;
            return tile_id[(size_t)std::max(ti, tj) * T + std::min(ti, tj)];
        };
        auto tile_of_cam = [&](int c) { const int t = P.cam_off[c] / kPlanTile; return tile_id[(size_t)t * T + t]; };
        for (int b = 0; b < n_blocks; ++b) cnt[tile_of_block(b) + 1]++;
        for (int c = 0; c < Nc; ++c) cnt[tile_of_cam(c) + 1]++;
        for (int q = 0; q < P.n_tiles_nz; ++q) cnt[q + 1] += cnt[q];
        P.tf_ptr = cnt;
        P.tf_ent.assign(cnt[P.n_tiles_nz], 0);
        std::vector<int> fill(cnt.begin(), cnt.end() - 1);
        for (int c = 0; c < Nc; ++c) P.tf_ent[fill[tile_of_cam(c)]++] = -(c + 1);
        for (int b = 0; b < n_blocks; ++b) P.tf_ent[fill[tile_of_block(b)]++] = b;
    }

    timer.mark("  fill lists");
    // ---- elimination-tree levels of the (filled) tile pattern and the per-level work lists
    std::vector<int> level(T, 0);
    int n_levels = 0;
    for (int kk = 0; kk < T; ++kk) {
        int lv = 0;
        for (int j = 0; j < kk; ++j) if (nz[(size_t)kk * T + j]) lv = std::max(lv, level[j] + 1);
        level[kk] = lv;
        n_levels = std::max(n_levels, lv + 1);
    }
    // Deep trees (unordered / dense patterns: about one panel per level): "panel schedule" = left-looking updates per panel
    // (split into chunks), pivot + triangular solve per panel, push-form backward substitution
    bool panel_ll = (2 * n_levels > T);
    const int panel_min_chunk = 4, panel_chunks = 1024;      // (measured on config D: 512 / 768 / 1400 / 2048 chunks are 3-10 % slower)
    P.panel_ll = panel_ll;
    P.lv_cptr.assign(1, 0); P.lv_bptr.assign(1, 0);
    P.lv_k_off.assign(n_levels + 1, 0); P.lv_tgt_off.assign(n_levels + 1, 0);
    P.sp_chunk_off.assign(n_levels + 1, 0); P.sp_rt_off.assign(n_levels + 1, 0);
    P.mp_off.assign(n_levels + 1, 0);
    P.fz_off.assign(n_levels + 1, 0); P.fz_dptr.assign(1, 0);
    P.md_off.assign(1, 0);
    std::vector<int> level_cols(n_levels, 0), level_first(n_levels, -1);
    for (int kk = 0; kk < T; ++kk) { if (level_cols[level[kk]]++ == 0) level_first[level[kk]] = kk; }
    // look-ahead needs level == column (the dependencies of the two streams are stated per column) and the fused kernels
    // ... and pays where a column's update is short of work for the whole chip (config T: ~1600 tile products per column, U: ~340);
    // a dense factor of >= 96 columns (config D: ~6700 per column) keeps the macro-tile panels, whose launches need the CUs to
    // themselves (two 74 KB workgroups per CU; the one-launch-per-column kernel holds 144 KB of LDS per workgroup)
    bool lookahead = panel_ll && n_levels == T && T >= 8 && (T < 96 || P.tile_products <= (long long)4096 * T);
    if (const char* fl = std::getenv("XRSFM_BA_LOOKAHEAD")) lookahead = panel_ll && n_levels == T && T >= 8 && fl[0] != '0';
    if (const char* fl = std::getenv("XRSFM_BA_FUSED")) lookahead = lookahead && fl[0] != '0';
    P.lookahead = lookahead;
    constexpr int kLookDepth = 1;             // columns the fused factor kernel adds itself (k - 1); k - 2 arrives as one late partial per tile
    // contributions below first_j[k] reach column k through partial products (macro-tile launch, or — look-ahead — chunks of a
    // split level too), those from first_j[k] on inside the fused factor kernel
    std::vector<int> first_j(T, 0);
    std::vector<char> later_of_panel(T, 0);   // second (third, ...) column of a macro panel: no partial launch of its own
    if (lookahead) for (int kk = 0; kk < T; ++kk) first_j[kk] = std::max(0, kk - kLookDepth);
    const int macro_chunks = 512;
    // macro tiles pay off from ~100 tile columns (config U, 45 columns: 25.7 ms without, 28.1 ms with); developer switches
    bool macro_on = T >= 96;
    if (const char* fl = std::getenv("XRSFM_BA_PANEL_MACRO")) macro_on = fl[0] == '1';
    if (lookahead) macro_on = false;        // (k_panel_slot carries chunks of split levels only)
    // (wider panels were measured on config D: 4 / 8 columns leave the update time where it is and lengthen the lists of the
    //  fused factor kernel: 237 -> 250 / 266 ms)
    const int panel_cols_max = std::getenv("XRSFM_BA_PANEL_COLS") ? std::atoi(std::getenv("XRSFM_BA_PANEL_COLS")) : 2;
    // number of consecutive single-column levels lv, lv+1, ... (columns k0, k0+1, ...) that can share one macro-tile launch: the
    // pattern is full to the left of k0 and below; an even count <= panel_cols_max, 0 = none
    auto dense_panel = [&](int lv) {
        if (!panel_ll || !macro_on || level_cols[lv] != 1) return 0;
        const int k0 = level_first[lv];
        if (k0 < (lookahead ? 2 + 2 * kLookDepth : 2) || later_of_panel[k0]) return 0;
        for (int i = k0; i < T; ++i)
            for (int j = 0; j < k0; ++j) if (!nz[(size_t)i * T + j]) return 0;
        int np = 0;
        while (np < panel_cols_max && lv + np < n_levels && level_cols[lv + np] == 1 && level_first[lv + np] == k0 + np) {
            bool full = true;
            for (int i = k0 + np; i < T && full; ++i) full = nz[(size_t)i * T + k0 + np] != 0;
            if (!full) break;
            ++np;
        }
        return np & ~1;
    };
    struct FzEnt { int i, k; };
    std::vector<FzEnt> fz_ents;
    // look-ahead on the level schedule: dissected collections with a deep tree (XRSFM_BA_LA_DEPTH: 0 = off, default 2)
    {
        const char* le = std::getenv("XRSFM_BA_LA_DEPTH");
        const int want = le ? std::max(0, std::min(8, std::atoi(le))) : 2;
        P.la_depth = (P.ordering == 3 && !panel_ll && n_levels >= 16) ? want : 0;
    }
    const bool la_lv = P.la_depth > 0;
    P.sp_e_cnt.assign(n_levels, 0);
    std::vector<int> lv_ne;               // per target (lv_tgt order): early contributions at the head of its list
    for (int lv = 0; lv < n_levels; ++lv) {
        fz_ents.clear();
        const int panel_cols = dense_panel(lv);
        const bool macro = panel_cols > 0;
        // partial products of a macro panel cover j < jhi for every column of the panel
        const int jhi = macro ? (lookahead ? level_first[lv] - kLookDepth : level_first[lv]) : 0;
        for (int q = lookahead ? 0 : 1; q < panel_cols; ++q) { first_j[level_first[lv] + q] = jhi; if (q > 0) later_of_panel[level_first[lv] + q] = 1; }
        for (int kk = 0; kk < T; ++kk) {
            if (level[kk] != lv) continue;
            for (int i = kk; i < T; ++i) {
                if (!nz[(size_t)i * T + kk]) continue;
                std::vector<int> contrib;
                if (lookahead) {        // the list of the partial products that are summed in place: j < k - 2
                    for (int j = 0; j < kk - 2; ++j) if (nz[(size_t)i * T + j] && nz[(size_t)kk * T + j]) contrib.push_back(j);
                } else
                for (int j = first_j[kk]; j < kk; ++j) if (nz[(size_t)i * T + j] && nz[(size_t)kk * T + j]) contrib.push_back(j);
                fz_ents.push_back({i, kk});
                if (contrib.empty()) continue;
                int n_early = 0;
                if (la_lv) {          // early contributions first (stable: ascending j inside both parts)
                    const auto mid = std::stable_partition(contrib.begin(), contrib.end(), [&](int j) { return level[j] <= lv - 1 - P.la_depth; });
                    n_early = (int)(mid - contrib.begin());
                }
                lv_ne.push_back(n_early);
                P.lv_tgt.push_back(i); P.lv_tgt.push_back(kk);
                for (int j : contrib) P.lv_cj.push_back(j);
                P.lv_cptr.push_back((int)P.lv_cj.size());
            }
        }
        P.lv_tgt_off[lv + 1] = (int)P.lv_tgt.size() / 2;
        // split this level?  (few workgroups, each with a long serial list)
        const int g0 = P.lv_tgt_off[lv], g1 = P.lv_tgt_off[lv + 1];
        const int nt = g1 - g0, nc = nt > 0 ? P.lv_cptr[g1] - P.lv_cptr[g0] : 0;
        // (panel schedule: every level with lists worth cutting is split, into chunks of >= kPanelMinChunk products so that the
        //  partial tile a chunk writes stays a small part of its traffic, and into <= ~kPanelChunks chunks per level)
        const bool second = level_cols[lv] == 1 && later_of_panel[level_first[lv]];    // second column of a macro pair: one contribution left
        // (nested dissection of an unordered collection: its levels hold hundreds of targets WITH long lists — every level with lists
        //  worth cutting is split, as on the panel schedule)
        const bool nd_lv = P.ordering == 3 && !panel_ll;
        const bool split = !macro && !second && (lookahead ? nc > 0 : ((panel_ll || nd_lv) ? (nt > 0 && nc > 2 * nt) : (nt > 0 && nt <= 128 && nc > 2 * nt)));
        if (macro) {
            // every macro target's j range is cut into chunks of >= panel_min_chunk steps, ~macro_chunks chunks per level (about two
            // rounds of resident workgroups: measured faster than one round of equal shares, whose partial-tile stores all
            // land at the same moment); a workgroup walks the entries [mp_wg[w], mp_wg[w+1]) - here one each
            // macro targets: column pair cp = (K0 + 2cp, K0 + 2cp + 1) x row pairs from that column pair's own rows downwards
            const int K0 = level_first[lv], ncp = panel_cols / 2;
            struct Tgt { int i0, i1, k0, k1; };
            std::vector<Tgt> tg;
            for (int cp = 0; cp < ncp; ++cp)
                for (int i0 = K0 + 2 * cp; i0 < T; i0 += 2) tg.push_back({i0, (i0 + 1 < T) ? i0 + 1 : -1, K0 + 2 * cp, K0 + 2 * cp + 1});
            const int nm = (int)tg.size();
            const int cs = std::max(panel_min_chunk, (int)(((long long)nm * jhi + macro_chunks - 1) / macro_chunks));
            std::vector<int> pieces(nm, 0);       // partial slots per macro target
            struct Ent { int m, q0, q1, piece; };
            std::vector<Ent> ents;
            std::vector<int> wg_first;
            for (int m = 0; m < nm; ++m)
                for (int q = 0; q < jhi; q += cs) {
                    wg_first.push_back((int)ents.size());
                    ents.push_back({m, q, std::min(jhi, q + cs), pieces[m]++});
                }
            const int W = (int)ents.size();
            wg_first.push_back((int)ents.size());
            std::vector<int> base(nm + 1, 0);
            for (int m = 0; m < nm; ++m) base[m + 1] = base[m] + 4 * pieces[m];
            const int e_off = (int)P.mp_chunk.size() / 8;
            for (const Ent& en : ents) {
                const Tgt& g = tg[en.m];
                const int e[8] = {g.i0, g.i1, g.k0, g.k1, en.q0, en.q1, base[en.m] + en.piece, pieces[en.m]};
                P.mp_chunk.insert(P.mp_chunk.end(), e, e + 8);
            }
            for (int w = 0; w <= W; ++w) P.mp_wg.push_back(e_off + wg_first[w]);
            for (int m = 0; m < nm; ++m) {
                const Tgt& g = tg[m];
                for (int a2 = 0; a2 < 2; ++a2)
                    for (int b2 = 0; b2 < 2; ++b2) {
                        const int ia = a2 ? g.i1 : g.i0, kb = b2 ? g.k1 : g.k0;
                        if (ia < 0 || ia < kb) continue;
                        P.sp_rt.push_back(ia); P.sp_rt.push_back(kb);
                        P.sp_rp.push_back(base[m] + (2 * a2 + b2) * pieces[m]); P.sp_rp.push_back(base[m] + (2 * a2 + b2 + 1) * pieces[m]);
                    }
            }
            P.sp_max_chunks = std::max(P.sp_max_chunks, base[nm]);
        }
        if (split) {
            // (look-ahead schedule: chunks of ~60 products — one round of k_panel_slot's workgroups per column — were measured at a
            //  quarter of config T: 100-196 us per column instead of 45; a chunk is a serial walk, short ones in several rounds win)
            // (look-ahead schedule, chunks of 4 / 6 / 8 / 12 / 16 products at a quarter of config T: 41 / 39 / 44 / 41 / 50 ms per four
            //  factorisations — the chunks are bound by their operand traffic, 64 KB per product, not by their number)
            const int la_chunk = 6, nd_chunk = 6;
            const int cs = lookahead ? std::max(la_chunk, (nc + panel_chunks - 1) / panel_chunks)
                         : panel_ll ? std::max(panel_min_chunk, (nc + panel_chunks - 1) / panel_chunks)
                         : nd_lv ? std::max(nd_chunk, (nc + 4095) / 4096) : std::max(1, (nc + 511) / 512);
            int np = 0;
            if (la_lv && nd_lv) {
                // early chunks of every target, then the late ones; a target's slots: [early chunks | late chunks], contiguous
                std::vector<int> base(g1 - g0 + 1, 0);
                auto n_chunks = [&](int len) { return (len + cs - 1) / cs; };
                for (int g = g0; g < g1; ++g) {
                    const int len = P.lv_cptr[g + 1] - P.lv_cptr[g], ne = lv_ne[g];
                    base[g - g0 + 1] = base[g - g0] + n_chunks(ne) + n_chunks(len - ne);
                }
                // (measured and not adopted, tools/runs/r06_call9.sh: the late contributions inside the fused factor kernel instead of
                //  chunks of their own — one launch less per level, but 12-17 us more on every factor kernel: config T 954 ms against 922)
                for (int pass = 0; pass < 2; ++pass)
                    for (int g = g0; g < g1; ++g) {
                        const int q_lo = P.lv_cptr[g], q_mid = q_lo + lv_ne[g], q_hi = P.lv_cptr[g + 1];
                        const int a = pass == 0 ? q_lo : q_mid, b = pass == 0 ? q_mid : q_hi;
                        int slot = base[g - g0] + (pass == 0 ? 0 : n_chunks(lv_ne[g]));
                        for (int q = a; q < b; q += cs, ++slot) {
                            P.sp_tgt.push_back(P.lv_tgt[2 * g]); P.sp_tgt.push_back(P.lv_tgt[2 * g + 1]);
                            P.sp_q.push_back(q); P.sp_q.push_back(std::min(q + cs, b));
                            P.sp_slot.push_back(slot);
                            if (pass == 0) P.sp_e_cnt[lv]++;
                        }
                    }
                for (int g = g0; g < g1; ++g) {
                    if (base[g - g0 + 1] == base[g - g0]) continue;       // (late contributions only: nothing to sum)
                    P.sp_rt.push_back(P.lv_tgt[2 * g]); P.sp_rt.push_back(P.lv_tgt[2 * g + 1]);
                    P.sp_rp.push_back(base[g - g0]); P.sp_rp.push_back(base[g - g0 + 1]);
                }
                np = base[g1 - g0];
            } else
            for (int g = g0; g < g1; ++g) {
                const int p0 = np;
                for (int q = P.lv_cptr[g]; q < P.lv_cptr[g + 1]; q += cs, ++np) {
                    P.sp_tgt.push_back(P.lv_tgt[2 * g]); P.sp_tgt.push_back(P.lv_tgt[2 * g + 1]);
                    P.sp_q.push_back(q); P.sp_q.push_back(std::min(q + cs, P.lv_cptr[g + 1]));
                    P.sp_slot.push_back(np);
                }
                P.sp_rt.push_back(P.lv_tgt[2 * g]); P.sp_rt.push_back(P.lv_tgt[2 * g + 1]);
                P.sp_rp.push_back(p0); P.sp_rp.push_back(np);
            }
            P.sp_max_chunks = std::max(P.sp_max_chunks, np);
            // (measured at config T and removed: the chunks of a level sorted by shared operands — column k, j range, row i — and dealt
            //  to the XCDs in runs of 128, so that neighbours in one L2 share tiles: 176.4 ms per ten factorisations against 176.9; chunks
            //  of 4 / 6 / 12 products: 176 / 176 / 187.  The launch is bound by the chunk kernel's own staging pipeline — 61 % of
            //  the matrix-core cycles while a CU is busy, CUs busy 81 % of the launch — not by where its operands come from.)
        }
        if (lookahead) {
            // late partials: column k - 2 (the pivot tile's first, so that every workgroup of the column finds its slot at once)
            const int kk = level_first[lv];
            const int m0 = (int)P.md_tgt.size() / 2;
            std::vector<int> slot_of(T, -1);
            if (kk >= 2 && nz[(size_t)kk * T + kk - 2])
                for (int i = kk; i < T; ++i)
                    if (nz[(size_t)i * T + kk] && nz[(size_t)i * T + kk - 2]) {
                        slot_of[i] = (int)P.md_tgt.size() / 2 - m0;
                        P.md_tgt.push_back(i); P.md_tgt.push_back(kk);
                        P.md_q.push_back((int)P.md_cj.size()); P.md_q.push_back((int)P.md_cj.size() + 1);
                        P.md_cj.push_back(kk - 2);
                    }
            P.md_max = std::max(P.md_max, (int)P.md_tgt.size() / 2 - m0);
            for (const FzEnt& e : fz_ents) { P.fz_late.push_back(slot_of[e.k]); P.fz_late.push_back(e.i == e.k ? -1 : slot_of[e.i]); }
        }
        P.md_off.push_back((int)P.md_tgt.size() / 2);
        for (const FzEnt& e : fz_ents) {        // work list of the fused level kernel
            P.fz_tile.push_back(e.i); P.fz_tile.push_back(e.k);
            if (lookahead || (!split && !macro))
                for (int j = first_j[e.k]; j < e.k; ++j)
                    if (nz[(size_t)e.k * T + j]) P.fz_dj.push_back((e.i == e.k || nz[(size_t)e.i * T + j]) ? j : ~j);
            P.fz_dptr.push_back((int)P.fz_dj.size());
        }
        P.fz_off[lv + 1] = (int)P.fz_tile.size() / 2;
        P.sp_chunk_off[lv + 1] = (int)P.sp_tgt.size() / 2;
        P.sp_rt_off[lv + 1] = (int)P.sp_rt.size() / 2;
        P.mp_off[lv + 1] = (int)P.mp_wg.size();
        for (int kk = 0; kk < T; ++kk) {
            if (level[kk] != lv) continue;
            P.lv_k.push_back(kk);
            // backward: column tiles i > k   (CSR aligned with lv_k)
            for (int i = kk + 1; i < T; ++i) if (nz[(size_t)i * T + kk]) P.lv_bi.push_back(i);
            P.lv_bptr.push_back((int)P.lv_bi.size());
        }
        P.lv_k_off[lv + 1] = (int)P.lv_k.size();
    }
    timer.mark("  level lists");
    if (std::getenv("XRSFM_BA_PLAN_VERBOSE")) {
        long long nc_all = 0, nc_split = 0; int n_split = 0, max_nt = 0;
        for (int lv = 0; lv < n_levels; ++lv) {
            const int g0 = P.lv_tgt_off[lv], g1 = P.lv_tgt_off[lv + 1];
            const long long nc = g1 > g0 ? P.lv_cptr[g1] - P.lv_cptr[g0] : 0;
            nc_all += nc; max_nt = std::max(max_nt, g1 - g0);
            if (P.sp_chunk_off[lv + 1] > P.sp_chunk_off[lv]) { ++n_split; nc_split += nc; }
        }
        fprintf(stderr, "[plan] %d tile columns, %d levels (%d split: %lld of %lld list entries), %d chunks in all, largest level %d targets, partial buffer %d tiles, %lld tile products; "
                        "%d block entries of which %d per pair\n",
                T, n_levels, n_split, nc_split, nc_all, (int)P.sp_tgt.size() / 2, max_nt, P.sp_max_chunks, P.tile_products, P.n_writes, P.n_pair_writes);
        {   // levels whose factor launch exceeds one workgroup per CU (k_lv_factor holds 109 KB of LDS)
            int n_over = 0; long long tiles_over = 0, tiles_all = 0;
            for (int lv = 0; lv < n_levels; ++lv) {
                const int nf = P.fz_off[lv + 1] - P.fz_off[lv];
                tiles_all += nf;
                if (nf > 256) { ++n_over; tiles_over += nf; }
            }
            fprintf(stderr, "[plan] factor launches: %lld tiles in %d levels, of which %lld in the %d levels of more than 256 tiles\n", tiles_all, n_levels, tiles_over, n_over);
        }
        if (n_levels <= 16)
            for (int lv = 0; lv < n_levels; ++lv) {
                const int g0 = P.lv_tgt_off[lv], g1 = P.lv_tgt_off[lv + 1];
                fprintf(stderr, "[plan]   level %d: %d columns, %d factor tiles, %d targets with %d list entries, %d chunks, %d fused list entries\n", lv, level_cols[lv],
                        P.fz_off[lv + 1] - P.fz_off[lv], g1 - g0, g1 > g0 ? P.lv_cptr[g1] - P.lv_cptr[g0] : 0, P.sp_chunk_off[lv + 1] - P.sp_chunk_off[lv],
                        P.fz_dptr[P.fz_off[lv + 1]] - P.fz_dptr[P.fz_off[lv]]);
            }
    }
    P.n_levels = n_levels;
    P.use_levels = (2 * n_levels <= T);
    // tile fill fused into the first level's factor launch (k_lv_factor<true>): every workgroup of a level-0 column composes
    // its tiles (k,k) and (i,k) from the block values itself; the tiles of all other columns are composed by extra workgroups
    // of the same launch (fill_rest)
    P.fz_q.resize(P.fz_tile.size());
    for (size_t e = 0; e < P.fz_tile.size() / 2; ++e) {
        const int i = P.fz_tile[2 * e], k2 = P.fz_tile[2 * e + 1];
        P.fz_q[2 * e] = tile_id[(size_t)k2 * T + k2]; P.fz_q[2 * e + 1] = tile_id[(size_t)i * T + k2];
    }
    for (int q = 0; q < P.n_tiles_nz; ++q) if (level[P.tiles_nz[2 * q + 1]] != 0) P.fill_rest.push_back(q);
    if (std::getenv("XRSFM_BA_PLAN_CHECK") && T <= 256) {      // (T^3 counters: 64 MB at 256 tile columns)
        // Self-check of the fused schedule (tests/test_plan_cpu.py, no GPU): every structurally non-zero tile (i,k) must receive
        // each contribution j < k with L_ij and L_kj non-zero exactly once — from a macro-tile entry, a chunk of a split level or
        // its own list in the fused factor kernel — and the forward substitution of row k each L_kj y_j exactly once.
        std::vector<int> upd((size_t)T * T * T, 0), fwd((size_t)T * T, 0);
        auto U = [&](int i, int k2, int j) -> int& { return upd[((size_t)i * T + k2) * T + j]; };
        for (size_t e = 0; e < P.mp_chunk.size() / 8; ++e) {
            const int* m = &P.mp_chunk[8 * e];
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    const int ia = a ? m[1] : m[0], kb = b ? m[3] : m[2];
                    if (ia < 0 || ia < kb) continue;
                    for (int j = m[4]; j < m[5]; ++j) U(ia, kb, j)++;
                }
            if (m[0] == m[2]) for (int j = m[4]; j < m[5]; ++j) { fwd[(size_t)m[2] * T + j]++; if (m[1] >= 0) fwd[(size_t)m[1] * T + j]++; }
        }
        for (size_t e = 0; e < P.sp_tgt.size() / 2; ++e) {
            const int i = P.sp_tgt[2 * e], k2 = P.sp_tgt[2 * e + 1];
            for (int q = P.sp_q[2 * e]; q < P.sp_q[2 * e + 1]; ++q) { U(i, k2, P.lv_cj[q])++; if (i == k2) fwd[(size_t)k2 * T + P.lv_cj[q]]++; }
        }
        for (size_t e = 0; e < P.md_tgt.size() / 2; ++e) {
            const int i = P.md_tgt[2 * e], k2 = P.md_tgt[2 * e + 1];
            for (int q = P.md_q[2 * e]; q < P.md_q[2 * e + 1]; ++q) { U(i, k2, P.md_cj[q])++; if (i == k2) fwd[(size_t)k2 * T + P.md_cj[q]]++; }
        }
        if (P.lookahead) {        // every fused-kernel entry reads the late partials that were written for its two tiles, and only those
            for (int lv = 0; lv < n_levels; ++lv)
                for (int e = P.fz_off[lv]; e < P.fz_off[lv + 1]; ++e)
                    for (int w = 0; w < 2; ++w) {
                        const int sl = P.fz_late[2 * (size_t)e + w];
                        const int ti = w ? P.fz_tile[2 * (size_t)e] : P.fz_tile[2 * (size_t)e + 1], tk = P.fz_tile[2 * (size_t)e + 1];
                        bool want = tk >= 2 && nz[(size_t)tk * T + tk - 2] && nz[(size_t)ti * T + tk - 2] && !(w == 1 && ti == tk);
                        if ((sl >= 0) != want) return kErrPlanCheck;
                        if (sl >= 0 && (sl >= P.md_off[lv + 1] - P.md_off[lv] || P.md_tgt[2 * (size_t)(P.md_off[lv] + sl)] != ti ||
                                        P.md_tgt[2 * (size_t)(P.md_off[lv] + sl) + 1] != tk)) return kErrPlanCheck;
                    }
        }
        for (size_t e = 0; e < P.fz_tile.size() / 2; ++e) {
            const int i = P.fz_tile[2 * e], k2 = P.fz_tile[2 * e + 1];
            for (int q = P.fz_dptr[e]; q < P.fz_dptr[e + 1]; ++q) {
                const int dj = P.fz_dj[q], j = dj >= 0 ? dj : ~dj;
                if (i == k2) { U(k2, k2, j)++; fwd[(size_t)k2 * T + j]++; }
                else if (dj >= 0) U(i, k2, j)++;
            }
        }
        // ... and, level by level, the fixed-order sum of a target reads exactly the partial slots that were written for it
        for (int lv = 0; lv < n_levels; ++lv) {
            std::vector<long long> owner;                    // partial slot (level-relative) -> target key i * T + k, -1 unwritten
            auto put = [&](int slot, int i, int k2) {
                if (slot < 0) return false;
                if ((size_t)slot >= owner.size()) owner.resize(slot + 1, -1);
                if (owner[slot] != -1) return false;         // two writers of one slot
                owner[slot] = (long long)i * T + k2;
                return true;
            };
            if (P.mp_off[lv + 1] - P.mp_off[lv] > 1) {
                for (int w = P.mp_off[lv]; w + 1 < P.mp_off[lv + 1]; ++w)
                    for (int e = P.mp_wg[w]; e < P.mp_wg[w + 1]; ++e) {
                        const int* m = &P.mp_chunk[8 * (size_t)e];
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b) {
                                const int ia = a ? m[1] : m[0], kb = b ? m[3] : m[2];
                                if (ia < 0 || ia < kb) continue;
                                if (!put(m[6] + (2 * a + b) * m[7], ia, kb)) return kErrPlanCheck;
                            }
                    }
            } else {
                for (int c2 = P.sp_chunk_off[lv]; c2 < P.sp_chunk_off[lv + 1]; ++c2) {
                    if (!put(P.sp_slot[c2], P.sp_tgt[2 * (size_t)c2], P.sp_tgt[2 * (size_t)c2 + 1])) return kErrPlanCheck;
                    // look-ahead: an early chunk only names columns that are final when it may run, and the early chunks come first
                    const bool early = c2 - P.sp_chunk_off[lv] < P.sp_e_cnt[lv];
                    for (int q = P.sp_q[2 * (size_t)c2]; q < P.sp_q[2 * (size_t)c2 + 1]; ++q)
                        if (P.la_depth > 0 && (level[P.lv_cj[q]] <= lv - 1 - P.la_depth) != early) return kErrPlanCheck;
                    if (P.la_depth == 0 && early) return kErrPlanCheck;
                }
            }
            size_t read = 0;
            for (int r = P.sp_rt_off[lv]; r < P.sp_rt_off[lv + 1]; ++r) {
                const long long key = (long long)P.sp_rt[2 * (size_t)r] * T + P.sp_rt[2 * (size_t)r + 1];
                for (int q = P.sp_rp[2 * (size_t)r]; q < P.sp_rp[2 * (size_t)r + 1]; ++q, ++read)
                    if (q < 0 || (size_t)q >= owner.size() || owner[q] != key) return kErrPlanCheck;
            }
            size_t written = 0;
            for (long long o2 : owner) written += (o2 != -1);
            if (read != written || (long long)owner.size() > (long long)P.sp_max_chunks) return kErrPlanCheck;
        }
        for (int k2 = 0; k2 < T; ++k2)
            for (int j = 0; j < T; ++j) {
                if (fwd[(size_t)k2 * T + j] != ((j < k2 && nz[(size_t)k2 * T + j]) ? 1 : 0)) return kErrPlanCheck;
                for (int i = k2; i < T; ++i) {
                    const int want = (j < k2 && nz[(size_t)i * T + k2] && nz[(size_t)i * T + j] && nz[(size_t)k2 * T + j]) ? 1 : 0;
                    if (U(i, k2, j) != want) return kErrPlanCheck;
                }
            }
    }
    if (pre) {
        const size_t base = (size_t)64 * 15 * sizeof(double);
        P.n_pairs_small = 0; P.n_pairs_big = 0; P.pairs_shm = base; P.pairs_shm_big = base;
        for (int b = 0; b < 8; ++b) {
            P.gram_n[b] = pre->gram_n[b]; P.gram_shm[b] = pre->gram_shm[b];
            ((b & 1) ? P.n_pairs_big : P.n_pairs_small) += P.gram_n[b];
            ((b & 1) ? P.pairs_shm_big : P.pairs_shm) = std::max((b & 1) ? P.pairs_shm_big : P.pairs_shm, P.gram_shm[b]);
        }
        P.n_pairs_other = pre->n_other;
    } else {
        // S-assembly launches: one per (operand height NI = ceil(6 C / 16), LDS class) of the Gram tiles — the kernel is
        // instantiated per NI so that the 10 accumulators of a 10-camera tile do not shape (and spill) the register allocation
        // of the 4-camera tiles everything else consists of — and one for the other items
        const size_t base = (size_t)64 * 15 * sizeof(double);      // reduction buffer of the diagonal terms (kRedLd)
        const size_t small_cap = kGramSmallLds;
        const int n_items = (int)k.items.size() / 2;
        std::vector<int> bucket[8], other;
        for (int b = 0; b < 8; ++b) P.gram_shm[b] = base;
        for (int it = 0; it < n_items; ++it) {
            const int t = k.items[2 * it];
            if (k.items[2 * it + 1] != 1 || k.tile_ncam[t] <= 0) { other.push_back(it); continue; }   // per-pair path, long tracks
            const int C = k.tile_ncam[t];
            int ntrk = 0;
            for (int q = 0; q < 64 && k.slot_cam[64 * t + q] >= 0; ++q) ntrk += (q == 0 || k.slot_pt[64 * t + q] != k.slot_pt[64 * t + q - 1]);
            int passes = 1;
            const size_t need = std::max(base, (size_t)gram_lds_need(C, ntrk, &passes, P.cam_width));
            // (Gram classes list the TILE itself: one dependent load less at the head of every workgroup)
            const int b = 2 * ((P.cam_width * C + 15) / 16 - 1) + (need <= small_cap ? 0 : 1);
            bucket[b].push_back(t);
            P.gram_shm[b] = std::max(P.gram_shm[b], need);
        }
        P.pairs_items.clear();
        for (int b = 0; b < 8; ++b) {
            P.gram_n[b] = (int)bucket[b].size();
            P.pairs_items.insert(P.pairs_items.end(), bucket[b].begin(), bucket[b].end());
        }
        P.n_pairs_small = 0; P.n_pairs_big = 0;
        for (int b = 0; b < 8; ++b) ((b & 1) ? P.n_pairs_big : P.n_pairs_small) += P.gram_n[b];
        P.pairs_shm = base; P.pairs_shm_big = base;
        for (int b = 0; b < 8; ++b) ((b & 1) ? P.pairs_shm_big : P.pairs_shm) = std::max((b & 1) ? P.pairs_shm_big : P.pairs_shm, P.gram_shm[b]);
        P.n_pairs_other = (int)other.size();
        P.pairs_items.insert(P.pairs_items.end(), other.begin(), other.end());
    }
    return 0;
}

}  // namespace xba
