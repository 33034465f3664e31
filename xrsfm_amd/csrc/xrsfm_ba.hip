// libxrsfm_ba.so — MI355X-native bundle adjustment behind include/xrsfm_ba.h.
//
// Host controller: the Levenberg-Marquardt trust-region loop that the reference
// delegates to ceres::Solve (/root/reference/src/optimization/ba_solver.cc:591,
// 636,672) with SPARSE_SCHUR + LEVENBERG_MARQUARDT (ba_solver.cc:74-75); the
// accept/reject/termination rules restate Ceres' TrustRegionMinimizer
// (SURVEY.md Appendix A.5/A.6).  All arithmetic on the state runs in the HIP
// kernels of ba_kernels.h / ba_chol.h; the host only sees a handful of scalars
// per step.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/xrsfm_ba.h"
#include "ba_chol.h"
#include "ba_filter.h"
#include "ba_kernels.h"
#include "ba_pack.h"
#include "ba_plan.h"
#include "ba_refine.h"
#include "ba_wide.h"
#include "ba_pack_dev.h"
#include "pose_graph.h"
#include "tag_refine.h"

using namespace xba;
static_assert(kCamsPerTileDev == kCamsPerTile, "k_lv_factor / k_lv_bwd index tile_cam with the device constant, the plan fills it with the host constant");
static_assert(kNB == kPlanTile, "tile size of the kernels (ba_chol.h) and of the plan (ba_plan.h)");

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "[xrsfm_ba] HIP error %s at %s:%d: %s\n", hipGetErrorName(e_),   \
                    __FILE__, __LINE__, #expr);                                                \
            return XRSFM_BA_ENODEV;                                                            \
        }                                                                                      \
    } while (0)

// ---------------------------------------------------------------- RCCL (loaded lazily, only for n_ranks > 1)
namespace {
struct UniqueId { char internal[128]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;   // ncclUniqueId is passed by value (128 B struct)
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;          // optional: tears a communicator down without waiting for its pending collectives
};
Rccl g_rccl;
bool load_rccl() {
    if (g_rccl.lib) return true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return false;
    g_rccl.GetUniqueId = (int (*)(UniqueId*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, UniqueId, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.CommAbort = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommAbort");
    return g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
}
constexpr int kNcclFloat64 = 8;   // ncclDouble
constexpr int kNcclSum = 0, kNcclMax = 2;

// kernel classes for the HIP-event profile
enum Kid {
    K_LINEARIZE = 0, K_COST, K_CAM_SEGSUM, K_SCHUR_PREP, K_SCHUR_MATVEC, K_PCG_VEC, K_BACKSUB, K_SCHUR_PAIRS, K_BLOCK_SEGSUM,
    K_DENSE_FILL, K_POTRF, K_TRSM, K_UPDATE, K_TRISOLVE, K_SMALL, K_COUNT
};
const char* kKidName[K_COUNT] = {
    "k_linearize", "k_cost", "k_cam_segsum", "k_schur_prep", "k_schur_matvec", "k_pcg_update", "k_backsub", "k_schur_pairs",
    "k_block_segsum", "k_dense_fill", "k_potrf", "k_trsm", "k_update", "k_fwd_bwd", "small_kernels"};

constexpr int kMaxRanks = 64;          // slots for the per-rank point-gradient maxima behind camlin
constexpr int kCholMaxN = 12288;          // dense-pattern reduced systems (right-looking schedule) up to this many unknowns
constexpr size_t kCholMaxBytes = (size_t)96 << 30;   // tile storage of S for band-ordered problems (level schedule): sized for 288 GB
}  // namespace

// ---------------------------------------------------------------- context
struct CholHost {
    bool ready = false;
    CholDev dev{};
    int n_blocks = 0, n_pairs = 0, n_tiles_nz = 0, T = 0;
    int *slot_pair_ptr = nullptr, *pair_dst = nullptr, *blk_ptr = nullptr, *blk_rc = nullptr;
    double *scat2 = nullptr, *Sblk = nullptr;
    bool bwd_push = false;                        // level schedule with a deep tree: backward substitution in push form (k_bwd2)
    bool bwd_chunk = false; int4* bc_chunks = nullptr; std::vector<int> bc_off; double* bc_part = nullptr; unsigned* bc_ctr = nullptr;   // ... or per level, columns in chunks (k_lv_bwd_chunk)
    int gram4 = 1;                  // Gram tiles staged in one round: 4x4 result blocks (ba_chol.h: gram_tile4); XRSFM_BA_GRAM4=0: 16x16 tiles
    bool pair_from_v = false; int2* ent_src = nullptr; double* pair_v = nullptr;      // long tracks: blocks formed from stored operands (k_chol_segsum_v)
    int* tiles_nz = nullptr;                      // device: (ti,tj) of every structurally non-zero tile
    size_t pairs_shm = 0, pairs_shm_big = 0;      // dynamic LDS of k_schur_pairs per class (ba_plan.h)
    int* pairs_items = nullptr; int n_pairs_small = 0, n_pairs_big = 0, n_pairs_other = 0;
    int gram_n[8] = {0}; size_t gram_shm[8] = {0};   // Gram tiles / dynamic LDS per launch bucket (ba_plan.h)
    bool gram_merge = true;         // the buckets of operand heights 1..3 as ONE launch (XRSFM_BA_GRAM_MERGE=0: one launch per bucket)
    hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;     // the non-Gram items run concurrently
    // right-looking schedule (dense patterns): one panel after the other
    int* cols_flat = nullptr;                     // device list: per tile column its row tiles j < k (push-form backward substitution)
    std::vector<int> cols_off;                    // host offsets per column (size T+1)
    int* bw2_ent = nullptr; std::vector<int> bw2_off, bw2_link;      // two columns per backward launch (ba_plan.h, k_bwd2)
    // level schedule (elimination-tree levels of the tile pattern; left-looking updates)
    bool use_levels = false, panel_ll = false;
    bool lookahead = false;                       // look-ahead panel schedule (ba_plan.h): one launch per column (k_panel_slot)
    int n_levels = 0;
    int *lv_k = nullptr, *lv_tgt = nullptr, *lv_cptr = nullptr, *lv_cj = nullptr;
    int *lv_bptr = nullptr, *lv_bi = nullptr;
    std::vector<int> lv_k_off, lv_tgt_off;        // per level offsets (size n_levels+1)
    // one-launch backward substitution of a level schedule (k_lv_bwd_all): columns by level (root side first), which columns the
    // last factor launch has solved already, the granule buffer, the ticket counter / error word, launches so far
    int* bw_order = nullptr; unsigned char* bw_final = nullptr; unsigned long long* bw_gx = nullptr; unsigned* bw_ctr = nullptr;
    bool bw_debug_timeout = false;  // XRSFM_BA_DEBUG_BWD_TIMEOUT=1 (tests): every hand-off of k_lv_bwd_all waits for a tag that never comes
    int bw_n = 0; unsigned bw_launches = 0; bool bwd_all = false;
    int *sp_tgt = nullptr, *sp_q = nullptr, *sp_rt = nullptr, *sp_rp = nullptr;   // split levels (ba_plan.h)
    int* sp_slot = nullptr;                         // per chunk: its partial slot
    // level look-ahead (ba_plan.h: la_depth): early chunks of level l on the second stream while the main stream is at the levels before it
    int la_depth = 0; std::vector<int> sp_e_cnt; std::vector<hipEvent_t> la_ev_factor, la_ev_early;
    std::vector<int> sp_chunk_off, sp_rt_off, mp_off;
    int *mp_chunk = nullptr, *mp_wg = nullptr;    // macro-tile entries of the panel schedule and their split over workgroups (ba_plan.h)
    double* sp_work = nullptr; int sp_max_chunks = 0;
    int *fz_tile = nullptr, *fz_dptr = nullptr, *fz_dj = nullptr, *tile_cam = nullptr;   // fused level kernel (ba_plan.h)
    int *md_tgt = nullptr, *md_q = nullptr, *md_cj = nullptr, *fz_late = nullptr;        // look-ahead schedule: late partials (ba_plan.h)
    std::vector<int> md_off; int md_max = 0; double* md_work = nullptr;
    int *fz_q = nullptr, *fill_rest = nullptr; int n_fill_rest = 0;    // tile fill inside the first level's launch
    bool S_filled = false;                                   // chol_assemble ran k_tile_fill (else the first level composes its tiles)
    std::vector<int> fz_off;
    int *tf_ptr = nullptr, *tf_ent = nullptr;                // per non-zero tile: its 6x6 blocks (k_tile_fill)
    std::vector<int> cam_off_host;
    std::vector<int> tile_map_host; size_t S_doubles = 0;    // packed tile storage of S (CholDev::tmap), its size in doubles
    int ordering = 0;                                        // 0 natural, 1 nested dissection of a band/ring, 2 reverse Cuthill-McKee, 3 nested dissection of an unordered graph
};

struct xrsfm_ba_context {
    int device = 0;
    hipStream_t stream = nullptr;
    Packed pk;
    Dev d{};
    std::vector<void*> allocs;
    std::vector<size_t> alloc_class;
    CamRec* cam0 = nullptr; double* P0 = nullptr;   // pristine copies for reset
    int n_points_caller = 0;
    double* h_scal = nullptr;       // pinned, coherent, device-visible: S_COUNT scalars + a sequence word
    double* h_scal_dev = nullptr;   // its device address
    unsigned long long seq = 0;
    PcgStatus* h_st = nullptr;      // pinned
    void* comm = nullptr; int n_ranks = 1, rank = 0;
    xrsfm_ba_allreduce_fn hook = nullptr; void* hook_user = nullptr; std::vector<double> hook_buf;   // test transport (host copy)
    bool multi() const { return comm != nullptr || hook != nullptr; }
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    struct Rec { int kid; size_t e0; int tag; };
    std::vector<Rec> recs;
    double prof_ms[K_COUNT] = {0}; int prof_n[K_COUNT] = {0};
    // device-side packing (ba_pack_dev.h): the slot / tile arrays exist on the device only; the host copies the Cholesky plan and
    // the debug entry points read (Packed::slot_cam, ...) are downloaded on first use (ensure_host_pack)
    bool dev_packed = false; int host_pack_level = 2;       // 0 nothing downloaded, 1 what the plan reads, 2 everything (host packing: always 2)
    int* dpk_slot_obs = nullptr; unsigned char* dpk_gt_cell = nullptr;
    bool linearized = false;
    unsigned long long debug_stall_ticks = 0;      // XRSFM_BA_DEBUG_STALL_S (test hook): 100 MHz ticks the next scalar hand-over is held back by
    bool poisoned = false;          // the watchdog tripped: the stream may never drain — destroy must not wait for it (fetch_scalars)
    bool fused = true;              // one-launch linearisation tail (k_lin_tail) and candidate cameras in trailing workgroups of k_backsub;
                                    // XRSFM_BA_FUSED=0 (A/B aid, and always with several ranks for the tail): k_cam_segsum -> k_reduce_multi ->
                                    // k_gradmax_cams -> k_publish and k_cam_update as launches of their own.  (The factorisation always runs
                                    // the fused level / panel kernels: the launch-per-phase Cholesky kernels were removed in round 3.)
    bool wide = false;              // bal9 mode: 9-wide camera blocks (ba_wide.h); single rank, exact solver only
    DevW w{};
    std::vector<int> cam_intr_host; // wide: intrinsics entry of every camera (xrsfm_ba_download_intrinsics)
    bool pcg_coarse = true; double* pcg_w = nullptr;      // PCG path: gauge coarse space of the preconditioner (ba_kernels.h: k_pcg_gauge) and its buffers
    bool prep_fused = true;         // Cholesky path: damped point blocks factored inside k_schur_pairs / k_backsub, LM diagonal of the
                                    // cameras inside the tile fill: no k_point_prep launch (XRSFM_BA_PREP_FUSED=0: round-2 schedule)
    double step_radius = 0.0;       // radius of the step being assembled / solved (prepare_step)
    bool step_prep = false;         // ... and whether its kernels form the point factors themselves
    bool step_valid = false;        // a step of the current linearisation has been assembled and solved (xrsfm_ba_debug_backsub needs it)
    bool gradmax_done = false, published = false;    // the linearisation tail did these in its own launch
    double* part2 = nullptr; unsigned* ticket = nullptr;      // k_lin_tail
    // Second set of linearisation buffers: every LM step linearises at the CANDIDATE point right after the back-substitution
    // (its cost is the candidate cost the step test needs, so no separate cost pass exists); an accepted step swaps the sets.
    struct LinBuf { double* rt = nullptr; double* Jp = nullptr; CamLin* camrec = nullptr; double* Hpp = nullptr; double* gp = nullptr; double* camlin = nullptr; } alt;
    std::vector<unsigned long long> pattern_keys;   // union of the ranks' off-diagonal camera pairs ((row << 32) | col), sorted
    bool have_pattern = false;
    CholHost chol;
};

namespace {

// Process-wide cache of device allocations: BASolver::LBA runs once per registered frame (thousands of small problems per
// reconstruction, incremental_mapper.cc:71) and hipMalloc/hipFree of ~50 buffers would dominate such a call.  Blocks are
// rounded up to size classes, returned to the cache by xrsfm_ba_destroy and reused by the next xrsfm_ba_create on the same device.
struct DevCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;     // (device, class bytes) -> blocks
    size_t cached_bytes = 0;
    static size_t size_class(size_t bytes) {
        size_t c = 256;
        while (c < bytes) c += (c < (1u << 20)) ? c : c / 4;     // x2 up to 1 MiB, then +25 % steps
        return c;
    }
    void* get(int dev, size_t bytes, size_t* cls) {
        *cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_blocks.find({dev, *cls});
            if (it != free_blocks.end() && !it->second.empty()) { void* p = it->second.back(); it->second.pop_back(); cached_bytes -= *cls; return p; }
        }
        void* p = nullptr;
        if (hipMalloc(&p, *cls) != hipSuccess) {          // out of memory: drop the cache and retry once
            release_all();
            if (hipMalloc(&p, *cls) != hipSuccess) return nullptr;
        }
        return p;
    }
    void put(int dev, void* p, size_t cls) {
        std::lock_guard<std::mutex> g(mu);
        if (cached_bytes + cls > (size_t)16 << 30) { (void)hipFree(p); return; }       // keep at most 16 GiB around
        free_blocks[{dev, cls}].push_back(p); cached_bytes += cls;
    }
    void release_all() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& kv : free_blocks) for (void* p : kv.second) (void)hipFree(p);
        free_blocks.clear(); cached_bytes = 0;
    }
};
DevCache g_cache;

// stream + pinned scalar buffers are recycled too (hipStreamCreate/Destroy and hipHostMalloc/Free cost ~0.5 ms per call)
// (and the second stream + fork / join events of the S assembly, created the first time a context needs them: creating and
// destroying them per context cost 2.9 ms of a 4.1 ms LBA call in the mapper replay)
struct HostBundle { hipStream_t stream; double* h_scal; PcgStatus* h_st; hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; };
static void bundle_release(const HostBundle& b) {
    (void)hipHostFree(b.h_scal); (void)hipHostFree(b.h_st); (void)hipStreamDestroy(b.stream);
    if (b.aux) { (void)hipStreamDestroy(b.aux); (void)hipEventDestroy(b.ev_fork); (void)hipEventDestroy(b.ev_join); }
}
struct BundleCache {
    std::mutex mu;
    std::map<int, std::vector<HostBundle>> free_bundles;
    bool get(int dev, HostBundle* b) {
        {
            std::lock_guard<std::mutex> g(mu);
            auto& v = free_bundles[dev];
            if (!v.empty()) { *b = v.back(); v.pop_back(); return true; }
        }
        *b = HostBundle{nullptr, nullptr, nullptr};
        if (hipStreamCreate(&b->stream) != hipSuccess) return false;
        // coherent + mapped: the device publishes the LM scalars straight into this buffer (k_publish) and the host polls
        if (hipHostMalloc((void**)&b->h_scal, sizeof(double) * (S_COUNT + 2), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&b->h_st, sizeof(PcgStatus)) != hipSuccess) {
            if (b->h_scal) (void)hipHostFree(b->h_scal);
            (void)hipStreamDestroy(b->stream);
            return false;
        }
        return true;
    }
    void put(int dev, const HostBundle& b) {
        std::lock_guard<std::mutex> g(mu);
        auto& v = free_bundles[dev];
        if (v.size() >= 8) { bundle_release(b); return; }
        v.push_back(b);
    }
};
BundleCache g_bundles;

// Pinned staging memory of the one-shot helpers (track filter): grow-only, one block per use, recycled.  A multi-megabyte
// hipMemcpyAsync from PAGEABLE memory makes the runtime pin the caller's pages for the copy and release them afterwards — measured
// in the mapper replay as a 10-25 ms stall of the NEXT GPU call of the process (the pose refinement after a whole-map filter).
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_blocks;
    void* get(size_t bytes, size_t* cap) {
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_blocks.size();          // best fit: a small request must not take the block a large one will need next
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].second >= bytes && (best == free_blocks.size() || free_blocks[i].second < free_blocks[best].second)) best = i;
            if (best < free_blocks.size()) { void* p = free_blocks[best].first; *cap = free_blocks[best].second; free_blocks.erase(free_blocks.begin() + best); return p; }
        }
        size_t c = 1 << 16;
        while (c < bytes) c *= 2;
        void* p = nullptr;
        if (hipHostMalloc(&p, c, hipHostMallocDefault) != hipSuccess) return nullptr;
        *cap = c;
        return p;
    }
    void put(void* p, size_t cap) {
        std::lock_guard<std::mutex> g(mu);
        if (free_blocks.size() >= 8 || cap > ((size_t)256 << 20)) { (void)hipHostFree(p); return; }
        free_blocks.push_back({p, cap});
    }
    void release_all() {
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : free_blocks) (void)hipHostFree(b.first);
        free_blocks.clear();
    }
};
PinnedPool g_pinned;

// Deferred release of the host side of large contexts (xrsfm_ba_destroy).  One joinable thread, started on first use.
struct Reaper {
    static constexpr size_t kReaperBacklog = 4;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<xrsfm_ba_context*> q;
    std::thread th;
    bool started = false, stop = false;
    bool push(xrsfm_ba_context* c);
    void loop();
    void drain();          // blocks until everything handed over so far is released
    size_t busy = 0;
    ~Reaper() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv.notify_all();
        if (started && th.joinable()) th.join();
    }
};
Reaper g_reaper;

template <typename T>
int dev_alloc(xrsfm_ba_context* c, T** p, size_t n) {
    if (n == 0) n = 1;
    size_t cls = 0;
    void* q = g_cache.get(c->device, n * sizeof(T), &cls);
    if (!q) return XRSFM_BA_ENOMEM;
    c->allocs.push_back(q);
    c->alloc_class.push_back(cls);
    *p = (T*)q;
    return 0;
}
template <typename T>
int dev_upload(xrsfm_ba_context* c, T** p, const std::vector<T>& v) {
    int e = dev_alloc(c, p, v.size());
    if (e) return e;
    if (!v.empty() && hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return XRSFM_BA_ENODEV;
    return 0;
}

// Many small host -> device arrays per call (an LBA-sized solve uploads ~45 of them at ~10 us each): arrays below 1 MB are
// gathered into one staging buffer, one device block and ONE copy; the big ones keep their own copy.
struct BatchUpload {
    struct Req { void** dst; const void* src; size_t bytes, off; };
    std::vector<Req> reqs;
    size_t total = 0;
    xrsfm_ba_context* c;
    int err = 0;
    explicit BatchUpload(xrsfm_ba_context* ctx) : c(ctx) {}
    template <typename T, typename A> void add(T** dst, const std::vector<T, A>& v) { add_raw(reinterpret_cast<void**>(dst), v.data(), v.size() * sizeof(T)); }
    void add_raw(void** dst, const void* src, size_t bytes) {
        if (err) return;
        if (bytes >= ((size_t)1 << 20)) {
            unsigned char* q = nullptr;
            if ((err = dev_alloc(c, &q, bytes))) return;
            if (hipMemcpy(q, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { err = XRSFM_BA_ENODEV; return; }
            *dst = q;
            return;
        }
        reqs.push_back({dst, src, bytes, total});
        total += ((bytes > 0 ? bytes : 1) + 255) & ~(size_t)255;
    }
    int flush() {
        if (err) return err;
        if (reqs.empty()) return 0;
        unsigned char* base = nullptr;
        if ((err = dev_alloc(c, &base, total))) return err;
        // (round 6) staged in pinned memory from the recycled pool: a copy from a pageable vector goes through the runtime's own
        // staging (or has its pages pinned on the fly)
        size_t cap = 0;
        unsigned char* stage = static_cast<unsigned char*>(g_pinned.get(total, &cap));
        if (!stage) return err = XRSFM_BA_ENOMEM;
        for (const Req& r : reqs) if (r.bytes) memcpy(stage + r.off, r.src, r.bytes);
        const bool ok = hipMemcpy(base, stage, total, hipMemcpyHostToDevice) == hipSuccess;
        g_pinned.put(stage, cap);
        if (!ok) return err = XRSFM_BA_ENODEV;
        for (const Req& r : reqs) *r.dst = base + r.off;
        reqs.clear(); total = 0;
        return 0;
    }
};

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

int allreduce(xrsfm_ba_context* c, double* buf, size_t n, int op) {
    if (c->hook) {
        c->hook_buf.resize(n);
        if (hipMemcpyAsync(c->hook_buf.data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return XRSFM_BA_ENODEV;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return XRSFM_BA_ENODEV;
        if (c->hook(c->hook_user, c->hook_buf.data(), (uint64_t)n, op == kNcclSum ? 0 : 1) != 0) return XRSFM_BA_ECOMM;
        if (hipMemcpyAsync(buf, c->hook_buf.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) return XRSFM_BA_ENODEV;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return XRSFM_BA_ENODEV;
        return 0;
    }
    if (!c->comm) return 0;
    const int e = g_rccl.AllReduce(buf, buf, n, kNcclFloat64, op, c->comm, c->stream);
    return e == 0 ? 0 : XRSFM_BA_ECOMM;
}

// Launch wrapper: in profile mode every launch is bracketed by HIP events on the library's stream.
struct Timed {
    xrsfm_ba_context* c; int kid; int tag; size_t e0 = 0; bool on; hipStream_t st;
    Timed(xrsfm_ba_context* c_, int kid_, int tag_ = -1, hipStream_t st_ = nullptr) : c(c_), kid(kid_), tag(tag_), on(c_->profiling), st(st_ ? st_ : c_->stream) {
        if (!on) return;
        while (c->ev_pool.size() < c->ev_used + 2) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } c->ev_pool.push_back(e); }
        e0 = c->ev_used; c->ev_used += 2;
        (void)hipEventRecord(c->ev_pool[e0], st);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(c->ev_pool[e0 + 1], st);
        c->recs.push_back({kid, e0, tag});
    }
};
#define LAUNCH(ctx, kid, kern, grid, block, shmem, ...)                                  \
    do { Timed t_((ctx), (kid)); hipLaunchKernelGGL(kern, grid, block, shmem, (ctx)->stream, __VA_ARGS__); } while (0)


// resolve recorded event pairs (stream must be idle); records tagged with a PCG iteration index are only
// counted if the iteration really ran (launches after `done` are no-ops)
void profile_collect(xrsfm_ba_context* c, int tag_limit = 1 << 30) {
    for (const auto& r : c->recs) {
        if (r.tag >= tag_limit) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[r.e0], c->ev_pool[r.e0 + 1]) == hipSuccess) { c->prof_ms[r.kid] += ms; c->prof_n[r.kid]++; }
    }
    c->recs.clear();
    c->ev_used = 0;
}

// The LM controller needs ~10 scalars on the host twice per iteration.  A D2H copy + hipStreamSynchronize costs 20-30 us of
// idle GPU each time; instead one tiny kernel stores the scalars and then a sequence number (system-scope release) into
// coherent host memory and the host spins on the sequence number.
// developer aid (XRSFM_BA_TRACE_CALLS): the device's 100 MHz clock into (pinned) memory
__global__ void k_stamp(unsigned long long* out) { *out = wall_clock64(); __threadfence_system(); }

// n 8-byte words from (pinned, device-visible) host memory to device memory
__global__ void k_copy_words(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ void k_publish(const double* __restrict__ scal, double* __restrict__ host, unsigned long long seq) {
    if (threadIdx.x < S_COUNT) host[threadIdx.x] = scal[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + S_COUNT), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// TEST HOOK (XRSFM_BA_DEBUG_STALL_S=<seconds>, tests/test_lifetime_gpu.py): holds the context's stream for a BOUNDED time (<= 10 s) in
// front of the first scalar hand-over of a run — a stand-in for a collective no peer joins — so that the watchdog of fetch_scalars() can be seen to trip,
// poison the context and let xrsfm_ba_destroy return; the kernel always ends by itself (constant 100 MHz counter).
__global__ void k_debug_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

// A hand-off of the one-launch backward substitution that timed out (k_lv_bwd_all) leaves a garbage camera step behind: the kernel
// raises scalar slot S_BWD_ERR, which arrives with the very next hand-over of the scalar block — the run stops here, before the LM
// controller evaluates a step built on it (until round 5 the device word was read once per solve, after the stream had drained).
static int bwd_err_check(xrsfm_ba_context* c) {
    if (c->h_scal[S_BWD_ERR] == 0.0) return 0;
    fprintf(stderr, "[xrsfm_ba] backward substitution: a hand-off between workgroups timed out (k_lv_bwd_all); rerun with XRSFM_BA_BWD_ALL=0\n");
    c->h_scal[S_BWD_ERR] = 0.0;
    (void)hipMemsetAsync(c->d.scal + S_BWD_ERR, 0, sizeof(double), c->stream);
    if (c->chol.bw_ctr) (void)hipMemsetAsync(c->chol.bw_ctr + 1, 0, sizeof(unsigned), c->stream);
    return XRSFM_BA_EINTERNAL;
}

int fetch_scalars(xrsfm_ba_context* c) {
    bool tail_published = c->published;
    c->published = false;               // (cleared on every exit path, errors included)
    HIPCHK(hipGetLastError());          // a failed launch since the last sync point
    if (c->debug_stall_ticks) {         // test hook: the stream stalls HERE, where the host polls (not in front of a blocking copy of the set-up)
        hipLaunchKernelGGL(k_debug_stall, dim3(1), dim3(1), 0, c->stream, c->debug_stall_ticks);
        c->debug_stall_ticks = 0;
        tail_published = false;         // (a fresh hand-over behind the stall)
    }
    if (c->profiling || !c->h_scal_dev) {
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d.scal, sizeof(double) * S_COUNT, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->profiling) profile_collect(c);
        return bwd_err_check(c);
    }
    unsigned long long want;
    if (tail_published) want = c->seq;                                  // k_lin_tail hands the block over itself
    else {
        want = ++c->seq;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, c->stream, c->d.scal, c->h_scal_dev, want);
        HIPCHK(hipGetLastError());
    }
    const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(c->h_scal + S_COUNT);
    // Watchdog: a collective that never completes (a rank that died, ranks that disagree on the call sequence) would otherwise
    // leave every other rank spinning here for ever.  XRSFM_BA_WATCHDOG_S (default 300 s without progress; 0 = off).
    // Default: multi-rank contexts only (300 s) — on one rank nothing can deadlock, and a legitimately long wait (hipStreamQuery
    // still returns NotReady) must not become a spurious ENODEV; an explicit XRSFM_BA_WATCHDOG_S applies to every context.
    static const double watchdog_env = [] { const char* e = std::getenv("XRSFM_BA_WATCHDOG_S"); return e ? std::atof(e) : -1.0; }();
    const double watchdog_s = watchdog_env >= 0.0 ? watchdog_env : (c->multi() ? 300.0 : 0.0);
    std::chrono::steady_clock::time_point t_wait{};
    bool waiting = false;
    for (unsigned spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want; ++spins) {
        if ((spins & 0xfffff) == 0xfffff) {                 // every ~1M polls: has the stream died?
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) break; return XRSFM_BA_ENODEV; }
            if (q != hipErrorNotReady) return XRSFM_BA_ENODEV;
            const auto now = std::chrono::steady_clock::now();
            if (!waiting) { waiting = true; t_wait = now; }
            else if (watchdog_s > 0.0 && std::chrono::duration<double>(now - t_wait).count() > watchdog_s) {
                fprintf(stderr, "[xrsfm_ba] rank %d of %d: the device has made no progress for %.0f s%s\n", c->rank, c->n_ranks, watchdog_s,
                        c->multi() ? " — an all-reduce that never completed? every rank must make the same calls in the same order (XRSFM_BA_WATCHDOG_S)" : "");
                c->poisoned = true;        // the queued kernels / collective may never finish: xrsfm_ba_destroy will not wait for them
                return c->multi() ? XRSFM_BA_ECOMM : XRSFM_BA_ENODEV;
            }
        }
        __builtin_ia32_pause();
    }
    return bwd_err_check(c);
}

// Linearise at the state `d` views (the context's own Dev, or the candidate view of finish_step).  Leaves S_COST, S_XNORM2_PTS in
// the scalar block and camlin / Hpp / gp / rt / Jp of that view filled.  with_step: the partial sums the preceding
// back-substitution left (model decrease, squared step norms, |x_cams|^2) are reduced by the same launch and, with several
// ranks, travel in the same all-reduce: layout behind the camera block = [cost, |x_pts|^2, model, |step_pts|^2, rank slots].
enum { LIN_SKIP_CAMLIN = 1, LIN_FINAL = 2 };    // LIN_FINAL: gradient max-norm and the hand-over to the host follow this linearisation
int linearize(xrsfm_ba_context* c, double huber_a, const Dev& d, bool with_step, int flags = 0) {
    const Dev& own = c->d;
    // "the tail of the previous linearisation has already done this" never outlives that linearisation (an error return
    // between a linearisation and its fetch_scalars(), or a debug entry point that skips the fetch, must not make a later
    // caller skip k_gradmax_cams / k_publish and read stale scalars)
    c->gradmax_done = false; c->published = false;
    c->step_valid = false;
    if (d.n_cams > 0 && !(flags & LIN_SKIP_CAMLIN)) LAUNCH(c, K_SMALL, k_cam_lin, dim3(cdiv(d.n_cams, kBlock)), dim3(kBlock), 0, d);
    if (d.n_items > 0) LAUNCH(c, K_LINEARIZE, k_linearize, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), kWavesPerBlock * kWave * 13 * sizeof(double), d, huber_a);
    if (!c->fused && d.n_cams > 0) LAUNCH(c, K_CAM_SEGSUM, k_cam_segsum<12>, dim3(d.n_cams), dim3(kBlock), 0, d.scat, d.cam_ptr_g, d.camlin, (const PcgStatus*)nullptr);
    {
        double* tail = d.camlin + (size_t)d.n_cams * 12;
        const bool multi = c->multi();
        ReduceJobs j{};
        int nj = 0;
        auto job = [&](const double* in, int n, double* out, int op) { j.in[nj] = in; j.n[nj] = n; j.out[nj] = out; j.op[nj] = op; ++nj; };
        job(d.part, d.n_items, multi ? tail : d.scal + S_COST, 0);
        job(d.part + d.n_items, d.n_items, multi ? tail + 1 : d.scal + S_XNORM2_PTS, 0);
        // max-norm of the point gradient: points are rank-local, so with several ranks every rank writes its maximum to its own
        // slot behind the sums (the other slots are 0): the SUM all-reduce then hands every rank all the maxima
        job(d.part + 2 * (size_t)d.n_items, d.n_items, multi ? tail + 4 + c->rank : d.scal + S_GRADMAX_PTS, 1);
        if (with_step) {
            job(own.part + 2 * (size_t)own.n_items, own.n_items, multi ? tail + 2 : d.scal + S_MODEL, 0);
            job(own.part + 3 * (size_t)own.n_items, own.n_items, multi ? tail + 3 : d.scal + S_STEP2_PTS, 0);
            job(own.campart, own.n_cams, d.scal + S_STEP2_CAMS, 0);
            job(own.campart + own.n_cams, own.n_cams, d.scal + S_XNORM2_CAMS, 0);
        }
        if (multi) HIPCHK(hipMemsetAsync(tail + 2, 0, sizeof(double) * (2 + (size_t)c->n_ranks), c->stream));
        if (c->fused) {
            // one launch: per-camera sums, the reductions and — on one rank, when this is the last linearisation before the
            // host looks — the camera gradient max-norm and the hand-over of the scalar block
            TailArgs a{};
            a.jobs = j; a.njobs = nj; a.part2 = c->part2; a.ticket = c->ticket;
            const bool final_here = (flags & LIN_FINAL) && !multi;
            a.gradmax = final_here ? 1 : 0;
            if (final_here && !c->profiling && c->h_scal_dev) { a.host = c->h_scal_dev; a.seq = ++c->seq; c->published = true; }
            LAUNCH(c, K_SMALL, k_lin_tail, dim3(std::min(kTailGrid, std::max(d.n_cams, 1))), dim3(kBlock), 0, d, a);
            if (final_here) c->gradmax_done = true;
        } else {
            LAUNCH(c, K_SMALL, k_reduce_multi, dim3(nj), dim3(kPcgThreads), 0, j);
        }
        if (multi) {
            int e = allreduce(c, d.camlin, (size_t)d.n_cams * 12 + 4 + (size_t)c->n_ranks, kNcclSum);
            if (e) return e;
            HIPCHK(hipMemcpyAsync(d.scal + S_COST, tail, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));   // S_COST .. S_STEP2_PTS adjacent
        }
    }
    return 0;
}

// after linearize() of the same view, which leaves the point part in S_GRADMAX_PTS (one rank) or in the per-rank slots behind camlin
int gradient_max_enqueue(xrsfm_ba_context* c, const Dev& d) {
    if (c->gradmax_done) { c->gradmax_done = false; return 0; }        // k_lin_tail of the same view has done it
    const double* rank_max = c->multi() ? d.camlin + (size_t)d.n_cams * 12 + 4 : nullptr;
    LAUNCH(c, K_SMALL, k_gradmax_cams, dim3(1), dim3(kPcgThreads), 0, d, d.scal + S_GRADMAX_CAMS, rank_max, c->n_ranks, d.scal + S_GRADMAX_PTS);
    return 0;
}
int gradient_max(xrsfm_ba_context* c, double* out) {
    int e = gradient_max_enqueue(c, c->d);
    if (e) return e;
    e = fetch_scalars(c);
    if (e) return e;
    *out = std::fmax(c->h_scal[S_GRADMAX_PTS], c->h_scal[S_GRADMAX_CAMS]);
    return 0;
}

// Everything that depends on the radius: D^2, Hpp^-1, diagonal blocks of S and the reduced right-hand side.
int prepare_step(xrsfm_ba_context* c, double radius, bool with_blocks = false) {
    Dev& d = c->d;
    const double dmin = kLmDiagMin, dmax = kLmDiagMax;
    c->step_radius = radius;
    c->step_prep = with_blocks && c->prep_fused;
    if (c->step_prep) return 0;     // Cholesky path: k_schur_pairs / k_backsub / the tile fill form what they need from Hpp, camlin and the radius
    {
        const int nbp = cdiv(d.n_pts, kBlock), nbc = cdiv((long long)d.n_cams * 6, kBlock);
        if (nbp + nbc > 0) LAUNCH(c, K_SMALL, k_point_prep, dim3(nbp + nbc), dim3(kBlock), 0, d, radius, dmin, dmax, nbp);
    }
    if (with_blocks) return 0;      // Cholesky path: k_schur_pairs also produces the diagonal blocks / rhs (chol_assemble)
    if (d.n_slots > 0) LAUNCH(c, K_SCHUR_PREP, k_schur_prep, dim3(cdiv(d.n_slots, kBlock)), dim3(kBlock), 0, d);
    if (d.n_cams > 0) LAUNCH(c, K_CAM_SEGSUM, k_cam_segsum<28>, dim3(d.n_cams), dim3(kBlock), 0, d.scat, d.cam_ptr, d.camS, (const PcgStatus*)nullptr);
    int e = allreduce(c, d.camS, (size_t)d.n_cams * 28, kNcclSum);
    if (e) return e;
    if (d.n_cams > 0) LAUNCH(c, K_SMALL, k_cam_factor, dim3(cdiv(d.n_cams, 64)), dim3(64), 0, d);
    return 0;
}

// y = sum_obs F^T (F p - E Hinv E^T F p)  (+ D_c^2 p is added by the consumer)
int schur_product(xrsfm_ba_context* c, const double* p_dev, double* out_dev, int tag) {
    Dev& d = c->d;
    if (d.n_items > 0) { Timed t_(c, K_SCHUR_MATVEC, tag); hipLaunchKernelGGL(k_schur_matvec, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d, p_dev); }
    if (d.n_cams > 0) { Timed t_(c, K_CAM_SEGSUM, tag); hipLaunchKernelGGL(k_cam_segsum<6>, dim3(d.n_cams), dim3(kBlock), 0, c->stream, d.scat, d.cam_ptr, out_dev, (const PcgStatus*)d.st); }
    return allreduce(c, out_dev, (size_t)d.n_cams * 6, kNcclSum);
}

int pcg_solve(xrsfm_ba_context* c, const xrsfm_ba_options& opt, xrsfm_ba_summary* sum) {
    Dev& d = c->d;
    // two-level preconditioner: the seven gauge vectors at the current cameras, (S + D^2) W by seven products, the 7 x 7 coarse
    // matrix and its inverse — per LM step: both S and D^2 depend on the radius (XRSFM_BA_PCG_COARSE=0: block-Jacobi alone)
    d.pcgW = (c->pcg_coarse && d.n_cams > 0) ? c->pcg_w : nullptr;        // (no cameras: the coarse matrix would never be written)
    if (d.pcgW && d.n_cams > 0) {
        const size_t n6 = 6 * (size_t)d.n_cams;
        d.pcgSW = c->pcg_w + kGauge * n6; d.pcgE = c->pcg_w + 2 * kGauge * n6;
        LAUNCH(c, K_PCG_VEC, k_pcg_gauge, dim3(cdiv(d.n_cams, kBlock)), dim3(kBlock), 0, d);
        HIPCHK(hipMemsetAsync(d.st, 0, sizeof(PcgStatus), c->stream));        // (the product kernels return at once while st->done is set)
        for (int g = 0; g < kGauge; ++g) {
            int e = schur_product(c, d.pcgW + g * n6, d.pcgSW + g * n6, -1);
            if (e) return e;
        }
        LAUNCH(c, K_PCG_VEC, k_pcg_gauge_damp, dim3(cdiv((long long)n6, kBlock)), dim3(kBlock), 0, d);
        LAUNCH(c, K_PCG_VEC, k_pcg_coarse, dim3(1), dim3(kPcgThreads), 0, d);
    }
    LAUNCH(c, K_PCG_VEC, k_pcg_init, dim3(1), dim3(kPcgThreads), 0, d);
    const int chunk = 8;
    int launched = 0;
    while (true) {
        HIPCHK(hipMemcpyAsync(c->h_st, d.st, sizeof(PcgStatus), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->profiling) profile_collect(c, c->h_st->it);
        if (c->h_st->done || launched >= opt.pcg_max_iterations) break;
        for (int i = 0; i < chunk; ++i) {
            // one rank: q = S p + D^2 p and the per-camera shares of p.q are formed where a camera's partials of the product are
            // summed (k_pcg_segsum_q); several ranks: the sums are all-reduced first, k_pcg_q follows
            const bool fused_q = !c->multi() && d.n_cams > 0;
            if (fused_q) {
                if (d.n_items > 0) { Timed t_(c, K_SCHUR_MATVEC, launched); hipLaunchKernelGGL(k_schur_matvec, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d, (const double*)d.pp); }
                { Timed t_(c, K_CAM_SEGSUM, launched); hipLaunchKernelGGL(k_pcg_segsum_q, dim3(d.n_cams), dim3(kBlock), 0, c->stream, d, d.pcgpart); }
            } else {
                int e = schur_product(c, d.pp, d.pq, launched);
                if (e) return e;
            }
            if (d.n_cams > 0) {
                Timed t_(c, K_PCG_VEC, launched);
                const dim3 grid(cdiv(d.n_cams, kPcgBlock));
                if (!fused_q) hipLaunchKernelGGL(k_pcg_q, grid, dim3(kPcgBlock), 0, c->stream, d, d.pcgpart);
                hipLaunchKernelGGL(k_pcg_xr, grid, dim3(kPcgBlock), 0, c->stream, d, d.pcgpart);
                hipLaunchKernelGGL(k_pcg_p, grid, dim3(kPcgBlock), 0, c->stream, d, (const double*)d.pcgpart, opt.pcg_tolerance, opt.pcg_max_iterations);
            }
            ++launched;
        }
    }
    sum->pcg_iterations += c->h_st->it;
    return 0;
}

// Host copies of the packed arrays of a device-packed context (ba_pack_dev.h): level 1 = what the Cholesky plan reads (slot_cam,
// slot_pt, slot_cidx, tile_ncam, tile_gt_off, gt_cell), level 2 = everything but the observations' u, v (debug entry points).
int ensure_host_pack(xrsfm_ba_context* c, int level) {
    if (!c->dev_packed || c->host_pack_level >= level) return 0;
    Packed& k = c->pk;
    const Dev& d = c->d;
    HIPCHK(hipSetDevice(c->device));
    const size_t ns = (size_t)k.n_slots, nt = (size_t)k.n_tiles;
    auto get = [&](auto& vec, const void* src, size_t n) -> int {
        vec.resize(n);
        if (n && hipMemcpy(vec.data(), src, n * sizeof(vec[0]), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
        return 0;
    };
    int e = 0;
    if (c->host_pack_level < 1) {
        if ((e = get(k.slot_cam, d.slot_cam, ns)) || (e = get(k.slot_pt, d.slot_pt, ns)) || (e = get(k.slot_cidx, d.slot_cidx, ns)) ||
            (e = get(k.tile_ncam, d.tile_ncam, nt)) || (e = get(k.tile_gt_off, d.tile_gt_off, nt)) || (e = get(k.gt_cell, c->dpk_gt_cell, (size_t)k.n_gt_cells))) return e;
        c->host_pack_level = 1;
    }
    if (level >= 2) {
        if ((e = get(k.slot_obs, c->dpk_slot_obs, ns)) || (e = get(k.slot_campos, d.slot_campos, ns)) || (e = get(k.slot_campos_g, d.slot_campos_g, ns)) ||
            (e = get(k.tile_stride, d.tile_stride, nt)) || (e = get(k.tile_maxlen, d.tile_maxlen, nt))) return e;
        c->host_pack_level = 2;
    }
    return 0;
}

// dynamic-LDS limits of the S-assembly kernels: once per device and process (each call costs a few microseconds, an LBA-sized solve
// has few to spare; xrsfm_ba_warmup does it ahead of the first call)
void set_kernel_attributes(int device) {
    static std::mutex mu;
    static std::vector<char> done_for(64, 0);
    std::lock_guard<std::mutex> g(mu);
    if (device < 0 || device >= 64 || done_for[device]) return;
    const int pairs_max = kGramMaxLds + kGramTabLd * kGramTabLd * (int)sizeof(int);
#define XBA_PAIRS_ATTR(NI) \
    (void)hipFuncSetAttribute((const void*)k_schur_pairs<true, true, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max); \
    (void)hipFuncSetAttribute((const void*)k_schur_pairs<true, false, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    XBA_PAIRS_ATTR(0) XBA_PAIRS_ATTR(1) XBA_PAIRS_ATTR(2) XBA_PAIRS_ATTR(3) XBA_PAIRS_ATTR(4)
#undef XBA_PAIRS_ATTR
    (void)hipFuncSetAttribute((const void*)k_schur_pairs<false, true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    (void)hipFuncSetAttribute((const void*)k_schur_pairs<false, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    (void)hipFuncSetAttribute((const void*)k9_pairs_gram<1>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    (void)hipFuncSetAttribute((const void*)k9_pairs_gram<2>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    (void)hipFuncSetAttribute((const void*)k9_pairs_gram<3>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    (void)hipFuncSetAttribute((const void*)k9_pairs_gram<4>, hipFuncAttributeMaxDynamicSharedMemorySize, pairs_max);
    done_for[device] = 1;
}

// ---------------------------------------------------------------- Cholesky path: structures
int chol_setup(xrsfm_ba_context* c) {
    CholHost& h = c->chol;
    if (h.ready) return 0;
    const Packed& k = c->pk;
    const int Nc = k.n_cams;
    std::vector<int> spp;
    PairKeys keyed;
    PhaseTimer timer("chol setup");
    int e = 0;
    // Device-packed context on one rank: the pair keys, blocks and destinations are generated on the device as well (ba_pack_dev.h:
    // device_keys) — the host plan only sees the block list; XRSFM_BA_DEVICE_KEYS=0: download the packed arrays, host keys.
    devpack::KeysResult KR;
    PlanPrebuilt pre;
    bool dev_keys = false;
    {
        const char* dk = std::getenv("XRSFM_BA_DEVICE_KEYS");
        dev_keys = c->dev_packed && !c->wide && c->n_ranks == 1 && !c->have_pattern && !(dk && dk[0] == '0');
    }
    if (dev_keys) {
        std::vector<std::pair<void*, size_t>> scratch;
        auto keep = [&](size_t bytes) -> void* { unsigned char* q = nullptr; return dev_alloc(c, &q, bytes) ? nullptr : (void*)q; };
        auto scr = [&](size_t bytes) -> void* { size_t cls = 0; void* q = g_cache.get(c->device, bytes, &cls); if (q) scratch.push_back({q, cls}); return q; };
        e = devpack::device_keys(k, c->d.slot_cam, c->d.slot_pt, c->d.slot_cidx, c->d.tile_ncam, c->d.tile_gt_off, c->dpk_gt_cell, c->stream, keep, scr, KR);
        (void)hipStreamSynchronize(c->stream);
        for (auto& b : scratch) g_cache.put(c->device, b.first, b.second);
        if (e) return e;
        if (KR.duplicate) return kErrDuplicateObs;
        pre.n_pairs = KR.n_pairs; pre.n_writes = KR.n_writes; pre.n_pair_writes = KR.n_pair_writes; pre.blk_rc = KR.blk_rc_host; pre.n_other = KR.n_other;
        for (int b = 0; b < 8; ++b) { pre.gram_n[b] = KR.gram_n[b]; pre.gram_shm[b] = KR.gram_shm[b]; }
    } else {
        if ((e = ensure_host_pack(c, 1))) return e;
        e = chol_local_keys(k, spp, keyed);
        if (e) return e;          // (kErrDuplicateObs: translated by the callers)
    }
    timer.mark("pair keys");
    // multi-GPU: every rank must hold the same blocks in the same order so that the block values can be all-reduced:
    // union of the ranks' camera pairs by an all-reduce(max) of an N_c x N_c occupancy map (once per problem)
    if (c->n_ranks > 1 && !c->have_pattern) {
        std::vector<double> occ((size_t)Nc * Nc, 0.0);
        for (const auto& kv : keyed) occ[(size_t)(kv.first >> 32) * Nc + (kv.first & 0xffffffffu)] = 1.0;
        double* d_occ = nullptr;
        if (hipMalloc((void**)&d_occ, occ.size() * sizeof(double)) != hipSuccess) return XRSFM_BA_ENOMEM;
        int e2 = XRSFM_BA_OK;
        if (hipMemcpy(d_occ, occ.data(), occ.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) e2 = XRSFM_BA_ENODEV;
        if (!e2) e2 = allreduce(c, d_occ, occ.size(), kNcclMax);
        if (!e2 && hipStreamSynchronize(c->stream) != hipSuccess) e2 = XRSFM_BA_ENODEV;
        if (!e2 && hipMemcpy(occ.data(), d_occ, occ.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) e2 = XRSFM_BA_ENODEV;
        (void)hipFree(d_occ);
        if (e2) return e2;
        c->pattern_keys.clear();
        for (int rb = 0; rb < Nc; ++rb)
            for (int ca = 0; ca < rb; ++ca)
                if (occ[(size_t)rb * Nc + ca] > 0.0) c->pattern_keys.push_back(((unsigned long long)rb << 32) | (unsigned)ca);
        c->have_pattern = true;
    }
    CholPlan P;
    // (a host allocation that fails inside the plan — the T x T tile maps of a very large unordered problem — is reported as
    //  ENOMEM from here, so that AUTO still falls back to the PCG instead of the C boundary turning it into a failed run)
    try { e = chol_plan_build(k, spp, keyed, c->have_pattern ? &c->pattern_keys : nullptr, P, kCholMaxN, kCholMaxBytes, c->wide ? kW : 6, dev_keys ? &pre : nullptr); }
    catch (const std::bad_alloc&) { return XRSFM_BA_ENOMEM; }
    if (e) return e == kErrPlanCheck ? XRSFM_BA_EINTERNAL : e;
    timer.mark("plan");
    // dense tile storage: any pattern up to kCholMaxN unknowns; beyond that only with a shallow elimination tree (band / ring
    // ordering found) and while the n_pad^2 doubles stay within kCholMaxBytes
    // (... or a reverse Cuthill-McKee order whose symbolic factorisation stays within the work budget of ba_plan.h: panel schedule)
    if (P.n > kCholMaxN && !(P.use_levels || P.ordering >= 2)) return XRSFM_BA_ETOOBIG;       // (the size of the tile storage is checked where it is allocated)
    h.n_blocks = P.n_blocks; h.n_pairs = P.n_pairs; h.T = P.T; h.n_tiles_nz = P.n_tiles_nz; h.n_levels = P.n_levels;
    h.use_levels = P.use_levels; h.panel_ll = P.panel_ll; h.lookahead = P.lookahead; h.ordering = P.ordering; h.pairs_shm = P.pairs_shm; h.pairs_shm_big = P.pairs_shm_big; h.n_pairs_small = P.n_pairs_small; h.n_pairs_big = P.n_pairs_big; h.n_pairs_other = P.n_pairs_other; h.cam_off_host = P.cam_off;
    for (int b = 0; b < 8; ++b) { h.gram_n[b] = P.gram_n[b]; h.gram_shm[b] = P.gram_shm[b]; }
    { const char* me = std::getenv("XRSFM_BA_GRAM_MERGE"); h.gram_merge = !(me && me[0] == '0'); }      // (read per set-up: the A/B test switches it)
    h.cols_off = P.cols_off; h.bw2_off = P.bw2_off; h.bw2_link = P.bw2_link;
    h.lv_k_off = P.lv_k_off; h.lv_tgt_off = P.lv_tgt_off;
    h.sp_chunk_off = P.sp_chunk_off; h.sp_rt_off = P.sp_rt_off; h.mp_off = P.mp_off; h.fz_off = P.fz_off;
    int *d_cam_off = nullptr, *d_tile_rows = nullptr, *d_tmap = nullptr;
#define TRYC(x) do { e = (x); if (e) return e; } while (0)
    BatchUpload up(c);
    if (dev_keys) { h.slot_pair_ptr = KR.spp; h.pair_dst = KR.pair_dst; h.blk_ptr = KR.blk_ptr; h.blk_rc = KR.blk_rc; }
    else { up.add(&h.slot_pair_ptr, P.spp); up.add(&h.pair_dst, P.pair_dst); up.add(&h.blk_ptr, P.blk_ptr); up.add(&h.blk_rc, P.blk_rc); }
    up.add(&h.tiles_nz, P.tiles_nz); up.add(&h.cols_flat, P.cols_flat); up.add(&h.bw2_ent, P.bw2_ent);
    up.add(&h.lv_k, P.lv_k); up.add(&h.lv_tgt, P.lv_tgt); up.add(&h.lv_cptr, P.lv_cptr);
    up.add(&h.lv_cj, P.lv_cj);
    up.add(&h.lv_bptr, P.lv_bptr); up.add(&h.lv_bi, P.lv_bi);
    up.add(&h.sp_tgt, P.sp_tgt); up.add(&h.sp_q, P.sp_q); up.add(&h.mp_chunk, P.mp_chunk); up.add(&h.mp_wg, P.mp_wg);
    up.add(&h.sp_rt, P.sp_rt); up.add(&h.sp_rp, P.sp_rp); up.add(&h.sp_slot, P.sp_slot);
    h.la_depth = P.la_depth; h.sp_e_cnt = P.sp_e_cnt;
    up.add(&h.tf_ptr, P.tf_ptr); up.add(&h.tf_ent, P.tf_ent);
    up.add(&h.fz_tile, P.fz_tile); up.add(&h.fz_dptr, P.fz_dptr); up.add(&h.fz_dj, P.fz_dj);
    up.add(&h.fz_q, P.fz_q); up.add(&h.fill_rest, P.fill_rest); h.n_fill_rest = (int)P.fill_rest.size();
    up.add(&h.md_tgt, P.md_tgt); up.add(&h.md_q, P.md_q); up.add(&h.md_cj, P.md_cj); up.add(&h.fz_late, P.fz_late);
    h.md_off = P.md_off; h.md_max = P.md_max;
    up.add(&h.tile_cam, P.tile_cam);
    if (dev_keys) h.pairs_items = KR.pairs_items; else up.add(&h.pairs_items, P.pairs_items);
    h.sp_max_chunks = P.sp_max_chunks;
    // (partial tiles: one buffer; two with the look-ahead panel schedule; la_depth + 1 with the level look-ahead — level l's buffer is
    //  l mod (la_depth + 1): written from the moment level l - 1 - la_depth is factored until level l has been summed)
    TRYC(dev_alloc(c, &h.sp_work, (size_t)std::max(1, P.sp_max_chunks) * kPartStride * (P.lookahead ? 2 : (P.la_depth > 0 ? P.la_depth + 1 : 1))));
    if (P.lookahead) TRYC(dev_alloc(c, &h.md_work, (size_t)std::max(1, P.md_max) * kPartStride * 2));
    up.add(&d_cam_off, P.cam_off); up.add(&d_tile_rows, P.tile_rows); up.add(&d_tmap, P.tile_map);
    TRYC(up.flush());
    timer.mark("uploads");
    const size_t blk_vals = c->wide ? kWB : 36, cam_vals = c->wide ? kWS : 28;
    {   // diagonal-block buffer and off-diagonal block values in one allocation: one all-reduce per LM step
        double* both = nullptr;
        TRYC(dev_alloc(c, &both, (size_t)Nc * cam_vals + (size_t)(P.n_blocks > 0 ? P.n_blocks : 1) * blk_vals));
        if (c->wide) c->w.camS = both; else c->d.camS = both;
        h.Sblk = both + (size_t)Nc * cam_vals;
    }
    {   // problems whose block entries are mostly PER-PAIR blocks (tracks that fit no Gram tile: random visibility, the long tracks of a
        // photo collection; >= 256 k of them and at least half of all entries): the blocks are formed from stored operands where they
        // are summed (ba_kernels.h: k_chol_segsum_v) instead of being written per pair and read back; XRSFM_BA_PAIR_V=0 / 1 forces
        // one form.  Measured per LM iteration, pairs + sum: config T 11.8 + 3.3 -> 1.3 + 4.4 ms, D 0.37 + 0.86 -> 0.18 + 0.43,
        // U 0.19 + 0.13 -> 0.10 + 0.09; a ragged sequential map (config R: 0.66 M entries, most of them Gram cells) is faster with
        // the plain segmented sum (45 us against 73).
        const char* g4 = std::getenv("XRSFM_BA_GRAM4");          // (read per set-up: the A/B test switches it)
        h.gram4 = !(g4 && g4[0] == '0');
        const char* pe = std::getenv("XRSFM_BA_PAIR_V");        // (read per set-up: the A/B test switches it)
        h.pair_from_v = !c->wide && h.n_pairs_other > 0 && (pe ? pe[0] != '0' : (P.n_pair_writes >= 262144 && 2LL * P.n_pair_writes >= P.n_writes));
        // the scatter buffer of the block entries: one 36-value record per entry — with stored operands only the Gram tiles' cells are
        // ever written, so it holds those alone, numbered compactly (config T: 0.6 MB instead of 17.8 GB)
        const size_t n_scat2 = h.pair_from_v ? (size_t)std::max(0, P.n_writes - P.n_pair_writes) : (size_t)std::max(0, P.n_writes);
        TRYC(dev_alloc(c, &h.scat2, std::max<size_t>(1, n_scat2) * blk_vals));
        if (h.pair_from_v) {
            const size_t ne = (size_t)std::max(1, P.n_writes);
            TRYC(dev_alloc(c, &h.ent_src, ne));
            TRYC(dev_alloc(c, &h.pair_v, (size_t)std::max(1, k.n_slots) * 18));
            HIPCHK(hipMemsetAsync(h.ent_src, 0xff, sizeof(int2) * ne, c->stream));
            hipLaunchKernelGGL(k_pair_sources, dim3(h.n_pairs_other), dim3(kWave), 0, c->stream, c->d, (const int*)(h.pairs_items + h.n_pairs_small + h.n_pairs_big),
                               (const int*)h.slot_pair_ptr, (const int*)h.pair_dst, h.ent_src);
            HIPCHK(hipGetLastError());
            const int n_cells = c->pk.n_gt_cells;
            if (n_cells > 0) {
                const int nb = cdiv(n_cells, 256);
                int* block_off = nullptr;
                TRYC(dev_alloc(c, &block_off, (size_t)nb));
                int* cells = h.pair_dst + (h.n_pairs - n_cells);
                hipLaunchKernelGGL(k_gram_compact_count, dim3(nb), dim3(256), 0, c->stream, (const int*)cells, n_cells, block_off);
                hipLaunchKernelGGL(k_gram_compact_scan, dim3(1), dim3(64), 0, c->stream, block_off, nb);
                hipLaunchKernelGGL(k_gram_compact, dim3(nb), dim3(256), 0, c->stream, cells, n_cells, h.ent_src, (const int*)block_off);
                HIPCHK(hipGetLastError());
            }
        }
    }
    h.dev.n = P.n; h.dev.n_pad = P.n_pad; h.dev.T = P.T; h.dev.cam_off = d_cam_off; h.dev.tile_rows = d_tile_rows;
    h.dev.cw = P.cam_width; h.dev.cpt = P.cams_per_tile;
    // tile storage of S: dense n_pad x n_pad while that is small (<= 4 GB: one address computation less per tile; measured at L / X / D:
    // 1-3 % faster than the packed form), else packed = only the structurally non-zero tiles + one zero tile (config T: 1.4 GB
    // instead of 16 GB, X: 0.1 instead of 8; the same speed at T — the factorisation is not bound by the stride of its operands).
    // XRSFM_BA_PACKED=0 / 1 forces one form.
    const char* packed_e = std::getenv("XRSFM_BA_PACKED");        // (read per set-up: the tests switch it)
    const int packed_env = packed_e ? (packed_e[0] == '0' ? 0 : 1) : -1;
    const bool packed = packed_env >= 0 ? packed_env == 1 : (size_t)P.n_pad * P.n_pad * sizeof(double) > ((size_t)4 << 30);
    h.dev.tmap = packed ? d_tmap : nullptr;
    h.dev.ld = packed ? (size_t)kNB : (size_t)P.n_pad;
    h.dev.tstride = (size_t)kNB * kNB + kNB;
    h.S_doubles = packed ? ((size_t)P.n_tiles_nz + 1) * h.dev.tstride : (size_t)P.n_pad * P.n_pad;
    if (h.S_doubles * sizeof(double) > kCholMaxBytes) return XRSFM_BA_ETOOBIG;
    h.tile_map_host = P.tile_map;
    TRYC(dev_alloc(c, &h.dev.S, h.S_doubles));
    TRYC(dev_alloc(c, &h.dev.Linv, (size_t)P.T * kNB * kNB));
    TRYC(dev_alloc(c, &h.dev.y, (size_t)P.n_pad)); TRYC(dev_alloc(c, &h.dev.rhs, (size_t)P.n_pad)); TRYC(dev_alloc(c, &h.dev.x, (size_t)P.n_pad));
    {   // one-launch backward substitution (level schedules with at least two levels; XRSFM_BA_BWD_ALL=0: one launch per level)
        const char* be = std::getenv("XRSFM_BA_BWD_ALL");        // (read per context: the A/B test switches it inside one process)
        const bool on = !(be && be[0] == '0');
        // (round 4 kept deep trees off it: with the 174 levels of a dissected photo collection a solve took 20 ms against 4 ms of per-level
        //  chunk launches — every workgroup walked its list parent first and ran its other rounds only after the parent was solved;
        //  with the lists walked from the root side (round 5, ba_chol.h) the same launch takes 0.68 ms: config T 1086 -> 970 ms)
        h.bwd_all = on && P.use_levels && !P.panel_ll && P.n_levels >= 2;
        { const char* te = std::getenv("XRSFM_BA_DEBUG_BWD_TIMEOUT"); h.bw_debug_timeout = te && te[0] == '1'; }      // (read per set-up: the test switches it)
        // XRSFM_BA_BWD_ALL=0 on a deep level schedule: one launch per level with the tiles of a column shared out over workgroups
        // (k_lv_bwd_chunk), or — XRSFM_BA_BWD_CHUNK=0 as well — the push form of the panel schedules, two columns per launch
        const char* bce = std::getenv("XRSFM_BA_BWD_CHUNK");
        const bool deep = P.use_levels && !P.panel_ll && !h.bwd_all && P.n_levels > 32;
        h.bwd_chunk = deep && !(bce && bce[0] == '0');
        h.bwd_push = deep && !h.bwd_chunk;
        if (h.bwd_chunk) {
            std::vector<int4> chunks;
            h.bc_off.assign(P.n_levels + 1, 0);
            int max_lv = 1;
            for (int lv = 0; lv < P.n_levels; ++lv) {
                for (int e2 = P.lv_k_off[lv]; e2 < P.lv_k_off[lv + 1]; ++e2) {
                    const int q0 = P.lv_bptr[e2], q1 = P.lv_bptr[e2 + 1];
                    const int nch = std::max(1, (q1 - q0 + kBwdChunk - 1) / kBwdChunk);
                    for (int j = 0; j < nch; ++j) chunks.push_back(make_int4(e2, q0 + j * kBwdChunk, std::min(q1, q0 + (j + 1) * kBwdChunk), j | (nch << 16)));
                }
                h.bc_off[lv + 1] = (int)chunks.size();
                max_lv = std::max(max_lv, h.bc_off[lv + 1] - h.bc_off[lv]);
            }
            TRYC(dev_upload(c, &h.bc_chunks, chunks));
            TRYC(dev_alloc(c, &h.bc_part, (size_t)max_lv * kNB)); TRYC(dev_alloc(c, &h.bc_ctr, (size_t)P.T));
            HIPCHK(hipMemsetAsync(h.bc_ctr, 0, sizeof(unsigned) * (size_t)P.T, c->stream));
        }
        if (h.bwd_all) {
            std::vector<int> order;
            std::vector<unsigned char> fin(P.T, 0);
            for (int lv = P.n_levels - 2; lv >= 0; --lv)
                for (int e2 = P.lv_k_off[lv]; e2 < P.lv_k_off[lv + 1]; ++e2) order.push_back(e2);
            for (int e2 = P.lv_k_off[P.n_levels - 1]; e2 < P.lv_k_off[P.n_levels]; ++e2) fin[P.lv_k[e2]] = 1;
            h.bw_n = (int)order.size(); h.bw_launches = 0;
            TRYC(dev_upload(c, &h.bw_order, order)); TRYC(dev_upload(c, &h.bw_final, fin));
            TRYC(dev_alloc(c, &h.bw_gx, (size_t)P.T * 2 * kNB)); TRYC(dev_alloc(c, &h.bw_ctr, (size_t)2));
            HIPCHK(hipMemsetAsync(h.bw_gx, 0, sizeof(unsigned long long) * (size_t)P.T * 2 * kNB, c->stream));      // tag 0 = never written
            HIPCHK(hipMemsetAsync(h.bw_ctr, 0, sizeof(unsigned) * 2, c->stream));                                   // [0] tickets, [1] error word
        }
    }
#undef TRYC
    HIPCHK(hipMemsetAsync(h.dev.S, 0, sizeof(double) * h.S_doubles, c->stream));
    const int shm = 2 * kNB * kLdT * (int)sizeof(double);
    set_kernel_attributes(c->device);
    if ((h.n_pairs_other > 0 || h.la_depth > 0) && !h.aux) {              // lives in the context's recycled bundle (xrsfm_ba_destroy hands it back)
        if (hipStreamCreateWithFlags(&h.aux, hipStreamNonBlocking) != hipSuccess) h.aux = nullptr;
        if (h.aux && (hipEventCreateWithFlags(&h.ev_fork, hipEventDisableTiming) != hipSuccess ||
                      hipEventCreateWithFlags(&h.ev_join, hipEventDisableTiming) != hipSuccess)) {
            if (h.ev_fork) (void)hipEventDestroy(h.ev_fork);
            (void)hipStreamDestroy(h.aux); h.aux = nullptr; h.ev_fork = h.ev_join = nullptr;
        }
    }
    if (h.la_depth > 0) {
        // one event per level and direction (created once per context, destroyed with it): "level l is factored" for the second
        // stream, "the early chunks of level l are done" for the main one
        bool ok = h.aux != nullptr;
        h.la_ev_factor.assign(h.n_levels, nullptr); h.la_ev_early.assign(h.n_levels, nullptr);
        for (int lv = 0; lv < h.n_levels && ok; ++lv)
            ok = hipEventCreateWithFlags(&h.la_ev_factor[lv], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&h.la_ev_early[lv], hipEventDisableTiming) == hipSuccess;
        if (!ok) {          // no second stream / no events: every chunk in the main stream's launch, early ones first (still a valid order)
            for (hipEvent_t e2 : h.la_ev_factor) if (e2) (void)hipEventDestroy(e2);
            for (hipEvent_t e2 : h.la_ev_early) if (e2) (void)hipEventDestroy(e2);
            h.la_ev_factor.clear(); h.la_ev_early.clear();
        }
    }
    timer.mark("allocations + attributes");
    h.ready = true;
    return 0;
}

// Assemble the reduced camera matrix (after prepare_step): block values, and — unless the fused level schedule composes the
// tiles inside its first launch (chol_factor_solve) — the dense tile storage.  materialize: always fill the tiles here
// (diagnostics that read S before the factorisation).
int chol_assemble(xrsfm_ba_context* c, bool materialize = false) {
    Dev& d = c->d;
    CholHost& h = c->chol;
    const int n_obs_pairs = h.n_pairs - c->pk.n_gt_cells;
    {   // one pass over the items = up to three launches (Gram tiles per LDS class; everything else, concurrently on a second
        // stream: it is usually tiny and would otherwise add its full latency); the profile counts the pass
        Timed t_(c, K_SCHUR_PAIRS);
        const bool fork = h.n_pairs_other > 0 && h.aux;
        const double radius = c->step_radius;
        auto launch_other = [&](hipStream_t st) {
            const int* items = h.pairs_items + h.n_pairs_small + h.n_pairs_big;
            double* pv = h.pair_from_v ? h.pair_v : nullptr;
            if (c->step_prep) hipLaunchKernelGGL((k_schur_pairs<false, true, 0>), dim3(h.n_pairs_other), dim3(kWave), h.pairs_shm, st, d, items, h.slot_pair_ptr, h.pair_dst, n_obs_pairs, h.scat2, radius, pv);
            else hipLaunchKernelGGL((k_schur_pairs<false, false, 0>), dim3(h.n_pairs_other), dim3(kWave), h.pairs_shm, st, d, items, h.slot_pair_ptr, h.pair_dst, n_obs_pairs, h.scat2, radius, pv);
        };
        auto launch_gram = [&](auto ni, int n, size_t shm, const int* items, hipStream_t st) {
            constexpr int NI = decltype(ni)::value;
            if (c->step_prep) hipLaunchKernelGGL((k_schur_pairs<true, true, NI>), dim3(n), dim3(kWave), shm, st, d, items, h.slot_pair_ptr, h.pair_dst, n_obs_pairs, h.scat2, radius, (double*)nullptr, h.gram4);
            else hipLaunchKernelGGL((k_schur_pairs<true, false, NI>), dim3(n), dim3(kWave), shm, st, d, items, h.slot_pair_ptr, h.pair_dst, n_obs_pairs, h.scat2, radius, (double*)nullptr, h.gram4);
        };
        // Gram tiles (round 6): ONE launch for the operand heights 1..3 — every tile fits the small LDS class in enough staging
        // passes (ba_pack.h: gram_lds_need), the height is read from the tile's camera count — on the main stream; only tiles of
        // 9-10 cameras (40 accumulator registers: 3 waves per SIMD) keep a launch of their own, next to the non-Gram items on the
        // second stream when there is one.  Until round 5: one launch per (height, LDS class), four to six on a ragged map (config R:
        // 158 + 23 + 16 + 25 + 77 us side by side on two streams).  XRSFM_BA_GRAM_MERGE=0 keeps those launches (bit-identical blocks:
        // tests/test_gpu_parity.py).
        int n_merged = 0; size_t shm_merged = 0;
        for (int b = 0; b < 6; ++b) { n_merged += h.gram_n[b]; if (h.gram_n[b] > 0) shm_merged = std::max(shm_merged, h.gram_shm[b]); }
        const bool merge = h.gram_merge && n_merged > 0;
        constexpr bool gram_fork = true;
        int big = -1;
        for (int b = 0; b < 8; ++b) if (h.gram_n[b] > 0 && (big < 0 || h.gram_n[b] > h.gram_n[big])) big = b;
        if (merge) big = -2;                 // (the merged launch is the main-stream launch)
        auto launch_buckets = [&](bool main_side) {
            const int* items = h.pairs_items;
            if (merge && main_side) launch_gram(std::integral_constant<int, 0>{}, n_merged, shm_merged, items, c->stream);
            for (int b = 0; b < 8; ++b) {
                const int n = h.gram_n[b];
                const bool on_main = !(fork && gram_fork) || b == big;
                if (n > 0 && on_main == main_side && !(merge && b < 6)) {
                    hipStream_t st = on_main ? c->stream : h.aux;
                    switch (b >> 1) {
                        case 0: launch_gram(std::integral_constant<int, 1>{}, n, h.gram_shm[b], items, st); break;
                        case 1: launch_gram(std::integral_constant<int, 2>{}, n, h.gram_shm[b], items, st); break;
                        case 2: launch_gram(std::integral_constant<int, 3>{}, n, h.gram_shm[b], items, st); break;
                        default: launch_gram(std::integral_constant<int, 4>{}, n, h.gram_shm[b], items, st); break;
                    }
                }
                items += n;
            }
        };
        // (round 6) the main-stream launch — the one that takes the time — is issued FIRST: until round 5 the side-stream launches were,
        // and the big kernel started 20 us after the first small one (config R trace: other @ 22 us, 10-camera tiles @ 32, main @ 43)
        if (fork) {
            HIPCHK(hipEventRecord(h.ev_fork, c->stream));
            launch_buckets(true);
            HIPCHK(hipStreamWaitEvent(h.aux, h.ev_fork, 0));
            launch_other(h.aux);
            launch_buckets(false);
            HIPCHK(hipEventRecord(h.ev_join, h.aux));
            HIPCHK(hipStreamWaitEvent(c->stream, h.ev_join, 0));
        } else {
            if (h.n_pairs_other > 0) launch_other(c->stream);
            launch_buckets(true);
        }
    }
    if (d.n_cams + h.n_blocks > 0) {
        if (h.pair_from_v) LAUNCH(c, K_BLOCK_SEGSUM, k_chol_segsum_v, dim3(d.n_cams + (h.n_blocks + 3) / 4), dim3(kBlock), 0, d.scat, d.cam_ptr_g, d.camS, d.n_cams, h.scat2, h.blk_ptr, h.Sblk,
                                  h.n_blocks, (const int2*)h.ent_src, (const double*)h.pair_v);
        else LAUNCH(c, K_BLOCK_SEGSUM, k_chol_segsum, dim3(d.n_cams + h.n_blocks), dim3(kBlock), 0, d.scat, d.cam_ptr_g, d.camS, d.n_cams, h.scat2, h.blk_ptr, h.Sblk);
    }
    int e = allreduce(c, d.camS, (size_t)d.n_cams * 28 + (size_t)h.n_blocks * 36, kNcclSum);   // camS | Sblk are contiguous
    if (e) return e;
    h.S_filled = materialize;          // (else the first level's factor launch composes the tiles from the block values)
    if (h.S_filled && h.n_tiles_nz > 0)
        LAUNCH(c, K_DENSE_FILL, k_tile_fill, dim3(h.n_tiles_nz), dim3(256), 0, h.dev, d, h.tiles_nz, h.tf_ptr, h.tf_ent, h.Sblk, h.blk_rc,
               c->step_prep ? c->step_radius : 0.0);
    return 0;
}

// Factor S = L L^T and solve S x = b (S and b from chol_assemble); the solution lands in d.px
// push-form backward substitution of a panel schedule: two tile columns per launch from the last one down
static void panel_backward(xrsfm_ba_context* c) {
    CholHost& h = c->chol;
    const int T = h.T;
    int k = T - 1;
    for (int p = 0; k >= 1; k -= 2, ++p) {
            const int n = h.bw2_off[p + 1] - h.bw2_off[p];
            LAUNCH(c, K_TRISOLVE, k_bwd2, dim3(1 + n), dim3(256), 0, h.dev, k, h.bw2_link[p], (const int*)(h.bw2_ent + 2 * (size_t)h.bw2_off[p]));
        }
    for (; k >= 0; --k) {
        const int ncol = h.cols_off[k + 1] - h.cols_off[k];
        LAUNCH(c, K_TRISOLVE, k_bwd, dim3(1 + ncol), dim3(256), 0, h.dev, k, h.cols_flat + h.cols_off[k]);
    }
}

constexpr int kFillRestApart = 4096;      // tiles outside the first level's columns from which a fill launch of their own pays (config T: 40 000)

int chol_factor_solve(xrsfm_ba_context* c) {
    Dev& d = c->d;
    CholHost& h = c->chol;
    const int T = h.T;
    double* px_out = c->wide ? c->w.px : d.px;        // solution in camera order, cw values per camera
    const size_t shm = 2 * (size_t)kNB * kLdT * sizeof(double);
    if (h.lookahead) {
        // Look-ahead panel schedule (ba_plan.h, k_panel_slot): level = column, one launch per column s =
        //   factor of column s (starts from the late partials of column s-2, adds column s-1 itself)
        //   | fixed-order sum of the partial products of column s+1 into its tiles   | late partials of column s+1 (column s-1)
        //   | partial products of column s+2 over the columns < s.
        // The partial buffers alternate with the parity of the column they belong to.
        auto chunks = [&](int lv) { return lv < h.n_levels ? h.sp_chunk_off[lv + 1] - h.sp_chunk_off[lv] : 0; };
        auto targets = [&](int lv) { return lv < h.n_levels ? h.sp_rt_off[lv + 1] - h.sp_rt_off[lv] : 0; };
        auto lates = [&](int lv) { return lv < h.n_levels ? h.md_off[lv + 1] - h.md_off[lv] : 0; };
        double* const Wbuf[2] = {h.sp_work, h.sp_work + (size_t)std::max(1, h.sp_max_chunks) * kPartStride};
        double* const Qbuf[2] = {h.md_work, h.md_work + (size_t)std::max(1, h.md_max) * kPartStride};
        for (int s2 = 0; s2 < h.n_levels; ++s2) {
            const int nf = h.fz_off[s2 + 1] - h.fz_off[s2];
            SlotArgs a{};
            const int lr = s2 + 1, lp = s2 + 2;
            a.n_reduce = targets(lr);
            if (a.n_reduce > 0) { a.sp_rt = h.sp_rt + 2 * (size_t)h.sp_rt_off[lr]; a.sp_rp = h.sp_rp + 2 * (size_t)h.sp_rt_off[lr]; a.Wr = Wbuf[lr & 1]; }
            a.n_part = chunks(lp);
            if (a.n_part > 0) { a.sp_tgt = h.sp_tgt + 2 * (size_t)h.sp_chunk_off[lp]; a.sp_q = h.sp_q + 2 * (size_t)h.sp_chunk_off[lp]; a.Wp = Wbuf[lp & 1]; }
            a.n_late = lates(lr);
            if (a.n_late > 0) { a.md_tgt = h.md_tgt + 2 * (size_t)h.md_off[lr]; a.md_q = h.md_q + 2 * (size_t)h.md_off[lr]; a.Wq = Qbuf[lr & 1]; }
            if (s2 == 0 && !h.S_filled) {      // (nothing else can run yet: the tiles of every column are composed here)
                LvFill lf{};
                lf.d = d; lf.f = FillLists{h.tiles_nz, h.tf_ptr, h.tf_ent, h.Sblk, h.blk_rc, c->step_prep ? c->step_radius : 0.0};
                lf.fz_q = h.fz_q; lf.rest = h.fill_rest; lf.n_factor = nf;
                if (nf + h.n_fill_rest > 0)
                    LAUNCH(c, K_POTRF, k_lv_factor<true>, dim3(nf + h.n_fill_rest), dim3(256), 0, h.dev, h.fz_tile, h.fz_dptr, h.fz_dj,
                           (const int*)h.tile_cam, (double*)nullptr, lf);
                if (a.n_reduce + a.n_part + a.n_late > 0) return XRSFM_BA_EINTERNAL;      // columns 1 and 2 have nothing older than column 0
                continue;
            }
            a.fz_tile = h.fz_tile + 2 * (size_t)h.fz_off[s2]; a.fz_dptr = h.fz_dptr + h.fz_off[s2]; a.n_factor = nf;
            a.fz_late = h.fz_late + 2 * (size_t)h.fz_off[s2]; a.Ql = Qbuf[s2 & 1];
            const int nwg = a.n_factor + a.n_reduce + a.n_late + a.n_part;
            if (nwg > 0) LAUNCH(c, K_POTRF, k_panel_slot, dim3(nwg), dim3(256), 0, h.dev, a, (const int*)h.fz_dj, (const int*)h.lv_cj, (const int*)h.md_cj, (const int*)h.tile_cam);
        }
        panel_backward(c);
        if (d.n_cams > 0) LAUNCH(c, K_SMALL, k_sol_gather, dim3(cdiv((long long)d.n_cams * h.dev.cw, 256)), dim3(256), 0, h.dev, px_out, d.n_cams);
        return 0;
    }
    if (h.use_levels || h.panel_ll) {
        // one launch per elimination-tree level (three on a split level: partial products, their fixed-order sum, then the
        // same fused kernel with empty lists), then one per level backwards
        for (int lv = 0; lv < h.n_levels; ++lv) {
            const int nt = h.lv_tgt_off[lv + 1] - h.lv_tgt_off[lv];
            const int nch = h.sp_chunk_off[lv + 1] - h.sp_chunk_off[lv], nmc = h.mp_off[lv + 1] - h.mp_off[lv] - 1;
            const int nrt = h.sp_rt_off[lv + 1] - h.sp_rt_off[lv];
            // level look-ahead (ba_plan.h): the early chunks of this level were put on the second stream when level lv - 1 - la_depth
            // had been factored; here only the late ones follow the previous level's factor kernel.  Without the second stream
            // (la_on false) all chunks run here, early ones first — the same sums.
            const bool la_on = h.la_depth > 0 && !h.la_ev_factor.empty();
            const int n_early = la_on ? h.sp_e_cnt[lv] : 0;
            double* const Wlv = h.sp_work + (h.la_depth > 0 ? (size_t)(lv % (h.la_depth + 1)) * (size_t)std::max(1, h.sp_max_chunks) * kPartStride : 0);
            if (nmc > 0)
                LAUNCH(c, K_UPDATE, k_panel2_part, dim3(nmc), dim3(256), 0, h.dev, h.mp_chunk, h.mp_wg + h.mp_off[lv], h.sp_work);
            else if (nch - n_early > 0)
                LAUNCH(c, K_UPDATE, k_ll_update_part, dim3(nch - n_early), dim3(256), 0, h.dev, h.sp_tgt + 2 * (size_t)(h.sp_chunk_off[lv] + n_early),
                       h.sp_q + 2 * (size_t)(h.sp_chunk_off[lv] + n_early), h.lv_cj, Wlv, (const int*)(h.sp_slot + h.sp_chunk_off[lv] + n_early));
            if (n_early > 0) HIPCHK(hipStreamWaitEvent(c->stream, h.la_ev_early[lv], 0));
            if (nrt > 0)
                LAUNCH(c, K_UPDATE, k_ll_update_reduce, dim3(nrt, 16), dim3(256), 0, h.dev, h.sp_rt + 2 * (size_t)h.sp_rt_off[lv],
                       h.sp_rp + 2 * (size_t)h.sp_rt_off[lv], Wlv);
            const int nf = h.fz_off[lv + 1] - h.fz_off[lv];
            // the columns of the last level have nothing below them: their backward substitution rides in the same launch
            // (a single-tile system — LBA-sized calls — is its own last level on either schedule)
            const bool with_bwd = ((!h.panel_ll && !h.bwd_push && !h.bwd_chunk) || T == 1) && lv == h.n_levels - 1;
            LvFill lf{};
            if (lv == 0 && !h.S_filled) {
                // first level: its workgroups compose their tiles from the block values (no k_tile_fill launch, no round trip
                // through S); trailing workgroups compose the tiles of all other columns
                lf.d = d; lf.f = FillLists{h.tiles_nz, h.tf_ptr, h.tf_ent, h.Sblk, h.blk_rc, c->step_prep ? c->step_radius : 0.0};
                lf.fz_q = h.fz_q; lf.rest = h.fill_rest; lf.n_factor = nf;
                // thousands of tiles outside the first level: composed by a launch of their own (several workgroups per CU)
                const bool rest_apart = h.n_fill_rest > kFillRestApart;
                if (rest_apart)
                    LAUNCH(c, K_DENSE_FILL, k_tile_fill, dim3(h.n_fill_rest), dim3(256), 0, h.dev, d, h.tiles_nz, h.tf_ptr, h.tf_ent, h.Sblk, h.blk_rc,
                           c->step_prep ? c->step_radius : 0.0, (const int*)h.fill_rest);
                const int n_rest = rest_apart ? 0 : h.n_fill_rest;
                if (nf + n_rest > 0)
                    LAUNCH(c, K_POTRF, k_lv_factor<true>, dim3(nf + n_rest), dim3(256), 0, h.dev, h.fz_tile, h.fz_dptr, h.fz_dj,
                           (const int*)h.tile_cam, with_bwd ? px_out : (double*)nullptr, lf);
            } else if (nf > 0)
                LAUNCH(c, K_POTRF, k_lv_factor<false>, dim3(nf), dim3(256), 0, h.dev, h.fz_tile + 2 * (size_t)h.fz_off[lv], h.fz_dptr + h.fz_off[lv], h.fz_dj,
                       (const int*)h.tile_cam, with_bwd ? px_out : (double*)nullptr, lf);
            if (la_on) {
                // level lv is factored: the early chunks of level lv + 1 + la_depth (every column they name has level <= lv) go to the
                // second stream, where they run next to the main stream's late chunks / sums / factor kernels of the levels in between
                const int le = lv + 1 + h.la_depth;
                const int ne = le < h.n_levels ? h.sp_e_cnt[le] : 0;
                if (ne > 0) {
                    HIPCHK(hipEventRecord(h.la_ev_factor[lv], c->stream));
                    HIPCHK(hipStreamWaitEvent(h.aux, h.la_ev_factor[lv], 0));
                    double* const Wle = h.sp_work + (size_t)(le % (h.la_depth + 1)) * (size_t)std::max(1, h.sp_max_chunks) * kPartStride;
                    {
                        Timed t_(c, K_UPDATE, -1, h.aux);
                        hipLaunchKernelGGL(k_ll_update_part, dim3(ne), dim3(256), 0, h.aux, h.dev, h.sp_tgt + 2 * (size_t)h.sp_chunk_off[le],
                                           h.sp_q + 2 * (size_t)h.sp_chunk_off[le], h.lv_cj, Wle, (const int*)(h.sp_slot + h.sp_chunk_off[le]));
                    }
                    HIPCHK(hipEventRecord(h.la_ev_early[le], h.aux));
                }
            }
        }
        if (h.panel_ll || h.bwd_push) {       // long columns: push form, one workgroup per tile of the column
            if (T == 1) return 0;               // solved inside the factor launch
            panel_backward(c);
            if (d.n_cams > 0) LAUNCH(c, K_SMALL, k_sol_gather, dim3(cdiv((long long)d.n_cams * h.dev.cw, 256)), dim3(256), 0, h.dev, px_out, d.n_cams);
            return 0;
        }
        if (h.bwd_chunk) {
            for (int lv = h.n_levels - 1; lv >= 0; --lv) {
                const int n = h.bc_off[lv + 1] - h.bc_off[lv];
                if (n > 0) LAUNCH(c, K_TRISOLVE, k_lv_bwd_chunk, dim3(n), dim3(256), 0, h.dev, (const int*)h.lv_k, (const int*)h.lv_bi, (const int*)h.tile_cam, px_out,
                                  (const int4*)(h.bc_chunks + h.bc_off[lv]), h.bc_part, h.bc_ctr);
            }
            return 0;
        }
        if (h.bwd_all && h.bw_n > 0) {
            // all remaining levels in one launch: x travels between the workgroups of the launch as tagged granules (ba_chol.h)
            const unsigned base = h.bw_launches * (unsigned)h.bw_n;       // tickets handed out so far (unsigned wrap-around is fine)
            const unsigned epoch = ++h.bw_launches;
            LAUNCH(c, K_TRISOLVE, k_lv_bwd_all, dim3(h.bw_n), dim3(256), 0, h.dev, (const int*)h.lv_k, (const int*)h.lv_bptr, (const int*)h.lv_bi,
                   (const int*)h.tile_cam, px_out, (const int*)h.bw_order, (const unsigned char*)h.bw_final, h.bw_gx, h.bw_ctr, base, epoch, h.bw_ctr + 1,
                   d.scal + S_BWD_ERR, h.bw_debug_timeout ? 64u : kBwdSpinMax, h.bw_debug_timeout ? epoch + 1u : epoch);
            return 0;
        }
        for (int lv = h.n_levels - 2; lv >= 0; --lv) {      // (the last level: inside its k_lv_factor launch)
            const int nk = h.lv_k_off[lv + 1] - h.lv_k_off[lv];
            LAUNCH(c, K_TRISOLVE, k_lv_bwd, dim3(nk), dim3(256), 0, h.dev, h.lv_k + h.lv_k_off[lv], h.lv_bptr + h.lv_k_off[lv], h.lv_bi, h.tile_cam, px_out);
        }
        return 0;
    }
    return XRSFM_BA_EINTERNAL;      // (every plan has a level or a panel schedule)
}

// back-substitute, build the candidate state, evaluate its cost; scalars end up in h_scal
// The candidate view of the problem: state = the candidate cameras / points, linearisation buffers = the alternate set,
// partial sums in the upper half of `part` (the back-substitution's partials in the lower half are still to be reduced).
Dev candidate_view(const xrsfm_ba_context* c) {
    Dev v = c->d;
    std::swap(v.cam, v.cam_cand); std::swap(v.P, v.P_cand);
    v.rt = c->alt.rt; v.Jp = c->alt.Jp; v.camrec = c->alt.camrec; v.Hpp = c->alt.Hpp; v.gp = c->alt.gp; v.camlin = c->alt.camlin;
    v.part = c->d.part + 4 * (size_t)c->d.n_items;
    return v;
}
void accept_candidate(xrsfm_ba_context* c) {
    Dev& d = c->d;
    std::swap(d.cam, d.cam_cand); std::swap(d.P, d.P_cand);
    std::swap(d.rt, c->alt.rt); std::swap(d.Jp, c->alt.Jp); std::swap(d.camrec, c->alt.camrec);
    std::swap(d.Hpp, c->alt.Hpp); std::swap(d.gp, c->alt.gp); std::swap(d.camlin, c->alt.camlin);
}

// Back-substitution -> candidate state, then either
//   speculate: linearisation AT the candidate (its cost is the candidate cost; if the step is accepted the next iteration
//              starts from it and nothing else has to run), or
//   otherwise: a cost-only pass over the candidate (the linearisation follows only if the step is accepted).
// The caller speculates while steps are being accepted: a rejected step wastes the difference between the two passes.
// One hand-off of all scalars to the host either way.
int finish_step(xrsfm_ba_context* c, double huber_a, bool speculate) {
    Dev& d = c->d;
    // back-substitution over the tracks; trailing workgroups turn the camera part of the solution into the candidate cameras
    // (and, when the candidate is linearised straight away, their CamLin records) next to it
    const bool cams_done = c->fused;
    {
        const int nbi = cdiv(d.n_items, kWavesPerBlock), nbc = cams_done ? cdiv(d.n_cams, kBlock) : 0;
        CamLin* cr = (cams_done && speculate) ? c->alt.camrec : (CamLin*)nullptr;
        if (nbi + nbc > 0) {
            if (c->step_prep) LAUNCH(c, K_BACKSUB, k_backsub<true>, dim3(nbi + nbc), dim3(kBlock), 0, d, nbi, cr, c->step_radius);
            else LAUNCH(c, K_BACKSUB, k_backsub<false>, dim3(nbi + nbc), dim3(kBlock), 0, d, nbi, cr, c->step_radius);
        }
    }
    if (d.n_cams > 0 && !cams_done) LAUNCH(c, K_SMALL, k_cam_update, dim3(cdiv(d.n_cams, kBlock)), dim3(kBlock), 0, d);
    if (speculate) {
        const Dev cand = candidate_view(c);
        int e = linearize(c, huber_a, cand, true, LIN_FINAL | (cams_done ? LIN_SKIP_CAMLIN : 0));
        if (e) return e;
        if ((e = gradient_max_enqueue(c, cand))) return e;
        return fetch_scalars(c);
    }
    if (d.n_items > 0) LAUNCH(c, K_COST, k_cost, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, d, huber_a);
    {
        ReduceJobs j{};
        const double* ins[5] = {d.part, d.part + 2 * (size_t)d.n_items, d.part + 3 * (size_t)d.n_items, d.campart, d.campart + d.n_cams};
        const int ns[5] = {d.n_items, d.n_items, d.n_items, d.n_cams, d.n_cams};
        double* outs[5] = {d.scal + S_COST_CAND, d.scal + S_MODEL, d.scal + S_STEP2_PTS, d.scal + S_STEP2_CAMS, d.scal + S_XNORM2_CAMS};
        for (int q = 0; q < 5; ++q) { j.in[q] = ins[q]; j.n[q] = ns[q]; j.out[q] = outs[q]; j.op[q] = 0; }
        LAUNCH(c, K_SMALL, k_reduce_multi, dim3(5), dim3(kPcgThreads), 0, j);
    }
    int e = allreduce(c, d.scal + S_MODEL, 3, kNcclSum);   // MODEL, STEP2_PTS, COST_CAND adjacent
    if (e) return e;
    return fetch_scalars(c);
}

// k_lv_bwd_all bounds its spins and raises a device word when one gives up (a granule that never arrived): checked once per
// solve, after the stream has drained (level schedules with >= 2 levels only: never an LBA-sized call).
int bwd_all_status(xrsfm_ba_context* c) {
    CholHost& h = c->chol;
    if (!h.bwd_all || h.bw_launches == 0 || !h.bw_ctr) return 0;
    unsigned ev = 0;
    if (hipMemcpy(&ev, h.bw_ctr + 1, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
    if (ev) {
        fprintf(stderr, "[xrsfm_ba] backward substitution: a hand-off between workgroups timed out (k_lv_bwd_all); rerun with XRSFM_BA_BWD_ALL=0\n");
        (void)hipMemset(h.bw_ctr + 1, 0, sizeof(unsigned));
        return XRSFM_BA_EINTERNAL;
    }
    return 0;
}

void print_progress(const xrsfm_ba_options& o, int it, double cost, double change, double gmax, double step, double rho, double radius) {
    if (!o.verbose) return;
    if (it == 0) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius\n");
    printf("%4d  %.6e  %9.2e  %9.2e  %9.2e  %9.2e  %9.2e\n", it, cost, change, gmax, step, rho, radius);
}

int init_scaling_and_linearize(xrsfm_ba_context* c, double huber_a, bool use_scaling) {
    Dev& d = c->d;
    int e;
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_cams * 6, kBlock) + 1), dim3(kBlock), 0, d.scale_c, 1.0, (size_t)d.n_cams * 6);
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_pts * 3, kBlock) + 1), dim3(kBlock), 0, d.scale_p, 1.0, (size_t)d.n_pts * 3);
    c->gradmax_done = false; c->published = false;
    if ((e = linearize(c, huber_a, d, false, use_scaling ? 0 : LIN_FINAL))) return e;
    if (use_scaling) {
        // point norms are local to the rank that owns the track; camera norms were all-reduced in linearize()
        const long long n = std::max((long long)d.n_cams * 6, (long long)d.n_pts * 3);
        LAUNCH(c, K_SMALL, k_scale_from_norms, dim3(cdiv(n, kBlock) + 1), dim3(kBlock), 0, d);
        if ((e = linearize(c, huber_a, d, false, LIN_FINAL))) return e;
    }
    c->linearized = true;
    return 0;
}

}  // namespace

// ---------------------------------------------------------------- C-ABI
// No C++ exception may cross the C boundary: host allocations that fail (std::bad_alloc from the packing, the plans or the
// host solvers) become XRSFM_BA_ENOMEM, anything else XRSFM_BA_EINTERNAL (a negative code like every other error).
template <typename F>
static int no_throw(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return XRSFM_BA_ENOMEM; }
    catch (...) { return XRSFM_BA_EINTERNAL; }
}

bool Reaper::push(xrsfm_ba_context* c) {
    try {
        std::lock_guard<std::mutex> g(mu);
        if (stop || q.size() + busy >= kReaperBacklog) return false;
        if (!started) { th = std::thread([this] { loop(); }); started = true; }
        q.push_back(c);
    } catch (...) { return false; }          // no thread / no memory: the caller releases in place
    cv.notify_all();
    return true;
}
void Reaper::loop() {
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
        cv.wait(lk, [this] { return stop || !q.empty(); });
        if (q.empty()) { if (stop) return; continue; }
        xrsfm_ba_context* c = q.front();
        q.pop_front();
        ++busy;
        lk.unlock();
        delete c;
        lk.lock();
        --busy;
        cv.notify_all();
    }
}
void Reaper::drain() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return q.empty() && busy == 0; });
}

extern "C" {

void xrsfm_ba_default_options(xrsfm_ba_options* o) {
    if (!o) return;
    o->max_iterations = 50;            // ba_solver.cc:627
    o->function_tolerance = 1e-5;      // :628
    o->parameter_tolerance = 1e-6;     // :629
    o->gradient_tolerance = 1e-10;     // Ceres default
    o->initial_radius = 1e4;           // Ceres default
    o->huber_a = 5.99;                 // :343
    o->linear_solver = XRSFM_BA_SOLVER_AUTO;
    o->pcg_tolerance = 1e-12;
    o->pcg_max_iterations = 2000;
    o->profile = 0;
    o->verbose = 0;
}

int xrsfm_ba_version(int* n_devices) {
    if (n_devices) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        *n_devices = n;
    }
    return XRSFM_BA_VERSION;
}

// First call of a process (profiles/r04_adapter_timing.txt: `create` 113 ms against 5.8 ms for the second call): HIP start-up, the
// code object of this library (75 kernels + the rocPRIM sorts, loaded by the first launch), a stream with its pinned scalar block,
// ~100 hipMalloc calls.  None of it depends on the problem — only the sizes of the buffers do, hence the hints.
int xrsfm_ba_warmup(int device, int64_t n_obs_hint, int64_t n_points_hint, int64_t n_cams_hint) {
    return no_throw([&]() -> int {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return XRSFM_BA_ENODEV;
        if (device < 0 || device >= ndev) return XRSFM_BA_EINVAL;
        HIPCHK(hipSetDevice(device));
        HostBundle hb;
        if (!g_bundles.get(device, &hb)) return XRSFM_BA_ENODEV;
        set_kernel_attributes(device);
        // the first launch loads the code object; device packing and the key generation bring rocPRIM's kernels with them (same object)
        double* probe = nullptr;
        size_t cls = 0;
        probe = (double*)g_cache.get(device, 256 * sizeof(double), &cls);
        int rc = XRSFM_BA_OK;
        if (!probe) rc = XRSFM_BA_ENOMEM;
        else {
            hipLaunchKernelGGL(k_fill, dim3(1), dim3(kBlock), 0, hb.stream, probe, 0.0, (size_t)256);
            if (hipStreamSynchronize(hb.stream) != hipSuccess) rc = XRSFM_BA_ENODEV;
            g_cache.put(device, probe, cls);
        }
        g_bundles.put(device, hb);
        if (rc != XRSFM_BA_OK || n_obs_hint <= 0) return rc;
        // Device buffers of a problem of about this size: the sizes xrsfm_ba_create / the Cholesky set-up ask for, in units of the
        // slot count (observations + tile padding), the point and the camera count; every block goes straight back to the cache.
        const size_t ns = (size_t)(n_obs_hint + n_obs_hint / 8 + 64), np = (size_t)std::max<int64_t>(n_points_hint, 1), nc = (size_t)std::max<int64_t>(n_cams_hint, 1);
        std::vector<size_t> want;
        auto add = [&](size_t bytes, int count = 1) { for (int i = 0; i < count; ++i) want.push_back(bytes); };
        add(ns * 8 * 6, 2);                 // Jp (both linearisation sets)
        add(ns * 8 * 2, 2);                 // rt
        add(ns * 8 * 28);                   // camera-major scatter buffer
        add(ns * 8, 2); add(ns * 4, 5); add(ns, 1);        // slot streams: u, v | cam, pt, campos, campos_g, obs | cidx
        add(np * 8 * 6, 4); add(np * 8 * 3, 7);            // Hpp x2, Hinv, Hc | gp x2, scale_p, yp, P, P_cand, P0
        add(nc * 128, 6);                   // camera records
        add(ns * 8 * 36 / 10);              // block scatter buffer (Gram cells: ~one 6x6 entry per ten observations)
        add(ns * 8, 6); add(ns * 16, 2);    // scratch of the device-side sorts
        //  the hints are the caller's guess: what they pin in the allocation cache is capped at a quarter of the
        // device memory that is free right now — a wrong hint must not take the memory the real problem needs
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return XRSFM_BA_OK;
        const size_t budget = free_b / 4;
        size_t taken = 0;
        std::vector<std::pair<void*, size_t>> got;
        for (size_t b : want) {
            if (taken + b > budget) break;
            size_t c2 = 0; void* q = g_cache.get(device, b, &c2); if (!q) break; got.push_back({q, c2}); taken += b;
        }
        for (auto& g2 : got) g_cache.put(device, g2.first, g2.second);
        return XRSFM_BA_OK;
    });
}

void xrsfm_ba_destroy(xrsfm_ba_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->poisoned) {
        // The watchdog gave up on this context's stream (a collective no peer joined): waiting for it here would only move the
        // hang.  The communicator is aborted (not destroyed: ncclCommDestroy waits for pending work), and the stream, its pinned
        // scalar block and the device buffers the stuck kernels may still touch are deliberately leaked — nothing of them goes
        // back to the caches.
        if (c->comm && g_rccl.CommAbort) g_rccl.CommAbort(c->comm);
        delete c;
        return;
    }
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->chol.aux) (void)hipStreamSynchronize(c->chol.aux);
    for (hipEvent_t e : c->chol.la_ev_factor) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->chol.la_ev_early) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (size_t i = 0; i < c->allocs.size(); ++i) g_cache.put(c->device, c->allocs[i], c->alloc_class[i]);
    if (c->stream) g_bundles.put(c->device, HostBundle{c->stream, c->h_scal, c->h_st, c->chol.aux, c->chol.ev_fork, c->chol.ev_join});
    // What is left is host memory.  A large context holds ~0.5 KB per observation in vectors whose release (munmap: page-table
    // teardown) takes ~10 ms per million observations: ONE reaper thread does it, the caller (one BA call of a mapper) goes on.
    // The thread is joined when the library is unloaded (g_reaper's destructor: dlclose / process exit), so no library code
    // runs after the unload; at most kReaperBacklog contexts wait, a further one is released in place.  Small contexts are
    // deleted in place (a hand-over costs more than their release).
    if (c->pk.n_obs > 200000 && g_reaper.push(c)) return;
    delete c;
}

int xrsfm_ba_quiesce(uint64_t* cached_bytes) {
    g_reaper.drain();
    size_t n;
    { std::lock_guard<std::mutex> g(g_cache.mu); n = g_cache.cached_bytes; }
    if (cached_bytes) *cached_bytes = (uint64_t)n;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return XRSFM_BA_OK;      // nothing can be cached without a device
    g_cache.release_all();
    g_pinned.release_all();
    return XRSFM_BA_OK;
}

int xrsfm_ba_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return XRSFM_BA_ENODEV;
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return XRSFM_BA_OK;
}

static int create_body(const xrsfm_ba_problem* p, int device, xrsfm_ba_context* c, xrsfm_ba_context** out);
static thread_local int g_force_device_pack = -1;       // xrsfm_ba_debug_device_pack_check: 1 = device packing whatever the size

int xrsfm_ba_create(const xrsfm_ba_problem* p, int device, xrsfm_ba_context** out) {
    if (!p || !out) return XRSFM_BA_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[xrsfm_ba] no HIP device visible: the BA path has no CPU fallback\n");
        return XRSFM_BA_ENODEV;
    }
    if (device < 0 || device >= ndev) return XRSFM_BA_EINVAL;
    xrsfm_ba_context* c = new (std::nothrow) xrsfm_ba_context();
    if (!c) return XRSFM_BA_ENOMEM;
    try { return create_body(p, device, c, out); }          // (the body releases c itself on the errors it returns)
    catch (const std::bad_alloc&) { xrsfm_ba_destroy(c); *out = nullptr; return XRSFM_BA_ENOMEM; }
    catch (...) { xrsfm_ba_destroy(c); *out = nullptr; return XRSFM_BA_EINTERNAL; }
}

static int create_body(const xrsfm_ba_problem* p, int device, xrsfm_ba_context* c, xrsfm_ba_context** out) {
    c->device = device;
    PhaseTimer timer("create");
    for (int i = 0; i < p->n_cams && p->cam_const; ++i) c->wide = c->wide || (p->cam_const[i] & kCamIntrVariable) != 0;
    int e = pack_validate(*p, c->wide);
    if (e) { delete c; return e; }
    {
        HostBundle hb;
        if (hipSetDevice(device) != hipSuccess || !g_bundles.get(device, &hb)) { delete c; return XRSFM_BA_ENODEV; }
        c->stream = hb.stream; c->h_scal = hb.h_scal; c->h_st = hb.h_st;
        c->chol.aux = hb.aux; c->chol.ev_fork = hb.ev_fork; c->chol.ev_join = hb.ev_join;
        c->seq = 0;
        *reinterpret_cast<unsigned long long*>(c->h_scal + S_COUNT) = 0;
        void* dp = nullptr;
        c->h_scal_dev = (hipHostGetDevicePointer(&dp, c->h_scal, 0) == hipSuccess) ? static_cast<double*>(dp) : nullptr;
    }
    Dev& d = c->d;
    // Packing: on the device for problems large enough to pay for its launches and round trips (ba_pack_dev.h: the same arrays as the
    // host's, element for element), on the host otherwise — small calls (LBA: a few thousand observations, 0.2 ms of host packing),
    // bal9 mode, tracks longer than 64 observations, 65 535 cameras or more.  XRSFM_BA_DEVICE_PACK=0 / 1: never / whenever possible.
    devpack::Result dres;
    {
        const char* dpe = std::getenv("XRSFM_BA_DEVICE_PACK");
        const bool allowed = !c->wide && p->n_cams < 65535 && p->n_obs > 0 && p->n_points > 0;
        const bool want = g_force_device_pack >= 0 ? g_force_device_pack == 1 : (dpe ? dpe[0] != '0' : p->n_obs >= 32768);      // (tools/pack_crossover.py: create + set-up 1.2 vs 1.5 ms at 48 k observations, 1.0 vs 0.75 at 20 k)
        if (allowed && want) {
            std::vector<std::pair<void*, size_t>> scratch;
            auto keep = [&](size_t bytes) -> void* { unsigned char* q = nullptr; return dev_alloc(c, &q, bytes) ? nullptr : (void*)q; };
            auto scr = [&](size_t bytes) -> void* { size_t cls = 0; void* q = g_cache.get(c->device, bytes, &cls); if (q) scratch.push_back({q, cls}); return q; };
            e = devpack::device_pack(*p, c->stream, keep, scr, c->pk, dres);
            (void)hipStreamSynchronize(c->stream);
            for (auto& b : scratch) g_cache.put(c->device, b.first, b.second);
            if (e == 0) { c->dev_packed = true; c->host_pack_level = 0; c->dpk_slot_obs = dres.slot_obs; c->dpk_gt_cell = dres.gt_cell; }
            else if (e != 1) { xrsfm_ba_destroy(c); return e; }
            else { c->pk = Packed(); }                    // (a track longer than 64 observations: the host path packs it)
        }
    }
    if (!c->dev_packed) {
        e = pack_problem(*p, c->pk, c->wide);
        if (e) { xrsfm_ba_destroy(c); return e; }
    }
    timer.mark("pack_problem");
    const Packed& k = c->pk;
    c->n_points_caller = p->n_points;
    d.n_cams = k.n_cams; d.n_pts = k.n_pts; d.n_tiles = k.n_tiles; d.n_slots = k.n_slots; d.n_items = (int)k.items.size() / 2;
    std::vector<CamRec> cams(k.n_cams);
    std::vector<int> model(k.n_cams);
    std::vector<unsigned char> cconst(k.n_cams);
    for (int i = 0; i < k.n_cams; ++i) {
        CamRec& r = cams[i];
        for (int j = 0; j < 4; ++j) r.q[j] = p->cam_q[4 * (size_t)i + j];
        for (int j = 0; j < 3; ++j) r.t[j] = p->cam_t[3 * (size_t)i + j];
        r.pad = 0.0;
        const int ii = p->cam_intr[i];
        for (int j = 0; j < 8; ++j) r.intr[j] = p->intr_params[8 * (size_t)ii + j];
        model[i] = p->intr_model[ii];
        cconst[i] = p->cam_const ? p->cam_const[i] : 0;
    }
    RawVec<double> P(c->dev_packed ? 0 : 3 * (size_t)k.n_pts);
    if (!c->dev_packed)
    pack_parallel_for(k.n_pts, [&](long long j0, long long j1) {
        for (long long j = j0; j < j1; ++j)
            for (int a = 0; a < 3; ++a) P[3 * (size_t)j + a] = p->points[3 * (size_t)k.pt_orig[j] + a];
    });
    std::vector<double> cam_act(k.n_cams);
    for (int i = 0; i < k.n_cams; ++i) cam_act[i] = (k.cam_ptr[i + 1] > k.cam_ptr[i]) ? 1.0 : 0.0;
    timer.mark("stream + host staging");
#define TRY(x) do { e = (x); if (e) { xrsfm_ba_destroy(c); return e; } } while (0)
    {
        // (const members of Dev are set through a cast: the arrays are written exactly once, here)
        BatchUpload up(c);
        auto P_ = [](auto& member) { return const_cast<std::remove_const_t<std::remove_pointer_t<std::remove_reference_t<decltype(member)>>>**>(&member); };
        up.add_raw(reinterpret_cast<void**>(const_cast<Item**>(&d.items)), k.items.data(), k.items.size() * sizeof(int));
        up.add(P_(d.cam), cams); up.add(P_(d.cam_cand), cams); up.add(&c->cam0, cams);
        up.add(P_(d.cam_model), model); up.add(P_(d.cam_const), cconst); up.add(P_(d.cam_act), cam_act);
        if (c->dev_packed) {
            d.slot_cam = dres.slot_cam; d.slot_pt = dres.slot_pt; d.slot_campos = dres.slot_campos; d.slot_u = dres.slot_u; d.slot_v = dres.slot_v;
            d.tile_stride = dres.tile_stride; d.tile_maxlen = dres.tile_maxlen; d.tile_ncam = dres.tile_ncam; d.tile_gt_off = dres.tile_gt_off;
            d.slot_cidx = dres.slot_cidx; d.slot_campos_g = dres.slot_campos_g; d.cam_ptr_g = dres.cam_ptr_g; d.cam_ptr = dres.cam_ptr;
            d.P = dres.P; d.pt_const = dres.pt_const;
            const size_t pb = sizeof(double) * 3 * (size_t)k.n_pts;
            TRY(dev_alloc(c, &d.P_cand, 3 * (size_t)k.n_pts)); TRY(dev_alloc(c, &c->P0, 3 * (size_t)k.n_pts));
            if (pb && (hipMemcpyAsync(d.P_cand, d.P, pb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess ||
                       hipMemcpyAsync(c->P0, d.P, pb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)) { xrsfm_ba_destroy(c); return XRSFM_BA_ENODEV; }
        } else {
            up.add(P_(d.slot_cam), k.slot_cam); up.add(P_(d.slot_pt), k.slot_pt); up.add(P_(d.slot_campos), k.slot_campos);
            up.add(P_(d.slot_u), k.slot_u); up.add(P_(d.slot_v), k.slot_v);
            up.add(P_(d.tile_stride), k.tile_stride); up.add(P_(d.tile_maxlen), k.tile_maxlen);
            up.add(P_(d.tile_ncam), k.tile_ncam); up.add(P_(d.tile_gt_off), k.tile_gt_off); up.add(P_(d.slot_cidx), k.slot_cidx);
            up.add(P_(d.slot_campos_g), k.slot_campos_g); up.add(P_(d.cam_ptr_g), k.cam_ptr_g);
            up.add(P_(d.cam_ptr), k.cam_ptr);
            up.add(P_(d.P), P); up.add(P_(d.P_cand), P); up.add(&c->P0, P);
            up.add(P_(d.pt_const), k.pt_const);
        }
        TRY(up.flush());
    }
    timer.mark("uploads");
    const size_t ns = (size_t)k.n_slots, nc = (size_t)k.n_cams, np = (size_t)k.n_pts;
    TRY(dev_alloc(c, &d.scale_c, nc * 6)); TRY(dev_alloc(c, &d.scale_p, np * 3));
    TRY(dev_alloc(c, &d.rt, ns * 2)); TRY(dev_alloc(c, &d.Jp, ns * 6)); TRY(dev_alloc(c, &d.camrec, nc));
    TRY(dev_alloc(c, &c->alt.rt, ns * 2)); TRY(dev_alloc(c, &c->alt.Jp, ns * 6)); TRY(dev_alloc(c, &c->alt.camrec, nc));
    TRY(dev_alloc(c, &d.Hpp, np * 6)); TRY(dev_alloc(c, &d.gp, np * 3)); TRY(dev_alloc(c, &c->alt.Hpp, np * 6)); TRY(dev_alloc(c, &c->alt.gp, np * 3)); TRY(dev_alloc(c, &d.Hinv, np * 6)); TRY(dev_alloc(c, &d.Hc, np * 6));
    TRY(dev_alloc(c, &d.camlin, nc * 12 + 4 + kMaxRanks)); TRY(dev_alloc(c, &c->alt.camlin, nc * 12 + 4 + kMaxRanks)); TRY(dev_alloc(c, &d.Dc2, nc * 6)); TRY(dev_alloc(c, &d.camS, nc * 28));
    TRY(dev_alloc(c, &d.Minv, nc * 21)); TRY(dev_alloc(c, &d.b, nc * 6));
    TRY(dev_alloc(c, &d.px, nc * 6 + kNB)); TRY(dev_alloc(c, &d.pr, nc * 6)); TRY(dev_alloc(c, &d.pz, nc * 6));
    TRY(dev_alloc(c, &d.pp, nc * 6)); TRY(dev_alloc(c, &d.pq, nc * 6));
    TRY(dev_alloc(c, &d.yp, np * 3));
    TRY(dev_alloc(c, &d.scat, (size_t)(k.n_obs > 0 ? k.n_obs : 1) * 28));
    TRY(dev_alloc(c, &d.part, (size_t)d.n_items * 7));      // [0,4n): own view (linearise 0..3n, back-substitution 2n..4n); [4n,7n): candidate view
    TRY(dev_alloc(c, &d.campart, nc * 2));
    TRY(dev_alloc(c, &d.ptpart, np / kBlock + 2));
    TRY(dev_alloc(c, &d.pcgpart, nc * (3 + kGauge)));
    TRY(dev_alloc(c, &c->pcg_w, nc * 6 * 2 * kGauge + kGauge * kGauge));      // W | (S + D^2) W | inverse of the coarse matrix
    d.pcgW = nullptr; d.pcgSW = nullptr; d.pcgE = nullptr;
    TRY(dev_alloc(c, &d.scal, (size_t)S_COUNT));
    TRY(dev_alloc(c, &d.st, (size_t)1));
    if (c->wide) {      // 9-wide path: stored Jacobian blocks, wide scale / sums / solution (ba_wide.h)
        TRY(dev_alloc(c, &c->w.Fw, ns * 18)); TRY(dev_alloc(c, &c->w.Ew, ns * 6));
        TRY(dev_alloc(c, &c->w.scale_c, nc * kW)); TRY(dev_alloc(c, &c->w.camlin, nc * 18));
        TRY(dev_alloc(c, &c->w.px, nc * kW + kNB));
        TRY(dev_alloc(c, &c->w.scat, (size_t)(k.n_obs > 0 ? k.n_obs : 1) * kWS));
        c->cam_intr_host.assign(p->cam_intr, p->cam_intr + p->n_cams);
    }
    TRY(dev_alloc(c, &c->part2, (size_t)kTailJobs * kTailGrid));
    TRY(dev_alloc(c, &c->ticket, (size_t)32 * 9));
    {   // camera-sorted deposit positions and per-camera runs of the Gram tiles (k_gram_runs, ba_kernels.h): once per context
        unsigned char* gpos = nullptr; int* trun = nullptr;
        TRY(dev_alloc(c, &gpos, ns ? ns : 1)); TRY(dev_alloc(c, &trun, (size_t)std::max(1, k.n_tiles) * kTileRunLd));
        if (k.n_tiles > 0) hipLaunchKernelGGL(k_gram_runs, dim3(cdiv(k.n_tiles, 4)), dim3(256), 0, c->stream, d.slot_cam, d.slot_cidx, d.tile_ncam, k.n_tiles, gpos, trun);
        d.slot_gpos = gpos; d.tile_run = trun;
    }
#undef TRY
    {
        const char* fz = std::getenv("XRSFM_BA_FUSED");
        c->fused = !(fz && fz[0] == '0');
        const char* pf = std::getenv("XRSFM_BA_PREP_FUSED");
        c->prep_fused = !(pf && pf[0] == '0');
        const char* pc = std::getenv("XRSFM_BA_PCG_COARSE");
        c->pcg_coarse = !(pc && pc[0] == '0');
    }
    // (the scatter buffers need no clearing: every entry is written before it is read)
    if (hipMemsetAsync(d.scal, 0, sizeof(double) * S_COUNT, c->stream) != hipSuccess || hipMemsetAsync(d.st, 0, sizeof(PcgStatus), c->stream) != hipSuccess ||
        hipMemsetAsync(c->ticket, 0, sizeof(unsigned) * 32 * 9, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { xrsfm_ba_destroy(c); return XRSFM_BA_ENODEV; }
    timer.mark("work buffers");
    *out = c;
    return XRSFM_BA_OK;
}

int xrsfm_ba_comm_unique_id(unsigned char id[128]) {
    if (!id) return XRSFM_BA_EINVAL;
    if (!load_rccl()) return XRSFM_BA_ECOMM;
    UniqueId u;
    if (g_rccl.GetUniqueId(&u) != 0) return XRSFM_BA_ECOMM;
    memcpy(id, u.internal, 128);
    return 0;
}

int xrsfm_ba_comm_init(xrsfm_ba_context* c, int n_ranks, int rank, const unsigned char id[128]) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || !id || n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks) return XRSFM_BA_EINVAL;
    // a single rank needs no communicator; XRSFM_BA_FORCE_COMM=1 creates one anyway (exercises the RCCL plumbing
    // on a 1-GPU box: every all-reduce then really goes through ncclAllReduce)
    const char* force = getenv("XRSFM_BA_FORCE_COMM");
    if (n_ranks == 1 && !(force && force[0] == '1')) { c->n_ranks = 1; c->rank = 0; return 0; }
    if (!load_rccl()) return XRSFM_BA_ECOMM;
    HIPCHK(hipSetDevice(c->device));
    UniqueId u;
    memcpy(u.internal, id, 128);
    void* comm = nullptr;
    if (g_rccl.CommInitRank(&comm, n_ranks, u, rank) != 0) return XRSFM_BA_ECOMM;
    c->comm = comm; c->n_ranks = n_ranks; c->rank = rank;
    // a camera is part of the program if ANY rank holds an observation of it
    int e = allreduce(c, c->d.cam_act, (size_t)c->d.n_cams, kNcclMax);
    if (e) return e;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int xrsfm_ba_debug_comm_hook(xrsfm_ba_context* c, int n_ranks, int rank, xrsfm_ba_allreduce_fn fn, void* user) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || !fn || n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks || c->comm) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    c->hook = fn; c->hook_user = user; c->n_ranks = n_ranks; c->rank = rank;
    int e = allreduce(c, c->d.cam_act, (size_t)c->d.n_cams, kNcclMax);      // as in xrsfm_ba_comm_init
    if (e) return e;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int xrsfm_ba_reset(xrsfm_ba_context* c) {
    if (!c) return XRSFM_BA_EINVAL;
    if (c->poisoned) return XRSFM_BA_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(c->d.cam, c->cam0, sizeof(CamRec) * (size_t)c->d.n_cams, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d.P, c->P0, sizeof(double) * 3 * (size_t)c->d.n_pts, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->linearized = false; c->step_valid = false;
    return 0;
}

int xrsfm_ba_download(xrsfm_ba_context* c, double* cam_q, double* cam_t, double* points) {
    if (!c) return XRSFM_BA_EINVAL;
    if (c->poisoned) return XRSFM_BA_ESTATE;       // (a blocking copy would wait for the stream the watchdog gave up on)
    HIPCHK(hipSetDevice(c->device));
    const Packed& k = c->pk;
    // (round 6) both copies land in ONE pinned block from the recycled pool (cameras | points) and are permuted from there
    const size_t cam_bytes = (cam_q || cam_t) ? sizeof(CamRec) * (size_t)k.n_cams : 0;
    const size_t off_P = (cam_bytes + 255) & ~(size_t)255, pt_bytes = points ? sizeof(double) * 3 * (size_t)k.n_pts : 0;
    if (cam_bytes + pt_bytes == 0) return 0;
    size_t cap = 0;
    unsigned char* host = static_cast<unsigned char*>(g_pinned.get(off_P + pt_bytes, &cap));
    if (!host) return XRSFM_BA_ENOMEM;
    bool ok = true;
    if (cam_bytes) ok = hipMemcpyAsync(host, c->d.cam, cam_bytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    if (ok && pt_bytes) ok = hipMemcpyAsync(host + off_P, c->d.P, pt_bytes, hipMemcpyDeviceToHost, c->stream) == hipSuccess;
    ok = (hipStreamSynchronize(c->stream) == hipSuccess) && ok;
    if (ok && cam_bytes) {
        const CamRec* cams = reinterpret_cast<const CamRec*>(host);
        for (int i = 0; i < k.n_cams; ++i) {
            if (cam_q) for (int j = 0; j < 4; ++j) cam_q[4 * (size_t)i + j] = cams[i].q[j];
            if (cam_t) for (int j = 0; j < 3; ++j) cam_t[3 * (size_t)i + j] = cams[i].t[j];
        }
    }
    if (ok && pt_bytes) {
        const double* P = reinterpret_cast<const double*>(host + off_P);
        pack_parallel_for(k.n_pts, [&](long long j0, long long j1) {      // back to the caller's point order
            for (long long j = j0; j < j1; ++j)
                for (int a = 0; a < 3; ++a) points[3 * (size_t)k.pt_orig[j] + a] = P[3 * (size_t)j + a];
        });
    }
    g_pinned.put(host, cap);
    return ok ? 0 : XRSFM_BA_ENODEV;
}

// bal9 mode: the same trust-region loop over the 9-wide kernels of ba_wide.h.  Plain schedule — linearise, assemble,
// factor, back-substitute, cost of the candidate, linearise again after an accepted step — through the launch-per-phase tail.
static int linearize_wide(xrsfm_ba_context* c, double huber_a, bool scaled_pass) {
    Dev& d = c->d;
    (void)scaled_pass;
    c->gradmax_done = false; c->published = false;
    if (d.n_items > 0) LAUNCH(c, K_LINEARIZE, k9_linearize, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, d, c->w, huber_a);
    if (d.n_cams > 0) LAUNCH(c, K_CAM_SEGSUM, k_cam_segsum<18>, dim3(d.n_cams), dim3(kBlock), 0, c->w.scat, d.cam_ptr_g, c->w.camlin, (const PcgStatus*)nullptr);
    ReduceJobs j{};
    const double* ins[3] = {d.part, d.part + d.n_items, d.part + 2 * (size_t)d.n_items};
    double* outs[3] = {d.scal + S_COST, d.scal + S_XNORM2_PTS, d.scal + S_GRADMAX_PTS};
    for (int q = 0; q < 3; ++q) { j.in[q] = ins[q]; j.n[q] = d.n_items; j.out[q] = outs[q]; j.op[q] = q == 2 ? 1 : 0; }
    LAUNCH(c, K_SMALL, k_reduce_multi, dim3(3), dim3(kPcgThreads), 0, j);
    return 0;
}

// S assembly in bal9 mode: Gram tiles per (operand height, LDS class) bucket (k9_pairs_gram), every other item through the
// per-pair kernel, then the fixed-order sums over the S assembly's camera-major entries (cam_ptr_g) and the tile fill.
static int assemble_wide(xrsfm_ba_context* c, double radius) {
    Dev& d = c->d;
    CholHost& h = c->chol;
    const int n_obs_pairs = h.n_pairs - c->pk.n_gt_cells;
    {
        Timed t_(c, K_SCHUR_PAIRS);
        const int* items = h.pairs_items;
        for (int b = 0; b < 8; ++b) {
            const int n = h.gram_n[b];
            if (n > 0) {
                switch (b >> 1) {
                    case 0: hipLaunchKernelGGL(k9_pairs_gram<1>, dim3(n), dim3(kWave), h.gram_shm[b], c->stream, d, c->w, items, (const int*)h.pair_dst, n_obs_pairs, h.scat2, radius); break;
                    case 1: hipLaunchKernelGGL(k9_pairs_gram<2>, dim3(n), dim3(kWave), h.gram_shm[b], c->stream, d, c->w, items, (const int*)h.pair_dst, n_obs_pairs, h.scat2, radius); break;
                    case 2: hipLaunchKernelGGL(k9_pairs_gram<3>, dim3(n), dim3(kWave), h.gram_shm[b], c->stream, d, c->w, items, (const int*)h.pair_dst, n_obs_pairs, h.scat2, radius); break;
                    default: hipLaunchKernelGGL(k9_pairs_gram<4>, dim3(n), dim3(kWave), h.gram_shm[b], c->stream, d, c->w, items, (const int*)h.pair_dst, n_obs_pairs, h.scat2, radius); break;
                }
            }
            items += n;
        }
        if (h.n_pairs_other > 0)
            hipLaunchKernelGGL(k9_pairs, dim3(cdiv(h.n_pairs_other, kWavesPerBlock)), dim3(kBlock), 0, c->stream, d, c->w, items, h.n_pairs_other,
                               (const int*)h.slot_pair_ptr, (const int*)h.pair_dst, h.scat2, radius);
    }
    if (d.n_cams + h.n_blocks > 0) LAUNCH(c, K_BLOCK_SEGSUM, k9_chol_segsum, dim3(d.n_cams + h.n_blocks), dim3(kBlock), 0, c->w.scat, d.cam_ptr_g, c->w.camS, d.n_cams, h.scat2, h.blk_ptr, h.Sblk);
    if (h.n_tiles_nz > 0) LAUNCH(c, K_DENSE_FILL, k9_tile_fill, dim3(h.n_tiles_nz), dim3(256), 0, h.dev, d, c->w, h.tiles_nz, h.tf_ptr, h.tf_ent, h.Sblk, h.blk_rc, radius);
    h.S_filled = true;
    return 0;
}

static int run_wide(xrsfm_ba_context* c, const xrsfm_ba_options& opt, xrsfm_ba_summary* sum) {
    Dev& d = c->d;
    CholHost& h = c->chol;
    hipStream_t st = c->stream;
    int e;
    sum->linear_solver_used = XRSFM_BA_SOLVER_CHOLESKY;
    c->profiling = opt.profile != 0;
    for (int i = 0; i < K_COUNT; ++i) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; }
    c->recs.clear(); c->ev_used = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    sum->num_residuals = 2 * c->pk.n_obs;
    {
        int n_var_i = 0;
        std::vector<unsigned char> cc(d.n_cams);
        std::vector<double> act(d.n_cams);
        if (d.n_cams) { HIPCHK(hipMemcpy(cc.data(), d.cam_const, d.n_cams, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(act.data(), d.cam_act, sizeof(double) * d.n_cams, hipMemcpyDeviceToHost)); }
        for (int i = 0; i < d.n_cams; ++i) n_var_i += (act[i] > 0.0 && (cc[i] & kCamIntrVariable)) ? 1 : 0;
        sum->num_effective_params = 3 * (c->pk.n_var_q + c->pk.n_var_t + c->pk.n_var_p + n_var_i);
    }
    auto finish = [&](int term, int reason, double cost) {
        sum->termination = term; sum->termination_reason = reason; sum->final_cost = cost;
        (void)hipStreamSynchronize(st);
        sum->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (int be = bwd_all_status(c)) return be;
        if (c->profiling) {
            profile_collect(c);
            int best = 0;
            for (int i = 1; i < K_COUNT - 1; ++i) if (c->prof_ms[i] > c->prof_ms[best]) best = i;
            sum->dom_kernel_id = best; sum->dom_kernel_ms = c->prof_ms[best]; sum->dom_kernel_launches = c->prof_n[best];
        }
        c->profiling = false;
        return XRSFM_BA_OK;
    };
    auto gradmax_fetch = [&](double* gmax) {
        LAUNCH(c, K_SMALL, k9_gradmax_cams, dim3(1), dim3(kPcgThreads), 0, d, c->w, d.scal + S_GRADMAX_CAMS);
        int e2 = fetch_scalars(c);
        if (e2) return e2;
        *gmax = std::fmax(c->h_scal[S_GRADMAX_PTS], c->h_scal[S_GRADMAX_CAMS]);
        return 0;
    };
    // iteration 0: Jacobi scaling from the unscaled column norms, then the scaled linearisation
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_cams * kW, kBlock) + 1), dim3(kBlock), 0, c->w.scale_c, 1.0, (size_t)d.n_cams * kW);
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_pts * 3, kBlock) + 1), dim3(kBlock), 0, d.scale_p, 1.0, (size_t)d.n_pts * 3);
    if ((e = linearize_wide(c, opt.huber_a, false))) return e;
    {
        const long long n = std::max((long long)d.n_cams * kW, (long long)d.n_pts * 3);
        LAUNCH(c, K_SMALL, k9_scale_from_norms, dim3(cdiv(n, kBlock) + 1), dim3(kBlock), 0, d, c->w);
    }
    if ((e = linearize_wide(c, opt.huber_a, true))) return e;
    c->linearized = true;
    double gmax = 0.0;
    if ((e = gradmax_fetch(&gmax))) return e;
    double cost = 0.5 * c->h_scal[S_COST];
    double xnorm2_pts = c->h_scal[S_XNORM2_PTS];
    sum->initial_cost = cost;
    double radius = opt.initial_radius, decrease = 2.0;
    print_progress(opt, 0, cost, 0.0, gmax, 0.0, 0.0, radius);
    if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
    int it = 0, invalid = 0;
    const double max_radius = 1e16, min_radius = 1e-32, min_rel_decrease = 1e-3;
    const int n_obs_pairs = h.n_pairs - c->pk.n_gt_cells;
    while (true) {
        if (it >= opt.max_iterations) return finish(XRSFM_BA_NO_CONVERGENCE, 5, cost);
        ++it;
        sum->lm_steps_attempted++;
        // reduced camera system: per-observation diagonal terms + per-pair blocks, fixed-order sums, tile fill, tile Cholesky
        (void)n_obs_pairs;
        if ((e = assemble_wide(c, radius))) return e;
        if ((e = chol_factor_solve(c))) return e;
        {   // back-substitution + candidate state, cost of the candidate, the scalars of the step
            const int nbi = cdiv(d.n_items, kWavesPerBlock), nbc = cdiv(d.n_cams, kBlock);
            if (nbi + nbc > 0) LAUNCH(c, K_BACKSUB, k9_backsub, dim3(nbi + nbc), dim3(kBlock), 0, d, c->w, nbi, radius);
            if (d.n_items > 0) LAUNCH(c, K_COST, k_cost, dim3(cdiv(d.n_items, kWavesPerBlock)), dim3(kBlock), 0, d, opt.huber_a);
            ReduceJobs j{};
            const double* ins[5] = {d.part, d.part + 2 * (size_t)d.n_items, d.part + 3 * (size_t)d.n_items, d.campart, d.campart + d.n_cams};
            const int ns[5] = {d.n_items, d.n_items, d.n_items, d.n_cams, d.n_cams};
            double* outs[5] = {d.scal + S_COST_CAND, d.scal + S_MODEL, d.scal + S_STEP2_PTS, d.scal + S_STEP2_CAMS, d.scal + S_XNORM2_CAMS};
            for (int q = 0; q < 5; ++q) { j.in[q] = ins[q]; j.n[q] = ns[q]; j.out[q] = outs[q]; j.op[q] = 0; }
            LAUNCH(c, K_SMALL, k_reduce_multi, dim3(5), dim3(kPcgThreads), 0, j);
            if ((e = fetch_scalars(c))) return e;
        }
        const double* s = c->h_scal;
        const double model_change = s[S_MODEL];
        const double xnorm = std::sqrt(xnorm2_pts + s[S_XNORM2_CAMS]);
        if (!(model_change > 0.0) || !std::isfinite(model_change)) {
            ++invalid;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, 0.0, gmax, 0.0, 0.0, radius);
            if (invalid >= 5) return finish(XRSFM_BA_FAILURE, 6, cost);
            radius /= decrease; decrease *= 2.0;
            continue;
        }
        invalid = 0;
        const double cost_cand = 0.5 * s[S_COST_CAND];
        const double step_norm = std::sqrt(s[S_STEP2_PTS] + s[S_STEP2_CAMS]);
        if (step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance)) return finish(XRSFM_BA_CONVERGENCE, 2, cost);
        const double cost_change = cost - cost_cand;
        if (std::fabs(cost_change) <= opt.function_tolerance * cost) return finish(XRSFM_BA_CONVERGENCE, 3, cost);
        const double rel = cost_change / model_change;
        if (rel > min_rel_decrease) {
            std::swap(d.cam, d.cam_cand);
            std::swap(d.P, d.P_cand);
            if ((e = linearize_wide(c, opt.huber_a, true))) return e;
            if ((e = gradmax_fetch(&gmax))) return e;
            cost = 0.5 * c->h_scal[S_COST];
            xnorm2_pts = c->h_scal[S_XNORM2_PTS];
            radius = std::fmin(max_radius, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0;
            sum->n_successful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
        } else {
            radius /= decrease; decrease *= 2.0;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (radius < min_radius) return finish(XRSFM_BA_CONVERGENCE, 4, cost);
        }
    }
}

static int ba_run_impl(xrsfm_ba_context* c, const xrsfm_ba_options* optp, xrsfm_ba_summary* sum) {
    if (!c || !optp || !sum) return XRSFM_BA_EINVAL;
    if (c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context: only xrsfm_ba_destroy is left
    const xrsfm_ba_options opt = *optp;
    HIPCHK(hipSetDevice(c->device));
    if (const char* se = std::getenv("XRSFM_BA_DEBUG_STALL_S"))          // test hook: see k_debug_stall
        c->debug_stall_ticks = (unsigned long long)(std::fmin(std::fmax(std::atof(se), 0.0), 10.0) * 1e8);
    memset(sum, 0, sizeof(*sum));
    Dev& d = c->d;
    hipStream_t st = c->stream;
    int solver = opt.linear_solver;
    int e;
    if (c->wide) {      // bal9 mode: exact solver, one rank
        if (solver == XRSFM_BA_SOLVER_PCG || c->multi()) {
            fprintf(stderr, "[xrsfm_ba] variable intrinsics (9-wide camera blocks): only the exact solver on one rank is implemented\n");
            return XRSFM_BA_EINVAL;
        }
        solver = XRSFM_BA_SOLVER_CHOLESKY;
    }
    if (solver == XRSFM_BA_SOLVER_AUTO) {
        // exact tile Cholesky whenever its plan is feasible (always up to kCholMaxN unknowns; larger problems when the camera
        // graph is a band / ring), implicit-Schur PCG otherwise
        try { e = chol_setup(c); } catch (const std::bad_alloc&) { e = XRSFM_BA_ENOMEM; }     // (pair keys / plan too large for the host: PCG)
        if (e == XRSFM_BA_ETOOBIG || e == XRSFM_BA_ENOMEM || e == kErrDuplicateObs) solver = XRSFM_BA_SOLVER_PCG;
        else if (e) return e;
        else solver = XRSFM_BA_SOLVER_CHOLESKY;
    }
    if (solver != XRSFM_BA_SOLVER_PCG && solver != XRSFM_BA_SOLVER_CHOLESKY) return XRSFM_BA_EINVAL;
    if (solver == XRSFM_BA_SOLVER_CHOLESKY && (e = chol_setup(c))) {
        if (e == kErrDuplicateObs) {
            fprintf(stderr, "[xrsfm_ba] a track is observed twice by one camera: the explicit reduced camera matrix is not available, use AUTO or PCG\n");
            return XRSFM_BA_EINVAL;
        }
        return e;
    }
    if (c->wide) return run_wide(c, opt, sum);
    sum->linear_solver_used = solver;
    c->profiling = opt.profile != 0;
    for (int i = 0; i < K_COUNT; ++i) { c->prof_ms[i] = 0.0; c->prof_n[i] = 0; }
    c->recs.clear(); c->ev_used = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    sum->num_residuals = 2 * c->pk.n_obs;
    sum->num_effective_params = 3 * (c->pk.n_var_q + c->pk.n_var_t + c->pk.n_var_p);
    auto finish = [&](int term, int reason, double cost) {
        sum->termination = term; sum->termination_reason = reason; sum->final_cost = cost;
        (void)hipStreamSynchronize(st);
        sum->total_time_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (int be = bwd_all_status(c)) return be;
        if (c->profiling) {
            profile_collect(c);
            int best = 0;
            for (int i = 1; i < K_COUNT - 1; ++i) if (c->prof_ms[i] > c->prof_ms[best]) best = i;
            sum->dom_kernel_id = best; sum->dom_kernel_ms = c->prof_ms[best]; sum->dom_kernel_launches = c->prof_n[best];
        }
        c->profiling = false;
        return XRSFM_BA_OK;
    };
    // iteration 0: Jacobi scaling from the unscaled column norms, then the scaled linearisation
    if ((e = init_scaling_and_linearize(c, opt.huber_a, true))) return e;
    double gmax = 0.0;
    if ((e = gradient_max(c, &gmax))) return e;     // also fetches the scalars
    double cost = 0.5 * c->h_scal[S_COST];
    double xnorm2_pts = c->h_scal[S_XNORM2_PTS];
    sum->initial_cost = cost;
    double radius = opt.initial_radius, decrease = 2.0;
    print_progress(opt, 0, cost, 0.0, gmax, 0.0, 0.0, radius);
    if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
    int it = 0, invalid = 0;
    const double max_radius = 1e16, min_radius = 1e-32, min_rel_decrease = 1e-3;
    // While steps are being accepted, finish_step() linearises at the candidate straight away: one host round trip per LM
    // iteration hands over the model decrease and step norms of the step AND the cost, |x|^2 and gradient max-norm at the
    // candidate.  After a rejected step (they come in runs while the radius collapses) only the cost of the candidate is
    // evaluated, and the linearisation follows once a step is accepted again; once a solve has seen a rejection, two accepted
    // steps in a row are needed before the next one is linearised ahead again (near convergence accepted and rejected steps
    // alternate, and a wasted linearisation costs more than a saved cost pass).
    bool speculate = true;
    int accepted_run = 0;
    while (true) {
        if (it >= opt.max_iterations) return finish(XRSFM_BA_NO_CONVERGENCE, 5, cost);
        ++it;
        sum->lm_steps_attempted++;
        if ((e = prepare_step(c, radius, solver == XRSFM_BA_SOLVER_CHOLESKY))) return e;
        if (solver == XRSFM_BA_SOLVER_PCG) {
            if ((e = pcg_solve(c, opt, sum))) return e;
        } else {
            if ((e = chol_assemble(c))) return e;
            if ((e = chol_factor_solve(c))) return e;
        }
        if ((e = finish_step(c, opt.huber_a, speculate))) return e;
        const double* s = c->h_scal;
        const double model_change = s[S_MODEL];
        const double xnorm = std::sqrt(xnorm2_pts + s[S_XNORM2_CAMS]);
        if (!(model_change > 0.0) || !std::isfinite(model_change)) {
            ++invalid;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, 0.0, gmax, 0.0, 0.0, radius);
            if (invalid >= 5) return finish(XRSFM_BA_FAILURE, 6, cost);
            radius /= decrease; decrease *= 2.0;
            continue;
        }
        invalid = 0;
        const double cost_cand = 0.5 * (speculate ? s[S_COST] : s[S_COST_CAND]);    // (S_COST: of the linearisation at the candidate)
        const double step_norm = std::sqrt(s[S_STEP2_PTS] + s[S_STEP2_CAMS]);
        if (step_norm <= opt.parameter_tolerance * (xnorm + opt.parameter_tolerance))
            return finish(XRSFM_BA_CONVERGENCE, 2, cost);
        const double cost_change = cost - cost_cand;
        if (std::fabs(cost_change) <= opt.function_tolerance * cost) return finish(XRSFM_BA_CONVERGENCE, 3, cost);
        const double rel = cost_change / model_change;
        if (rel > min_rel_decrease) {
            if (speculate) {
                accept_candidate(c);
                cost = cost_cand;
                xnorm2_pts = s[S_XNORM2_PTS];
                gmax = std::fmax(s[S_GRADMAX_PTS], s[S_GRADMAX_CAMS]);
            } else {
                std::swap(d.cam, d.cam_cand);
                std::swap(d.P, d.P_cand);
                if ((e = linearize(c, opt.huber_a, d, false, LIN_FINAL))) return e;
                if ((e = gradient_max(c, &gmax))) return e;
                cost = 0.5 * c->h_scal[S_COST];
                xnorm2_pts = c->h_scal[S_XNORM2_PTS];
            }
            ++accepted_run;
            speculate = sum->n_unsuccessful == 0 || accepted_run >= 2;
            radius = std::fmin(max_radius, radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3)));
            decrease = 2.0;
            sum->n_successful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (gmax <= opt.gradient_tolerance) return finish(XRSFM_BA_CONVERGENCE, 1, cost);
        } else {
            speculate = false; accepted_run = 0;
            radius /= decrease; decrease *= 2.0;
            sum->n_unsuccessful++;
            print_progress(opt, it, cost, cost_change, gmax, step_norm, rel, radius);
            if (radius < min_radius) return finish(XRSFM_BA_CONVERGENCE, 4, cost);
        }
    }
}

int xrsfm_ba_profile_entry(xrsfm_ba_context* c, int index, const char** name, double* total_ms, int* launches) {
    if (!c || index < 0 || index >= K_COUNT) return XRSFM_BA_EINVAL;
    if (name) *name = kKidName[index];
    if (total_ms) *total_ms = c->prof_ms[index];
    if (launches) *launches = c->prof_n[index];
    return 0;
}

int xrsfm_ba_solve(const xrsfm_ba_options* opt, xrsfm_ba_problem* problem, xrsfm_ba_summary* summary) {
    if (!opt || !problem || !summary) return XRSFM_BA_EINVAL;
    // XRSFM_BA_TRACE_CALLS=1: one stderr line per call (sizes, phases, LM steps) — how a mapper run spends its BA time
    static const bool trace = std::getenv("XRSFM_BA_TRACE_CALLS") != nullptr;
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    xrsfm_ba_context* c = nullptr;
    int e = xrsfm_ba_create(problem, 0, &c);
    if (e) return e;
    const auto t1 = clk::now();
    e = xrsfm_ba_run(c, opt, summary);
    const auto t2 = clk::now();
    if (!e) e = xrsfm_ba_download(c, problem->cam_q, problem->cam_t, problem->points);
    if (!e && problem->intr_params) e = xrsfm_ba_download_intrinsics(c, problem->intr_params);      // (bal9 mode only)
    const auto t3 = clk::now();
    xrsfm_ba_destroy(c);
    if (trace) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "[xrsfm_ba_solve] cams %d points %d obs %d | create %.3f run %.3f download %.3f destroy %.3f ms | LM %d+%d solver %d rc %d\n",
                     problem->n_cams, problem->n_points, problem->n_obs, ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, clk::now()),
                     summary->n_successful, summary->n_unsuccessful, summary->linear_solver_used, e);
    }
    return e;
}

// ---------------------------------------------------------------- scaled pose graph, host only
void xrsfm_pg_default_options(xrsfm_pg_options* o) {
    if (!o) return;
    o->max_iterations = 100; o->function_tolerance = 1e-6; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
    o->initial_radius = 1e16; o->verbose = 0; o->bounds_active_set = 0;
}

static int pg_solve_impl(const xrsfm_pg_options* opt, xrsfm_pg_problem* p, xrsfm_pg_summary* summary) {
    if (!p || !summary) return XRSFM_BA_EINVAL;
    if (p->n_frames < 0 || p->n_scales < p->n_frames || p->n_edges < 0 || p->n_scale_costs < 0) return XRSFM_BA_EINVAL;
    if (p->n_frames > 0 && (!p->rot_q || !p->pos)) return XRSFM_BA_EINVAL;
    if (p->n_scales > 0 && !p->scale) return XRSFM_BA_EINVAL;
    if (p->n_edges > 0 && (!p->edge_a || !p->edge_b || !p->edge_sa || !p->edge_sb || !p->edge_q_mea || !p->edge_p_mea)) return XRSFM_BA_EINVAL;
    if (p->n_scale_costs > 0 && (!p->sc_a || !p->sc_b || !p->sc_s12)) return XRSFM_BA_EINVAL;
    for (int e = 0; e < p->n_edges; ++e) {
        if (p->edge_a[e] < 0 || p->edge_a[e] >= p->n_frames || p->edge_b[e] < 0 || p->edge_b[e] >= p->n_frames) return XRSFM_BA_EINVAL;
        if (p->edge_sa[e] < 0 || p->edge_sa[e] >= p->n_scales || p->edge_sb[e] < 0 || p->edge_sb[e] >= p->n_scales) return XRSFM_BA_EINVAL;
    }
    for (int e = 0; e < p->n_scale_costs; ++e)
        if (p->sc_a[e] < 0 || p->sc_a[e] >= p->n_scales || p->sc_b[e] < 0 || p->sc_b[e] >= p->n_scales) return XRSFM_BA_EINVAL;
    xrsfm_pg_options o;
    if (opt) o = *opt; else xrsfm_pg_default_options(&o);
    xpg::Solver solver(*p);
    return solver.run(o, summary);
}

// ---------------------------------------------------------------- tag refinement (host code)
void xrsfm_tag_default_options(xrsfm_pg_options* o) {
    if (!o) return;
    o->max_iterations = 500;            // tag_extract.hpp:231
    o->function_tolerance = 1e-6; o->parameter_tolerance = 1e-8; o->gradient_tolerance = 1e-10;
    o->initial_radius = 1e4; o->verbose = 0; o->bounds_active_set = 0;
}

static int tag_refine_impl(const xrsfm_pg_options* opt, xrsfm_tag_problem* p, int32_t stages, xrsfm_pg_summary* summaries) {
    if (!p || !summaries || stages < 1 || stages > 2) return XRSFM_BA_EINVAL;
    if (p->n_frames < 0 || p->n_tags < 0 || p->n_tag_obs < 0 || p->n_points < 0 || p->n_obs < 0) return XRSFM_BA_EINVAL;
    if (p->n_frames > 0 && (!p->frame_q || !p->frame_t)) return XRSFM_BA_EINVAL;
    if (p->n_tags > 0 && (!p->tag_corners || !p->tag_q || !p->tag_t)) return XRSFM_BA_EINVAL;
    if (p->n_tag_obs > 0 && (!p->tag_obs_tag || !p->tag_obs_frame || !p->tag_obs_xy)) return XRSFM_BA_EINVAL;
    if (p->n_points > 0 && !p->points) return XRSFM_BA_EINVAL;
    if (p->n_obs > 0 && (!p->obs_frame || !p->obs_pt || !p->obs_xy)) return XRSFM_BA_EINVAL;
    if (!(p->tag_length > 0.0) || !std::isfinite(p->scale)) return XRSFM_BA_EINVAL;
    for (int i = 0; i < p->n_tag_obs; ++i)
        if (p->tag_obs_tag[i] < 0 || p->tag_obs_tag[i] >= p->n_tags || p->tag_obs_frame[i] < 0 || p->tag_obs_frame[i] >= p->n_frames) return XRSFM_BA_EINVAL;
    for (int i = 0; i < p->n_obs; ++i)
        if (p->obs_pt[i] < 0 || p->obs_pt[i] >= p->n_points || p->obs_frame[i] < 0 || p->obs_frame[i] >= p->n_frames) return XRSFM_BA_EINVAL;
    xrsfm_pg_options o;
    if (opt) o = *opt; else xrsfm_tag_default_options(&o);
    return xtag::refine(o, *p, stages, summaries);
}

// ---------------------------------------------------------------- pose-only refinement
void xrsfm_ba_refine_pose_options(xrsfm_ba_options* o) {
    if (!o) return;
    xrsfm_ba_default_options(o);
    o->max_iterations = 10;             // pnp.cc:59
    o->function_tolerance = 1e-6;       // ceres::Solver::Options defaults
    o->parameter_tolerance = 1e-8;
    o->gradient_tolerance = 1e-10;
    o->initial_radius = 1e4;
    o->huber_a = 5.99;                  // pnp.cc:49
}

// The same problem through the general engine (one camera, every point constant): kept as the cross-check of the persistent
// kernel (XRSFM_BA_REFINE_ENGINE=1 selects it; tests compare the two).
static int refine_pose_engine(const xrsfm_ba_options& o, int32_t model, const double* intr_params, int32_t n, const double* points3d,
                              const double* uv, const uint8_t* inlier_mask, double* q, double* t, xrsfm_ba_summary* summary);

// n_frames jobs in one upload / launch (grid.x = frames) / read-back.  corr_ptr[f]..corr_ptr[f+1] are frame f's correspondences.
static int refine_poses_kernel(const xrsfm_ba_options& o, int32_t n_frames, const int32_t* models, const double* intr_params,
                               const int32_t* corr_ptr, const double* points3d, const double* uv, const uint8_t* inlier_mask,
                               double* q, double* t, xrsfm_ba_summary* summaries) {
    const auto t_begin = std::chrono::steady_clock::now();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[xrsfm_ba] no HIP device visible: the BA path has no CPU fallback\n");
        return XRSFM_BA_ENODEV;
    }
    const int device = 0;
    HIPCHK(hipSetDevice(device));
    std::vector<int> m(n_frames, 0), first(n_frames + 1, 0);
    for (int f = 0; f < n_frames; ++f) {
        for (int i = corr_ptr[f]; i < corr_ptr[f + 1]; ++i) m[f] += (!inlier_mask || inlier_mask[i]) ? 1 : 0;
        first[f + 1] = first[f] + m[f];
    }
    const size_t M = (size_t)first[n_frames];
    // one staging block: jobs | points | observations | results
    static_assert(sizeof(RefineJob) % 8 == 0 && sizeof(RefineResult) % 8 == 0, "staging layout");
    const size_t off_P = (sizeof(RefineJob) * (size_t)n_frames + 255) & ~(size_t)255, off_uv = off_P + sizeof(double) * 3 * M;
    const size_t off_res = (off_uv + sizeof(double) * 2 * M + 255) & ~(size_t)255;
    const size_t total = off_res + sizeof(RefineResult) * (size_t)n_frames;
    size_t cls = 0;
    static const bool trace_phases = std::getenv("XRSFM_BA_TRACE_CALLS") != nullptr;
    const auto t_a = std::chrono::steady_clock::now();
    unsigned char* dev = static_cast<unsigned char*>(g_cache.get(device, total, &cls));
    if (!dev) return XRSFM_BA_ENOMEM;
    HostBundle hb;
    if (!g_bundles.get(device, &hb)) { g_cache.put(device, dev, cls); return XRSFM_BA_ENODEV; }
    const auto t_b = std::chrono::steady_clock::now();
    // (round 6) staging in PINNED memory from the recycled pool, results included: the hipMemcpyAsync of the 112-byte result into a
    // pageable vector was where the mapper replay's slow pose refinements sat — 20-28 ms inside that one call, each time right after a
    // KGBA + whole-map filter had released tens of MB of host memory (the runtime pins pageable pages on the fly; tools/runs/r06_call12.sh)
    struct Pin { unsigned char* p = nullptr; size_t cap = 0; unsigned char* data() const { return p; } ~Pin() { if (p) g_pinned.put(p, cap); } } stage;
    stage.p = (unsigned char*)g_pinned.get(total, &stage.cap);
    if (!stage.p) { g_bundles.put(device, hb); g_cache.put(device, dev, cls); return XRSFM_BA_ENOMEM; }
    double* hP = reinterpret_cast<double*>(stage.data() + off_P);
    double* hU = reinterpret_cast<double*>(stage.data() + off_uv);
    for (int f = 0; f < n_frames; ++f) {
        RefineJob job{};
        job.n = m[f]; job.model = models[f];
        job.P = reinterpret_cast<const double*>(dev + off_P) + 3 * (size_t)first[f];
        job.uv = reinterpret_cast<const double*>(dev + off_uv) + 2 * (size_t)first[f];
        for (int k = 0; k < 8; ++k) job.intr[k] = intr_params[8 * (size_t)f + k];
        for (int k = 0; k < 4; ++k) job.q[k] = q[4 * (size_t)f + k];
        for (int k = 0; k < 3; ++k) job.t[k] = t[3 * (size_t)f + k];
        memcpy(stage.data() + sizeof(RefineJob) * (size_t)f, &job, sizeof(job));
        size_t w = (size_t)first[f];
        for (int i = corr_ptr[f]; i < corr_ptr[f + 1]; ++i) {
            if (inlier_mask && !inlier_mask[i]) continue;      // only the inliers enter, like pnp.cc:43-45
            for (int k = 0; k < 3; ++k) hP[3 * w + k] = points3d[3 * (size_t)i + k];
            hU[2 * w] = uv[2 * (size_t)i]; hU[2 * w + 1] = uv[2 * (size_t)i + 1];
            ++w;
        }
    }
    RefineOpt ro{o.max_iterations, o.function_tolerance, o.parameter_tolerance, o.gradient_tolerance, o.initial_radius, o.huber_a};
    RefineResult* const res = reinterpret_cast<RefineResult*>(stage.data() + off_res);
    int e = XRSFM_BA_OK;
    double pending_ms = 0.0;
    if (trace_phases) {         // (developer aid: is the device still busy with something an earlier call left behind?)
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipDeviceSynchronize();
        pending_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    const auto t_c = std::chrono::steady_clock::now();
    // Both directions WITHOUT the copy engines: a kernel pulls the staged block from the pinned buffer (device-visible) and
    // k_refine_pose writes its result records straight into it.  With hipMemcpyAsync the same call took 20-37 ms whenever the
    // process had not used a copy engine for ~10 ms (the host-only phase after a whole-map filter): the copy, not the kernel
    // (0.1 ms in the kernel trace), was what hipStreamSynchronize waited for (tools/runs/r06_call13.sh).
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(hb.h_st);          // (pinned; the PCG status block is idle here)
    if (trace_phases) { stamps[0] = 0; stamps[1] = 0; hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, hb.stream, stamps); }
    hipLaunchKernelGGL(k_copy_words, dim3(cdiv((long long)(off_res / 8), 256)), dim3(256), 0, hb.stream, reinterpret_cast<const double*>(stage.data()),
                       reinterpret_cast<double*>(dev), off_res / 8);
    const auto t_d = std::chrono::steady_clock::now();
    if (!e) {
        hipLaunchKernelGGL(k_refine_pose, dim3(n_frames), dim3(kBlock), 0, hb.stream, reinterpret_cast<const RefineJob*>(dev), ro, res);
        if (hipGetLastError() != hipSuccess) e = XRSFM_BA_ENODEV;
    }
    if (trace_phases) hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, hb.stream, stamps + 1);
    const auto t_e = std::chrono::steady_clock::now();
    const auto t_f = std::chrono::steady_clock::now();
    double seen_ms = -1.0;
    if (trace_phases) {           // when does the last kernel's stamp become visible in pinned memory? (before hipStreamSynchronize is asked)
        volatile unsigned long long* v = stamps + 1;
        while (*v == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f).count() < 0.2) __builtin_ia32_pause();
        seen_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f).count();
    }
    if (hipStreamSynchronize(hb.stream) != hipSuccess) e = XRSFM_BA_ENODEV;
    const auto t_g = std::chrono::steady_clock::now();
    g_bundles.put(device, hb);
    g_cache.put(device, dev, cls);
    if (e) return e;
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (trace_phases && secs > 2e-3) {          // (the poll above is part of `sync` in trace mode)
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[xrsfm_ba_refine_pose] slow call: prologue %.3f | block + stream %.3f | staging %.3f (of which device sync %.3f) | upload call %.3f | launch call %.3f | download call %.3f | sync %.3f ms (%zu bytes) | on the DEVICE first kernel start -> last kernel end %.3f ms; the last kernel's stamp reached pinned memory %.3f ms after the launches\n",
                ms(t_begin, t_a), ms(t_a, t_b), ms(t_b, t_c), pending_ms, ms(t_c, t_d), ms(t_d, t_e), ms(t_e, t_f), ms(t_f, t_g), total, (double)(stamps[1] - stamps[0]) * 1e-5, seen_ms);
    }
    for (int f = 0; f < n_frames; ++f) {
        xrsfm_ba_summary* sm = summaries + f;
        memset(sm, 0, sizeof(*sm));
        for (int k = 0; k < 4; ++k) q[4 * (size_t)f + k] = res[f].q[k];
        for (int k = 0; k < 3; ++k) t[3 * (size_t)f + k] = res[f].t[k];
        sm->initial_cost = res[f].initial_cost; sm->final_cost = res[f].final_cost;
        sm->num_residuals = 2 * m[f]; sm->num_effective_params = 6;
        sm->n_successful = res[f].n_successful; sm->n_unsuccessful = res[f].n_unsuccessful;
        sm->termination = res[f].termination; sm->termination_reason = res[f].reason;
        sm->lm_steps_attempted = res[f].attempted; sm->linear_solver_used = XRSFM_BA_SOLVER_CHOLESKY;
        sm->total_time_s = secs;
    }
    return XRSFM_BA_OK;
}

int xrsfm_ba_refine_pose(const xrsfm_ba_options* opt, int32_t model, const double* intr_params, int32_t n, const double* points3d,
                         const double* uv, const uint8_t* inlier_mask, double* q, double* t, xrsfm_ba_summary* summary) {
    if (!intr_params || !q || !t || !summary || n < 0 || model < 0 || model > 4) return XRSFM_BA_EINVAL;
    if (n > 0 && (!points3d || !uv)) return XRSFM_BA_EINVAL;
    xrsfm_ba_options o;
    if (opt) o = *opt; else xrsfm_ba_refine_pose_options(&o);
    if (std::getenv("XRSFM_BA_REFINE_ENGINE")) return no_throw([&] { return refine_pose_engine(o, model, intr_params, n, points3d, uv, inlier_mask, q, t, summary); });
    const int32_t corr_ptr[2] = {0, n};
    return no_throw([&] { return refine_poses_kernel(o, 1, &model, intr_params, corr_ptr, points3d, uv, inlier_mask, q, t, summary); });
}

int xrsfm_ba_refine_poses(const xrsfm_ba_options* opt, int32_t n_frames, const int32_t* models, const double* intr_params,
                          const int32_t* corr_ptr, const double* points3d, const double* uv, const uint8_t* inlier_mask,
                          double* q, double* t, xrsfm_ba_summary* summaries) {
    if (n_frames < 0 || (n_frames > 0 && (!models || !intr_params || !corr_ptr || !q || !t || !summaries))) return XRSFM_BA_EINVAL;
    if (n_frames == 0) return XRSFM_BA_OK;
    if (corr_ptr[0] != 0) return XRSFM_BA_EINVAL;
    for (int f = 0; f < n_frames; ++f)
        if (models[f] < 0 || models[f] > 4 || corr_ptr[f + 1] < corr_ptr[f]) return XRSFM_BA_EINVAL;
    if (corr_ptr[n_frames] > 0 && (!points3d || !uv)) return XRSFM_BA_EINVAL;
    xrsfm_ba_options o;
    if (opt) o = *opt; else xrsfm_ba_refine_pose_options(&o);
    return no_throw([&] { return refine_poses_kernel(o, n_frames, models, intr_params, corr_ptr, points3d, uv, inlier_mask, q, t, summaries); });
}

static int refine_pose_engine(const xrsfm_ba_options& o, int32_t model, const double* intr_params, int32_t n, const double* points3d,
                              const double* uv, const uint8_t* inlier_mask, double* q, double* t, xrsfm_ba_summary* summary) {
    std::vector<double> P, UV;
    std::vector<int32_t> ocam, opt_idx;
    for (int i = 0; i < n; ++i) {
        if (inlier_mask && !inlier_mask[i]) continue;
        opt_idx.push_back((int32_t)(P.size() / 3));
        for (int k = 0; k < 3; ++k) P.push_back(points3d[3 * (size_t)i + k]);
        UV.push_back(uv[2 * (size_t)i]); UV.push_back(uv[2 * (size_t)i + 1]);
        ocam.push_back(0);
    }
    const int m = (int)ocam.size();
    std::vector<uint8_t> pconst(m > 0 ? m : 1, 1);
    uint8_t cconst = 0;
    int32_t cam_intr = 0, intr_model = model;
    double prm[8];
    for (int k = 0; k < 8; ++k) prm[k] = intr_params[k];
    xrsfm_ba_problem pr{};
    pr.n_cams = 1; pr.n_points = m; pr.n_obs = m; pr.n_intr = 1;
    pr.cam_q = q; pr.cam_t = t; pr.cam_const = &cconst; pr.cam_intr = &cam_intr;
    pr.intr_model = &intr_model; pr.intr_params = prm;
    pr.points = P.data(); pr.point_const = pconst.data();
    pr.obs_cam = ocam.data(); pr.obs_pt = opt_idx.data(); pr.obs_uv = UV.data();
    return xrsfm_ba_solve(&o, &pr, summary);
}

// ---------------------------------------------------------------- post-BA track filter
static int filter_tracks_impl(const xrsfm_ba_problem* p, double max_reproj_error, double min_tri_angle_rad, uint8_t* obs_delete,
                              uint8_t* track_outlier, double* track_error, double* track_angle, int32_t* num_filtered) {
    if (!p || !obs_delete || !track_outlier) return XRSFM_BA_EINVAL;
    // the same pointer checks as pack_problem (ba_pack.h)
    if (p->n_cams < 0 || p->n_points < 0 || p->n_obs < 0 || p->n_intr < 0) return XRSFM_BA_EINVAL;
    if (p->n_obs > 0 && (!p->obs_cam || !p->obs_pt || !p->obs_uv)) return XRSFM_BA_EINVAL;
    if (p->n_cams > 0 && (!p->cam_q || !p->cam_t || !p->cam_intr)) return XRSFM_BA_EINVAL;
    if (p->n_points > 0 && !p->points) return XRSFM_BA_EINVAL;
    if (p->n_intr > 0 && (!p->intr_model || !p->intr_params)) return XRSFM_BA_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        fprintf(stderr, "[xrsfm_ba] no HIP device visible: the track filter has no CPU fallback\n");
        return XRSFM_BA_ENODEV;
    }
    const int Nc = p->n_cams, Np = p->n_points, No = p->n_obs;
    if (Nc < 0 || Np < 0 || No < 0) return XRSFM_BA_EINVAL;
    for (int c2 = 0; c2 < Nc; ++c2) {
        const int ii = p->cam_intr[c2];
        if (ii < 0 || ii >= p->n_intr || p->intr_model[ii] < 0 || p->intr_model[ii] > 4) return XRSFM_BA_EINVAL;
    }
    // observations by track, inside a track by camera (= frame id order of Track::observations_)
    std::vector<int> ptr(Np + 1, 0);
    for (int i = 0; i < No; ++i) {
        if (p->obs_cam[i] < 0 || p->obs_cam[i] >= Nc || p->obs_pt[i] < 0 || p->obs_pt[i] >= Np) return XRSFM_BA_EINVAL;
        ptr[p->obs_pt[i] + 1]++;
    }
    for (int j = 0; j < Np; ++j) ptr[j + 1] += ptr[j];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1), order(No);
    for (int i = 0; i < No; ++i) order[fill[p->obs_pt[i]]++] = i;
    for (int j = 0; j < Np; ++j)
        std::stable_sort(order.begin() + ptr[j], order.begin() + ptr[j + 1], [&](int a, int b) { return p->obs_cam[a] < p->obs_cam[b]; });
    std::vector<int> ocam(No), model(Nc);
    std::vector<double> ouv(2 * (size_t)No);
    for (int k2 = 0; k2 < No; ++k2) { ocam[k2] = p->obs_cam[order[k2]]; ouv[2 * (size_t)k2] = p->obs_uv[2 * (size_t)order[k2]]; ouv[2 * (size_t)k2 + 1] = p->obs_uv[2 * (size_t)order[k2] + 1]; }
    std::vector<CamRec> cams(Nc);
    for (int i = 0; i < Nc; ++i) {
        for (int k2 = 0; k2 < 4; ++k2) cams[i].q[k2] = p->cam_q[4 * (size_t)i + k2];
        for (int k2 = 0; k2 < 3; ++k2) cams[i].t[k2] = p->cam_t[3 * (size_t)i + k2];
        cams[i].pad = 0.0;
        for (int k2 = 0; k2 < 8; ++k2) cams[i].intr[k2] = p->intr_params[8 * (size_t)p->cam_intr[i] + k2];
        model[i] = p->intr_model[p->cam_intr[i]];
    }
    HIPCHK(hipSetDevice(0));
    // (round 6) ONE device block from the allocation cache, ONE upload, the two kernels on a recycled stream, ONE download: until
    // round 5 a call cost 12 hipMalloc + 7 blocking uploads + 5 downloads + 12 hipFree — 0.31 ms for the 2 000 tracks of a frame,
    // 600 times per 300-frame reconstruction (the mapper replay's "filters" total was larger than its BA total).
    // Layout: inputs | outputs, every array 256-byte aligned; the outputs travel back as one contiguous range.
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al(bytes ? bytes : 8); return o; };
    const size_t o_cam = take(sizeof(CamRec) * (size_t)Nc), o_model = take(sizeof(int) * (size_t)Nc), o_P = take(sizeof(double) * 3 * (size_t)Np),
                 o_ptr = take(sizeof(int) * ((size_t)Np + 1)), o_ocam = take(sizeof(int) * (size_t)No), o_ouv = take(sizeof(double) * 2 * (size_t)No);
    const size_t in_bytes = off;
    const size_t o_centre = take(sizeof(double) * 3 * (size_t)Nc);
    const size_t out0 = off;
    const size_t o_cnt = take(sizeof(int) * 2), o_del = take((size_t)No), o_out = take((size_t)Np), o_err = take(sizeof(double) * (size_t)Np),
                 o_ang = take(sizeof(double) * (size_t)Np);
    const size_t out_bytes = off - out0;
    HostBundle hb;
    if (!g_bundles.get(0, &hb)) return XRSFM_BA_ENODEV;
    size_t cls = 0;
    unsigned char* base = (unsigned char*)g_cache.get(0, off, &cls);
    if (!base) { g_bundles.put(0, hb); return XRSFM_BA_ENOMEM; }
    int e = 0;
    {
        // (pinned staging from a recycled pool: see PinnedPool)
        struct Pin { unsigned char* p = nullptr; size_t cap = 0; unsigned char* data() const { return p; } ~Pin() { if (p) g_pinned.put(p, cap); } } stage, back;
        stage.p = (unsigned char*)g_pinned.get(in_bytes, &stage.cap); back.p = (unsigned char*)g_pinned.get(out_bytes, &back.cap);
        if (!stage.p || !back.p) { g_cache.put(0, base, cls); g_bundles.put(0, hb); return XRSFM_BA_ENOMEM; }
        auto put = [&](size_t o, const void* h, size_t bytes) { if (bytes) memcpy(stage.data() + o, h, bytes); };
        put(o_cam, cams.data(), sizeof(CamRec) * (size_t)Nc); put(o_model, model.data(), sizeof(int) * (size_t)Nc);
        put(o_P, p->points, sizeof(double) * 3 * (size_t)Np); put(o_ptr, ptr.data(), sizeof(int) * ((size_t)Np + 1));
        put(o_ocam, ocam.data(), sizeof(int) * (size_t)No); put(o_ouv, ouv.data(), sizeof(double) * 2 * (size_t)No);
        hipStream_t st = hb.stream;
        if (hipMemcpyAsync(base, stage.data(), in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) e = XRSFM_BA_ENODEV;
        if (!e && hipMemsetAsync(base + o_cnt, 0, sizeof(int) * 2, st) != hipSuccess) e = XRSFM_BA_ENODEV;
        if (!e) {
            if (Nc > 0) hipLaunchKernelGGL(k_cam_centres, dim3(cdiv(Nc, 256)), dim3(256), 0, st, (const CamRec*)(base + o_cam), Nc, (double*)(base + o_centre));
            if (Np > 0) hipLaunchKernelGGL(k_filter_tracks, dim3(cdiv(Np, 128)), dim3(128), 0, st, (const CamRec*)(base + o_cam), (const int*)(base + o_model),
                                           (const double*)(base + o_centre), (const double*)(base + o_P), (const int*)(base + o_ptr), (const int*)(base + o_ocam),
                                           (const double*)(base + o_ouv), Np, max_reproj_error, min_tri_angle_rad, (unsigned char*)(base + o_del),
                                           (unsigned char*)(base + o_out), (double*)(base + o_err), (double*)(base + o_ang), (int*)(base + o_cnt));
            if (hipGetLastError() != hipSuccess) e = XRSFM_BA_ENODEV;
        }
        if (!e && hipMemcpyAsync(back.data(), base + out0, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) e = XRSFM_BA_ENODEV;
        if (hipStreamSynchronize(st) != hipSuccess) e = XRSFM_BA_ENODEV;
        if (!e) {
            const unsigned char* o = back.data() - out0;            // (offsets above are relative to the block)
            const unsigned char* del = o + o_del;
            for (int k2 = 0; k2 < No; ++k2) obs_delete[order[k2]] = del[k2];
            if (Np) memcpy(track_outlier, o + o_out, (size_t)Np);
            if (Np && track_error) memcpy(track_error, o + o_err, sizeof(double) * (size_t)Np);
            if (Np && track_angle) memcpy(track_angle, o + o_ang, sizeof(double) * (size_t)Np);
            if (num_filtered) { int cnt[2]; memcpy(cnt, o + o_cnt, sizeof(cnt)); num_filtered[0] = cnt[0]; num_filtered[1] = cnt[1]; }
        }
    }
    g_cache.put(0, base, cls);
    g_bundles.put(0, hb);
    return e;
}

int xrsfm_ba_filter_tracks(const xrsfm_ba_problem* p, double max_reproj_error, double min_tri_angle_rad, uint8_t* obs_delete,
                           uint8_t* track_outlier, double* track_error, double* track_angle, int32_t* num_filtered) {
    return no_throw([&] { return filter_tracks_impl(p, max_reproj_error, min_tri_angle_rad, obs_delete, track_outlier, track_error, track_angle, num_filtered); });
}

// ---------------------------------------------------------------- diagnostics
int xrsfm_ba_debug_linearize(xrsfm_ba_context* c, double huber_a, int use_scaling, double* r, double* Jc, double* Jp,
                             double* Hpp, double* gp, double* Hcc_diag, double* gc, double* cost) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || c->wide) return XRSFM_BA_EINVAL;          // (bal9 contexts: xrsfm_ba_debug_wide)
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    int e;
    if ((e = ensure_host_pack(c, 2))) return e;
    if ((e = init_scaling_and_linearize(c, huber_a, use_scaling != 0))) return e;
    if ((e = fetch_scalars(c))) return e;
    if (cost) *cost = 0.5 * c->h_scal[S_COST];
    const Packed& k = c->pk;
    const size_t ns = (size_t)k.n_slots;
    auto fetch = [&](const double* dev, size_t n, std::vector<double>& h) -> int {
        h.resize(n);
        if (n && hipMemcpy(h.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
        return 0;
    };
    std::vector<double> h;
    if (r) {
        if ((e = fetch(d.rt, ns * 2, h))) return e;
        for (size_t s = 0; s < ns; ++s) if (k.slot_obs[s] >= 0) { r[2 * (size_t)k.slot_obs[s]] = h[s]; r[2 * (size_t)k.slot_obs[s] + 1] = h[ns + s]; }
    }
    if (Jc || Jp) {
        double *dF = nullptr, *dE = nullptr;
        if (hipMalloc((void**)&dF, (ns ? ns : 1) * 12 * sizeof(double)) != hipSuccess) return XRSFM_BA_ENOMEM;
        if (hipMalloc((void**)&dE, (ns ? ns : 1) * 6 * sizeof(double)) != hipSuccess) { (void)hipFree(dF); return XRSFM_BA_ENOMEM; }
        if (ns) hipLaunchKernelGGL(k_debug_materialize, dim3(cdiv((long long)ns, kBlock)), dim3(kBlock), 0, c->stream, d, dF, dE);
        (void)hipStreamSynchronize(c->stream);
        std::vector<double> hF, hE;
        e = fetch(dF, ns * 12, hF);
        if (!e) e = fetch(dE, ns * 6, hE);
        (void)hipFree(dF); (void)hipFree(dE);
        if (e) return e;
        for (size_t s2 = 0; s2 < ns; ++s2) {
            if (k.slot_obs[s2] < 0) continue;
            if (Jc) for (int q = 0; q < 12; ++q) Jc[12 * (size_t)k.slot_obs[s2] + q] = hF[q * ns + s2];
            if (Jp) for (int q = 0; q < 6; ++q) Jp[6 * (size_t)k.slot_obs[s2] + q] = hE[q * ns + s2];
        }
    }
    if (Hpp) {
        if ((e = fetch(d.Hpp, (size_t)k.n_pts * 6, h))) return e;
        for (int j = 0; j < c->n_points_caller; ++j) for (int q = 0; q < 6; ++q) Hpp[6 * (size_t)j + q] = 0.0;
        for (int j = 0; j < k.n_pts; ++j) for (int q = 0; q < 6; ++q) Hpp[6 * (size_t)k.pt_orig[j] + q] = h[6 * (size_t)j + q];
    }
    if (gp) {
        if ((e = fetch(d.gp, (size_t)k.n_pts * 3, h))) return e;
        for (int j = 0; j < c->n_points_caller; ++j) for (int q = 0; q < 3; ++q) gp[3 * (size_t)j + q] = 0.0;
        for (int j = 0; j < k.n_pts; ++j) for (int q = 0; q < 3; ++q) gp[3 * (size_t)k.pt_orig[j] + q] = h[3 * (size_t)j + q];
    }
    if (Hcc_diag || gc) {
        if ((e = fetch(d.camlin, (size_t)k.n_cams * 12, h))) return e;
        for (int i = 0; i < k.n_cams; ++i) for (int q = 0; q < 6; ++q) {
            if (Hcc_diag) Hcc_diag[6 * (size_t)i + q] = h[12 * (size_t)i + q];
            if (gc) gc[6 * (size_t)i + q] = h[12 * (size_t)i + 6 + q];
        }
    }
    return 0;
}

// bal9 mode diagnostics: linearise at the current state (Jacobi scaling from the column norms), optionally assemble and solve the
// reduced system for one radius.  Outputs in caller order: cost, per observation r [n_obs][2], Jc [n_obs][2][9], Jp [n_obs][2][3];
// per camera diag(Hcc) and g_c [n_cams][9]; step y [n_cams][9] (skipped if NULL).
int xrsfm_ba_debug_wide(xrsfm_ba_context* c, double huber_a, double radius, double* cost, double* r, double* Jc, double* Jp,
                        double* Hcc_diag, double* gc, double* y) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || !c->wide) return XRSFM_BA_EINVAL;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    CholHost& h = c->chol;
    int e;
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_cams * kW, kBlock) + 1), dim3(kBlock), 0, c->w.scale_c, 1.0, (size_t)d.n_cams * kW);
    LAUNCH(c, K_SMALL, k_fill, dim3(cdiv((long long)d.n_pts * 3, kBlock) + 1), dim3(kBlock), 0, d.scale_p, 1.0, (size_t)d.n_pts * 3);
    if ((e = linearize_wide(c, huber_a, false))) return e;
    {
        const long long n = std::max((long long)d.n_cams * kW, (long long)d.n_pts * 3);
        LAUNCH(c, K_SMALL, k9_scale_from_norms, dim3(cdiv(n, kBlock) + 1), dim3(kBlock), 0, d, c->w);
    }
    if ((e = linearize_wide(c, huber_a, true))) return e;
    if ((e = fetch_scalars(c))) return e;
    c->linearized = true;
    if (cost) *cost = 0.5 * c->h_scal[S_COST];
    const Packed& k = c->pk;
    const size_t ns = (size_t)k.n_slots;
    std::vector<double> hb;
    auto fetch = [&](const double* dev, size_t n) -> int {
        hb.resize(n);
        if (n && hipMemcpy(hb.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
        return 0;
    };
    if (r) {
        if ((e = fetch(d.rt, ns * 2))) return e;
        for (size_t s2 = 0; s2 < ns; ++s2) if (k.slot_obs[s2] >= 0) { r[2 * (size_t)k.slot_obs[s2]] = hb[s2]; r[2 * (size_t)k.slot_obs[s2] + 1] = hb[ns + s2]; }
    }
    if (Jc) {
        if ((e = fetch(c->w.Fw, ns * 18))) return e;
        for (size_t s2 = 0; s2 < ns; ++s2) if (k.slot_obs[s2] >= 0) for (int q = 0; q < 18; ++q) Jc[18 * (size_t)k.slot_obs[s2] + q] = hb[q * ns + s2];
    }
    if (Jp) {
        if ((e = fetch(c->w.Ew, ns * 6))) return e;
        for (size_t s2 = 0; s2 < ns; ++s2) if (k.slot_obs[s2] >= 0) for (int q = 0; q < 6; ++q) Jp[6 * (size_t)k.slot_obs[s2] + q] = hb[q * ns + s2];
    }
    if (Hcc_diag || gc) {
        if ((e = fetch(c->w.camlin, (size_t)k.n_cams * 18))) return e;
        for (int i = 0; i < k.n_cams; ++i) for (int q = 0; q < kW; ++q) {
            if (Hcc_diag) Hcc_diag[kW * (size_t)i + q] = hb[18 * (size_t)i + q];
            if (gc) gc[kW * (size_t)i + q] = hb[18 * (size_t)i + kW + q];
        }
    }
    if (y) {
        if ((e = chol_setup(c))) return e == kErrDuplicateObs ? XRSFM_BA_EINVAL : e;
        if ((e = assemble_wide(c, radius))) return e;
        if ((e = chol_factor_solve(c))) return e;
        HIPCHK(hipMemcpyAsync(y, c->w.px, sizeof(double) * (size_t)k.n_cams * kW, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// bal9 mode: the intrinsics {f, k1, k2} of the cameras that keep them variable, written into intr_params [n_intr][8] (rows of
// the other entries untouched) — what xrsfm_ba_solve does for the caller's problem->intr_params.
int xrsfm_ba_download_intrinsics(xrsfm_ba_context* c, double* intr_params) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || !intr_params) return XRSFM_BA_EINVAL;
    if (!c->wide) return XRSFM_BA_OK;
    HIPCHK(hipSetDevice(c->device));
    const int n = c->d.n_cams;
    std::vector<CamRec> cams(n);
    std::vector<unsigned char> cc(n);
    if (n) { HIPCHK(hipMemcpy(cams.data(), c->d.cam, sizeof(CamRec) * (size_t)n, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(cc.data(), c->d.cam_const, n, hipMemcpyDeviceToHost)); }
    for (int i = 0; i < n; ++i)
        if (cc[i] & kCamIntrVariable) for (int q = 0; q < 3; ++q) intr_params[8 * (size_t)c->cam_intr_host[i] + q] = cams[i].intr[q];
    return XRSFM_BA_OK;
}

int xrsfm_ba_debug_schur_product(xrsfm_ba_context* c, double radius, const double* x, double* y, double* b) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (c && c->wide) return XRSFM_BA_EINVAL;
    if (!c || !x || !y) return XRSFM_BA_EINVAL;
    if (!c->linearized) return XRSFM_BA_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    int e;
    if ((e = prepare_step(c, radius))) return e;
    HIPCHK(hipMemsetAsync(d.st, 0, sizeof(PcgStatus), c->stream));
    const size_t n = (size_t)d.n_cams * 6;
    HIPCHK(hipMemcpyAsync(d.pp, x, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if ((e = schur_product(c, d.pp, d.pq, 0))) return e;
    std::vector<double> q(n), dc(n);
    HIPCHK(hipMemcpyAsync(q.data(), d.pq, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(dc.data(), d.Dc2, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (b) HIPCHK(hipMemcpyAsync(b, d.b, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) y[i] = q[i] + dc[i] * x[i];
    return 0;
}

static bool problem_is_wide(const xrsfm_ba_problem* p) {
    for (int i = 0; i < p->n_cams && p->cam_const; ++i) if (p->cam_const[i] & kCamIntrVariable) return true;
    return false;
}

int xrsfm_ba_debug_pack(const xrsfm_ba_problem* p, int32_t stats[8], int32_t* slot_obs) {
    if (!p || !stats) return XRSFM_BA_EINVAL;
    Packed k;
    const int e = pack_problem(*p, k, problem_is_wide(p));
    if (e) return e;
    int regular = 0, longs = 0, maxlen = 0;
    for (int t = 0; t < k.n_tiles; ++t) { regular += k.tile_stride[t] > 0; maxlen = std::max(maxlen, k.tile_maxlen[t]); }
    for (size_t i = 0; i + 1 < k.items.size(); i += 2)
        if (k.items[i + 1] > 1) { ++longs; int len = 0; for (int s2 = 64 * k.items[i]; s2 < 64 * (k.items[i] + k.items[i + 1]); ++s2) len += k.slot_cam[s2] >= 0; maxlen = std::max(maxlen, len); }
    stats[0] = k.n_tiles; stats[1] = k.n_slots; stats[2] = (int32_t)k.items.size() / 2; stats[3] = regular; stats[4] = longs;
    stats[5] = k.n_cam_entries; stats[6] = maxlen; stats[7] = k.n_pts;
    if (slot_obs) for (int s2 = 0; s2 < k.n_slots; ++s2) slot_obs[s2] = k.slot_obs[s2];
    return 0;
}

int xrsfm_ba_debug_gram_schedule(int n_cams, int32_t* n_inst, int32_t* n_inst_all, uint16_t* entries) {
    if (n_cams < 1 || n_cams > kGramMaxCams || !n_inst || !n_inst_all || !entries) return XRSFM_BA_EINVAL;
    static const GramSched4<6> S = make_gram_sched4<6>();        // (the table the device copy g_gram_sched6 is initialised from)
    *n_inst_all = S.n[n_cams];
    *n_inst = S.n[n_cams] <= kGram4MaxInst ? S.n[n_cams] : 0;
    for (int i = 0; i < 4 * S.n[n_cams] && i < 128; ++i) entries[i] = S.e[n_cams][i];
    return 0;
}

int xrsfm_ba_debug_pack_gram(const xrsfm_ba_problem* p, int32_t stats[8], int32_t* tile_ncam, uint8_t* slot_cidx, int32_t* slot_campos_g) {
    if (!p || !stats) return XRSFM_BA_EINVAL;
    Packed k;
    int e = pack_problem(*p, k, problem_is_wide(p));
    if (e) return e;
    std::vector<int> spp;
    PairKeys keyed;
    if ((e = chol_local_keys(k, spp, keyed))) return e == kErrDuplicateObs ? XRSFM_BA_EINVAL : e;
    CholPlan P;
    if ((e = chol_plan_build(k, spp, keyed, nullptr, P, kCholMaxN, kCholMaxBytes))) return e;
    int n_gram = 0, cmax = 0;
    for (int t = 0; t < k.n_tiles; ++t) { n_gram += k.tile_ncam[t] > 0; cmax = std::max(cmax, k.tile_ncam[t]); }
    stats[0] = n_gram; stats[1] = k.n_gt_cells; stats[2] = k.n_cam_entries_g; stats[3] = cmax;
    stats[4] = P.n_pairs_small; stats[5] = P.n_pairs_big; stats[6] = P.n_pairs_other; stats[7] = P.n_writes;
    if (tile_ncam) for (int t = 0; t < k.n_tiles; ++t) tile_ncam[t] = k.tile_ncam[t];
    if (slot_cidx) for (int s2 = 0; s2 < k.n_slots; ++s2) slot_cidx[s2] = k.slot_cidx[s2];
    if (slot_campos_g) for (int s2 = 0; s2 < k.n_slots; ++s2) slot_campos_g[s2] = k.slot_campos_g[s2];
    return 0;
}

int xrsfm_ba_debug_chol_plan(const xrsfm_ba_problem* p, int32_t stats[8], int32_t* cam_offset) {
    if (!p || !stats) return XRSFM_BA_EINVAL;
    Packed k;
    const bool wide = problem_is_wide(p);
    int e = pack_problem(*p, k, wide);
    if (e) return e;
    std::vector<int> spp;
    PairKeys keyed;
    if ((e = chol_local_keys(k, spp, keyed))) return e == kErrDuplicateObs ? XRSFM_BA_EINVAL : e;
    CholPlan P;
    if ((e = chol_plan_build(k, spp, keyed, nullptr, P, kCholMaxN, kCholMaxBytes, wide ? kW : 6))) return e == kErrPlanCheck ? XRSFM_BA_EINTERNAL : e;
    stats[0] = P.T; stats[1] = P.n_levels; stats[2] = P.ordering; stats[3] = P.n_hubs; stats[4] = P.band; stats[5] = P.n_blocks;
    stats[6] = (P.use_levels ? 1 : 0) | (P.lookahead ? 2 : 0); stats[7] = P.n_tiles_nz;
    if (cam_offset) for (int i = 0; i < k.n_cams; ++i) cam_offset[i] = P.cam_off[i];
    return 0;
}

int xrsfm_ba_debug_set_block_pattern(xrsfm_ba_context* c, int n_pairs, const int32_t* row_col) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || n_pairs < 0 || (n_pairs > 0 && !row_col)) return XRSFM_BA_EINVAL;
    if (c->chol.ready) return XRSFM_BA_ESTATE;
    c->pattern_keys.clear();
    for (int i = 0; i < n_pairs; ++i) {
        const int rb = row_col[2 * i], ca = row_col[2 * i + 1];
        if (ca < 0 || rb <= ca || rb >= c->d.n_cams) return XRSFM_BA_EINVAL;
        c->pattern_keys.push_back(((unsigned long long)rb << 32) | (unsigned)ca);
    }
    std::sort(c->pattern_keys.begin(), c->pattern_keys.end());
    c->pattern_keys.erase(std::unique(c->pattern_keys.begin(), c->pattern_keys.end()), c->pattern_keys.end());
    c->have_pattern = true;
    return 0;
}

int xrsfm_ba_debug_cholesky_solve(xrsfm_ba_context* c, double radius, double* y, double* S_dense) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || !y || c->wide) return XRSFM_BA_EINVAL;
    if (!c->linearized) return XRSFM_BA_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    int e;
    if ((e = chol_setup(c))) return e == kErrDuplicateObs ? XRSFM_BA_EINVAL : e;
    if ((e = prepare_step(c, radius, true))) return e;
    if ((e = chol_assemble(c, S_dense != nullptr))) return e;      // (the fused fill of the first level is what a plain call runs)
    const CholDev& cd = c->chol.dev;
    if (S_dense) {
        std::vector<double> h(c->chol.S_doubles);
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipMemcpy(h.data(), cd.S, h.size() * sizeof(double), hipMemcpyDeviceToHost));
        const std::vector<int>& off = c->chol.cam_off_host;
        const std::vector<int>& tm = c->chol.tile_map_host;
        auto at = [&](int r, int col) -> double {      // element (r, col), r >= col, of the lower triangle in either storage form
            if (!cd.tmap) return h[(size_t)r * cd.n_pad + col];
            return h[(size_t)tm[(size_t)(r / kNB) * cd.T + col / kNB] * cd.tstride + (size_t)(r % kNB) * kNB + col % kNB];
        };
        const int Nc = d.n_cams;
        for (int ca = 0; ca < Nc; ++ca)
            for (int cb = 0; cb < Nc; ++cb)
                for (int a = 0; a < 6; ++a)
                    for (int b2 = 0; b2 < 6; ++b2) {
                        const int r = off[ca] + a, col = off[cb] + b2;
                        S_dense[(size_t)(6 * ca + a) * cd.n + 6 * cb + b2] = (r >= col) ? at(r, col) : at(col, r);
                    }
    }
    if ((e = chol_factor_solve(c))) return e;
    c->step_valid = true;
    HIPCHK(hipMemcpyAsync(y, d.px, sizeof(double) * (size_t)cd.n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int xrsfm_ba_debug_backsub(xrsfm_ba_context* c, double* part_model, double* part_step2, double* cand_points, double* point_step,
                           double* cand_cam_q, double* cand_cam_t) {
    if (c && c->poisoned) return XRSFM_BA_ESTATE;       // the watchdog gave up on this context's stream: nothing may wait for it again
    if (!c || c->wide) return XRSFM_BA_EINVAL;
    // needs the camera part of a step (d.px) and the radius / point factors it was assembled with: without a preceding
    // xrsfm_ba_debug_cholesky_solve of the SAME linearisation the kernel would read uninitialised Hinv / px (or divide by a zero radius)
    if (!c->linearized || !c->chol.ready || !c->step_valid) return XRSFM_BA_ESTATE;
    HIPCHK(hipSetDevice(c->device));
    Dev& d = c->d;
    const int nbi = cdiv(d.n_items, kWavesPerBlock), nbc = cdiv(d.n_cams, kBlock);
    if (nbi + nbc > 0) {
        if (c->step_prep) hipLaunchKernelGGL(k_backsub<true>, dim3(nbi + nbc), dim3(kBlock), 0, c->stream, d, nbi, (CamLin*)nullptr, c->step_radius);
        else hipLaunchKernelGGL(k_backsub<false>, dim3(nbi + nbc), dim3(kBlock), 0, c->stream, d, nbi, (CamLin*)nullptr, c->step_radius);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    const size_t ni = (size_t)d.n_items, np = (size_t)d.n_pts;
    if (part_model && ni) HIPCHK(hipMemcpy(part_model, d.part + 2 * ni, ni * sizeof(double), hipMemcpyDeviceToHost));
    if (part_step2 && ni) HIPCHK(hipMemcpy(part_step2, d.part + 3 * ni, ni * sizeof(double), hipMemcpyDeviceToHost));
    if (cand_points && np) HIPCHK(hipMemcpy(cand_points, d.P_cand, 3 * np * sizeof(double), hipMemcpyDeviceToHost));
    if (point_step && np) HIPCHK(hipMemcpy(point_step, d.yp, 3 * np * sizeof(double), hipMemcpyDeviceToHost));
    if ((cand_cam_q || cand_cam_t) && d.n_cams) {
        std::vector<CamRec> cams(d.n_cams);
        HIPCHK(hipMemcpy(cams.data(), d.cam_cand, sizeof(CamRec) * (size_t)d.n_cams, hipMemcpyDeviceToHost));
        for (int i = 0; i < d.n_cams; ++i) {
            if (cand_cam_q) for (int j = 0; j < 4; ++j) cand_cam_q[4 * (size_t)i + j] = cams[i].q[j];
            if (cand_cam_t) for (int j = 0; j < 3; ++j) cand_cam_t[3 * (size_t)i + j] = cams[i].t[j];
        }
    }
    return 0;
}

// Device-side packing against the host's (ba_pack_dev.h / ba_pack.h): packs `p` both ways and compares every array.
// *field = 0 and return 0 when they are identical; otherwise *field names the first array that differs (1 counts, 2 items,
// 3 pt_orig, 4 slot_cam, 5 slot_pt, 6 slot_obs, 7 slot_u/v, 8 tile_maxlen, 9 tile_stride, 10 tile_ncam, 11 tile_gt_off, 12 slot_cidx,
// 13 gt_cell, 14 slot_campos, 15 slot_campos_g, 16 cam_ptr, 17 cam_ptr_g, 18 pt_const, 19 points) and *index the first element;
// -100: the device path declined the problem (a track longer than 64 observations, bal9 mode, ...).
int xrsfm_ba_debug_device_pack_check(const xrsfm_ba_problem* p, int32_t* field, int32_t* index) {
    if (!p || !field || !index) return XRSFM_BA_EINVAL;
    *field = 0; *index = -1;
    return no_throw([&]() -> int {
        Packed h;
        int e = pack_problem(*p, h, problem_is_wide(p));
        if (e) return e;
        xrsfm_ba_context* c = nullptr;
        g_force_device_pack = 1;
        e = xrsfm_ba_create(p, 0, &c);
        g_force_device_pack = -1;
        if (e) return e;
        struct Guard { xrsfm_ba_context* c; ~Guard() { xrsfm_ba_destroy(c); } } guard{c};
        if (!c->dev_packed) { *field = -100; return 0; }
        if ((e = ensure_host_pack(c, 2))) return e;
        const Packed& k = c->pk;
        auto diff = [&](int f, long long i) { *field = f; *index = (int32_t)i; return 0; };
        if (k.n_cams != h.n_cams || k.n_pts != h.n_pts || k.n_obs != h.n_obs || k.n_tiles != h.n_tiles || k.n_slots != h.n_slots ||
            k.n_gt_cells != h.n_gt_cells || k.n_cam_entries != h.n_cam_entries || k.n_cam_entries_g != h.n_cam_entries_g ||
            k.n_var_q != h.n_var_q || k.n_var_t != h.n_var_t || k.n_var_p != h.n_var_p) return diff(1, 0);
        auto cmp = [&](int f, const auto& a, const auto& b) -> bool {
            if (a.size() != b.size()) { diff(f, -2); return false; }
            for (size_t i = 0; i < a.size(); ++i) if (a[i] != b[i]) { diff(f, (long long)i); return false; }
            return true;
        };
        if (!cmp(2, k.items, h.items) || !cmp(3, k.pt_orig, h.pt_orig) || !cmp(4, k.slot_cam, h.slot_cam) || !cmp(5, k.slot_pt, h.slot_pt) ||
            !cmp(6, k.slot_obs, h.slot_obs)) return 0;
        {
            const size_t ns = (size_t)k.n_slots;
            std::vector<double> u(ns), v(ns);
            if (ns && (hipMemcpy(u.data(), c->d.slot_u, ns * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
                       hipMemcpy(v.data(), c->d.slot_v, ns * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) return XRSFM_BA_ENODEV;
            for (size_t i = 0; i < ns; ++i) if (u[i] != h.slot_u[i] || v[i] != h.slot_v[i]) return diff(7, (long long)i);
        }
        if (!cmp(8, k.tile_maxlen, h.tile_maxlen) || !cmp(9, k.tile_stride, h.tile_stride) || !cmp(10, k.tile_ncam, h.tile_ncam) ||
            !cmp(11, k.tile_gt_off, h.tile_gt_off) || !cmp(12, k.slot_cidx, h.slot_cidx) || !cmp(13, k.gt_cell, h.gt_cell) ||
            !cmp(14, k.slot_campos, h.slot_campos) || !cmp(15, k.slot_campos_g, h.slot_campos_g) || !cmp(16, k.cam_ptr, h.cam_ptr) ||
            !cmp(17, k.cam_ptr_g, h.cam_ptr_g) || !cmp(18, k.pt_const, h.pt_const)) return 0;
        {
            const size_t np3 = 3 * (size_t)k.n_pts;
            std::vector<double> P(np3);
            if (np3 && hipMemcpy(P.data(), c->d.P, np3 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return XRSFM_BA_ENODEV;
            for (size_t j = 0; j < (size_t)k.n_pts; ++j)
                for (int a = 0; a < 3; ++a) if (P[3 * j + a] != p->points[3 * (size_t)h.pt_orig[j] + a]) return diff(19, (long long)j);
        }
        return 0;
    });
}

#ifdef XBA_TIMELINE
int xrsfm_ba_debug_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(xba::g_stamps), sizeof(unsigned long long) * 3 * 64 * 16) == hipSuccess ? 0 : XRSFM_BA_ENODEV;
}
#endif

}  // extern "C"

// ---------------------------------------------------------------- exception barrier of the entry points that allocate on the host
int xrsfm_ba_run(xrsfm_ba_context* c, const xrsfm_ba_options* optp, xrsfm_ba_summary* sum) { return no_throw([&] { return ba_run_impl(c, optp, sum); }); }
int xrsfm_pg_solve(const xrsfm_pg_options* opt, xrsfm_pg_problem* p, xrsfm_pg_summary* summary) { return no_throw([&] { return pg_solve_impl(opt, p, summary); }); }
int xrsfm_tag_refine(const xrsfm_pg_options* opt, xrsfm_tag_problem* p, int32_t stages, xrsfm_pg_summary* summaries) { return no_throw([&] { return tag_refine_impl(opt, p, stages, summaries); }); }
