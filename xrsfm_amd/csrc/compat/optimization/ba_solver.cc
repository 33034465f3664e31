// xrsfm::BASolver::{GBA,KGBA,LBA} on top of the C-ABI (include/xrsfm_ba.h).
//
// Behaviour restated from /root/reference/src/optimization/ba_solver.cc (no code shared with it):
//   which frames enter           GBA :598-607 (registered), KGBA :647-660 (registered key frames), LBA :525-549
//   residual blocks per frame    SetUp :330-356 / SetUpLBA :358-391 (track_ids_[i] != -1)
//   constant blocks              intrinsics always (:602-606); gauge t of init_id1/2 (:611-614, :662-663);
//                                fix_all_frames (:616-621); LBA points not seen by the new frame (:380-382);
//                                LBA gauge fallbacks (:551-584)
//   solver options               :70-77 with :586-589 (LBA), :626-634 (GBA), :667-670 (KGBA)
//   printed summary              PrintSolverSummary :14-68, "LBA:" line :537-549, "kf: a/b" :676
//   KGBA pre/post                KeyFrameSelection / UpdateByRefFrame stay in the reference (src/base/map.cc:428-663); builds
//                                without src/base link the restatement in ../base/map_ops.cc instead
//   pose graph                   ScalePoseGraphUnorder :147-328 with AddCovisibilityEdge :79-115, AddLoopEdge :117-145
#include "ba_solver.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iomanip>
#include <iostream>
#include <memory>
#include <set>
#include <thread>
#include <unordered_map>

#include "geometry/colmap/base/triangulation.h"
#include "geometry/colmap/util/math.h"
#include "xrsfm_ba.h"

namespace xrsfm {
int &BASolverFailureCount() { static int n = 0; return n; }

namespace {

}  // namespace

// (round 6) The observation side of the LAST large call (global BA), kept by the BASolver between calls: which tracks take part,
// their slots, obs_cam / obs_pt / obs_uv.  A second GBA over the same frames with the same track_ids_ — the mapper's accurate GBA
// after a KGBA, a final GBA pass, a re-run after the filters removed nothing — finds it by KEY and only refreshes poses and
// points: Map -> SoA 10-12 ms -> ~3 ms at BASELINE config 4's size.  The key is cheap and
// conservative: per frame of the call, in order, an FNV-1a hash over its id, camera id, every entry of track_ids_ and the
// address + size of frame.points (the key points of a frame are written once when it is loaded, map.h:29-64: their VALUES are
// not hashed); plus the size of Map::tracks_ and the LBA flag.  Any difference rebuilds.  Memory: 24 bytes per observation + 8
// per track of the map while the BASolver lives (48 MB at 2 M observations).
struct BASolverObsCache {
    uint64_t key = 0; bool valid = false;
    size_t n_frames = 0, total_feats = 0, n_map_tracks = 0, n_obs = 0;
    std::vector<int> tracks, track_slot;
    std::unique_ptr<int32_t[]> obs_cam, obs_pt;
    std::unique_ptr<double[]> obs_uv;
};

namespace {

// Flat copy of the parameter blocks of one BA call + the way back into the Map.
class FlatProblem {
  public:
    // (round 4) Tracks and frames are addressed by their index in Map::tracks_ / Map::frames_ (map.h:116-195).  AddFrame only records
    // the frame; the observations are laid out in one go (BuildObservations, at the start of Solve): per frame the count of tracked
    // features, a mark per track that occurs, slots = rank of the track id among the marked ones, then every frame fills its own
    // range of the observation arrays — each of these passes runs over the frames in parallel for a large call.  (Rounds 1-3: a
    // hash lookup and a cache-missing Track access per observation, 52 ms of a global BA of 2 M observations on one thread.)
    explicit FlatProblem(Map &map, BASolverObsCache *cache = nullptr) : map_(map), cache_(cache), t_begin_(std::chrono::steady_clock::now()) {}
    ~FlatProblem() { StoreCache(); }

    // One frame = SetUp(problem, map, frame).  `lba_frame_id >= 0` selects SetUpLBA's rule for constant points.
    void AddFrame(Frame &frame, int lba_frame_id = -1) {
        const int cam = static_cast<int>(frames_.size());
        const Camera &camera = map_.Camera(frame.camera_id);
        lba_frame_id_ = lba_frame_id;
        frames_.push_back(&frame);
        frame_slot_[static_cast<int>(frame.id)] = cam;
        const double *q = frame.Tcw.q.coeffs().data();   // x,y,z,w
        const double *t = frame.Tcw.t.data();
        cam_q_.insert(cam_q_.end(), q, q + 4);
        cam_t_.insert(cam_t_.end(), t, t + 3);
        cam_const_.push_back(0);
        // intrinsics: one constant block per camera_id
        auto ci = intr_slot_.find(static_cast<int>(frame.camera_id));
        if (ci == intr_slot_.end()) {
            ci = intr_slot_.emplace(static_cast<int>(frame.camera_id), static_cast<int>(intr_model_.size())).first;
            intr_model_.push_back(static_cast<int>(camera.model_id_));
            for (size_t k = 0; k < 8; ++k) intr_params_.push_back(k < camera.params_.size() ? camera.params_[k] : 0.0);
        }
        cam_intr_.push_back(ci->second);
    }

    bool HasFrame(int frame_id) const { return frame_slot_.count(frame_id) != 0; }
    void FixTranslation(int frame_id) {
        auto it = frame_slot_.find(frame_id);
        if (it != frame_slot_.end()) cam_const_[it->second] |= XRSFM_BA_CONST_T;
    }
    void FixPose(int frame_id) {
        auto it = frame_slot_.find(frame_id);
        if (it != frame_slot_.end()) cam_const_[it->second] |= (XRSFM_BA_CONST_Q | XRSFM_BA_CONST_T);
    }
    size_t NumFrames() const { return frames_.size(); }

    // run fn(begin, end) over [0, n) on up to 16 threads (large calls only); an exception of a worker is rethrown on the caller's thread
    template <typename F> static void ParallelFor(size_t n, size_t min_n, F &&fn) {
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = (n < min_n || hw < 2) ? 1 : std::min<size_t>(16, hw);
        if (nt == 1) { fn(0, n); return; }
        std::vector<std::thread> th;
        std::vector<std::exception_ptr> err(nt);
        for (size_t t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                try { fn(n * t / nt, n * (t + 1) / nt); } catch (...) { err[t] = std::current_exception(); }
            });
        for (auto &x : th) x.join();
        for (auto &e : err) if (e) std::rethrow_exception(e);
    }
    // Observations of the recorded frames -> obs_cam / obs_pt / obs_uv, tracks_ (ascending track id), point_const_.
    void BuildObservations() {
        static const bool trace = std::getenv("XRSFM_BA_TRACE_CALLS") != nullptr;
        auto tp = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!trace) return;
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[BASolver adapter]   %-28s %.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
            tp = now;
        };
        const size_t nf = frames_.size();
        std::vector<size_t> off(nf + 1, 0);
        size_t total_feats = 0;
        for (size_t c = 0; c < nf; ++c) total_feats += frames_[c]->track_ids_.size();
        // A large call (global BA) marks its tracks in a table over the whole map and the frames run in parallel; a small one (LBA: a
        // few thousand observations, once per registered frame) must not pay for the size of the MAP — 1-2 M tracks are 10 MB of
        // memset and a 2 M-iteration scan per call, as long as the solve itself —: it sorts the track ids it meets (round 5).
        // (the table over the map costs ~2 ns per track of the MAP, the sort ~50 ns per observation of the CALL: the table wherever the map
        //  is at most 16 x the call — mapper replay, 45 000 tracks: LBA 0.84 ms with the table, 1.04 ms sorted)
        dense_slots_ = total_feats >= 200000 || map_.tracks_.size() <= 16 * total_feats;
        const size_t par_min = total_feats >= 200000 ? 1 : (size_t)-1;       // frames in parallel only for a large call
        cache_key_ = 0; cache_total_feats_ = total_feats;
        if (cache_ && total_feats >= 200000 && lba_frame_id_ < 0) {
            // the key of this call (see BASolverObsCache): one hash per frame, in parallel, folded in frame order
            std::vector<uint64_t> fh(nf);
            ParallelFor(nf, 1, [&](size_t c0, size_t c1) {
                for (size_t c = c0; c < c1; ++c) {
                    const Frame &f = *frames_[c];
                    uint64_t h = 1469598103934665603ull;
                    auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
                    mix(static_cast<uint64_t>(f.id)); mix(static_cast<uint64_t>(f.camera_id)); mix(f.track_ids_.size());
                    mix(reinterpret_cast<uintptr_t>(f.points.data())); mix(f.points.size());
                    const int *ids = f.track_ids_.data();
                    const size_t n = f.track_ids_.size();
                    size_t i = 0;
                    for (; i + 2 <= n; i += 2) mix((static_cast<uint64_t>(static_cast<uint32_t>(ids[i])) << 32) | static_cast<uint32_t>(ids[i + 1]));
                    for (; i < n; ++i) mix(static_cast<uint32_t>(ids[i]));
                    fh[c] = h;
                }
            });
            uint64_t key = 1469598103934665603ull;
            for (size_t c = 0; c < nf; ++c) key = (key ^ fh[c]) * 1099511628211ull;
            key = (key ^ map_.tracks_.size()) * 1099511628211ull;
            cache_key_ = key ? key : 1;
            lap("cache key");
            if (cache_->valid && cache_->key == cache_key_ && cache_->n_frames == nf && cache_->total_feats == total_feats &&
                cache_->n_map_tracks == map_.tracks_.size()) {
                tracks_.swap(cache_->tracks); track_slot_.swap(cache_->track_slot);
                obs_cam_ = std::move(cache_->obs_cam); obs_pt_ = std::move(cache_->obs_pt); obs_uv_ = std::move(cache_->obs_uv);
                n_obs_ = cache_->n_obs;
                cache_->valid = false;               // (the arrays are ours until StoreCache hands them back)
                point_const_.assign(tracks_.size(), 0);
                cache_hit_ = true;
                lap("observations from the cache");
                return;
            }
        }
        tracks_.clear();
        if (dense_slots_) {
            // (relaxed atomic marks: several frames mark the same track from different threads, all with the same value)
            std::unique_ptr<std::atomic<unsigned char>[]> used(new std::atomic<unsigned char>[map_.tracks_.size() ? map_.tracks_.size() : 1]);
            static_assert(sizeof(std::atomic<unsigned char>) == 1, "the mark table is cleared as bytes");
            ParallelFor(map_.tracks_.size(), par_min, [&](size_t a, size_t b) { std::memset(static_cast<void *>(used.get() + a), 0, b - a); });
            ParallelFor(nf, par_min, [&](size_t c0, size_t c1) {
                for (size_t c = c0; c < c1; ++c) {
                    size_t n = 0;
                    for (const int tid : frames_[c]->track_ids_)
                        if (tid != -1) { ++n; if (!used[tid].load(std::memory_order_relaxed)) used[tid].store(1, std::memory_order_relaxed); }      // (test first: 16 threads storing into shared lines cost 22 ms)
                    off[c + 1] = n;
                }
            });
            lap("count + mark tracks");
            track_slot_.assign(map_.tracks_.size(), -1);
            for (size_t tid = 0; tid < map_.tracks_.size(); ++tid)
                if (used[tid].load(std::memory_order_relaxed)) { track_slot_[tid] = static_cast<int>(tracks_.size()); tracks_.push_back(static_cast<int>(tid)); }
        } else {
            tracks_.reserve(total_feats);
            for (size_t c = 0; c < nf; ++c) {
                size_t n = 0;
                for (const int tid : frames_[c]->track_ids_)
                    if (tid != -1) { ++n; tracks_.push_back(tid); }
                off[c + 1] = n;
            }
            std::sort(tracks_.begin(), tracks_.end());
            tracks_.erase(std::unique(tracks_.begin(), tracks_.end()), tracks_.end());
            track_slot_.clear();
            lap("count + sort tracks");
        }
        for (size_t c = 0; c < nf; ++c) {
            if (off[c + 1] == 0)
                std::cerr << (lba_frame_id_ >= 0 ? "LBA" : "BA") << ": NO Measurement In Frame " << frames_[c]->id << std::endl;
            off[c + 1] += off[c];
        }
        point_const_.assign(tracks_.size(), 0);
        if (lba_frame_id_ >= 0)          // reference rule: `angle_ > 5 || observations_.count(frame_id) == 0` (angle_ is in radians, so only
            for (size_t j = 0; j < tracks_.size(); ++j) {        // the second half can fire, ba_solver.cc:380)
                const Track &track = map_.tracks_[tracks_[j]];
                if (track.angle_ > 5 || track.observations_.count(lba_frame_id_) == 0) point_const_[j] = 1;
            }
        lap("track slots");
        const size_t no = off[nf];
        n_obs_ = no;            // (uninitialised storage: the pages are first touched by the threads that fill them, not zeroed by one thread first)
        obs_cam_.reset(new int32_t[no ? no : 1]); obs_pt_.reset(new int32_t[no ? no : 1]); obs_uv_.reset(new double[no ? 2 * no : 1]);
        lap("allocate observation arrays");
        ParallelFor(nf, par_min, [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; ++c) {
                const Frame &frame = *frames_[c];
                size_t o = off[c];
                for (size_t i = 0; i < frame.track_ids_.size(); ++i) {
                    const int tid = frame.track_ids_[i];
                    if (tid == -1) continue;
                    obs_cam_[o] = static_cast<int32_t>(c);
                    obs_pt_[o] = dense_slots_ ? track_slot_[tid] : static_cast<int32_t>(std::lower_bound(tracks_.begin(), tracks_.end(), tid) - tracks_.begin());
                    obs_uv_[2 * o] = frame.points[i](0); obs_uv_[2 * o + 1] = frame.points[i](1);
                    ++o;
                }
            }
        });
        lap("fill observation arrays");
    }

    // Track::point3d_ of every track of the call -> points_ (and back).  A Track is ~100 bytes of an array of structs
    // (map.h:12-27), so each of these is a cache miss: half a million of them cost 40 ms of a global BA on one thread — spread
    // over the host's threads (the solve itself takes 15 ms at that size).
    template <typename F> void ForTracks(F &&fn) { ParallelFor(tracks_.size(), 50000, fn); }
    void GatherPoints() {
        points_.reset(new double[3 * tracks_.size() + 1]);      // (uninitialised: resize() of a vector zero-fills 12 MB on one thread first, ~1 ms at config 4's size)
        ForTracks([&](size_t j0, size_t j1) {
            for (size_t j = j0; j < j1; ++j) {
                const double *p = map_.tracks_[tracks_[j]].point3d_.data();
                points_[3 * j] = p[0]; points_[3 * j + 1] = p[1]; points_[3 * j + 2] = p[2];
            }
        });
    }

    // ceres::Solve replacement; writes the result back into the Map on success.
    int Solve(const xrsfm_ba_options &opt, xrsfm_ba_summary *summary) {
        BuildObservations();
        GatherPoints();
        xrsfm_ba_problem p;
        p.n_cams = static_cast<int32_t>(frames_.size());
        p.n_points = static_cast<int32_t>(tracks_.size());
        p.n_obs = static_cast<int32_t>(n_obs_);
        p.n_intr = static_cast<int32_t>(intr_model_.size());
        p.cam_q = cam_q_.data(); p.cam_t = cam_t_.data(); p.cam_const = cam_const_.data(); p.cam_intr = cam_intr_.data();
        p.intr_model = intr_model_.data(); p.intr_params = intr_params_.data();
        p.points = points_.get(); p.point_const = point_const_.data();
        p.obs_cam = obs_cam_.get(); p.obs_pt = obs_pt_.get(); p.obs_uv = obs_uv_.get();
        const auto t_packed = std::chrono::steady_clock::now();
        const int rc = xrsfm_ba_solve(&opt, &p, summary);
        const auto t_solved = std::chrono::steady_clock::now();
        if (rc != XRSFM_BA_OK) {
            // The reference's call sites are void and never look at a status (ba_solver.cc:636-637 ignores Ceres' summary too),
            // so a persistent failure (no device, out of memory) would silently yield a reconstruction without any BA: count the
            // failures, say so every time, and stop the process if the user asked for that (XRSFM_BA_ABORT_ON_FAILURE=1).
            const int n = ++BASolverFailureCount();
            std::cerr << "xrsfm_ba_solve failed with code " << rc << " (no CPU fallback); map left unchanged — bundle adjustment call #"
                      << n << " of this process that did NOT run" << std::endl;
            const char *abort_env = std::getenv("XRSFM_BA_ABORT_ON_FAILURE");
            if (abort_env && abort_env[0] == '1') std::abort();
            return rc;
        }
        for (size_t c = 0; c < frames_.size(); ++c) {
            double *q = frames_[c]->Tcw.q.coeffs().data();
            double *t = frames_[c]->Tcw.t.data();
            for (int k = 0; k < 4; ++k) q[k] = cam_q_[4 * c + k];
            for (int k = 0; k < 3; ++k) t[k] = cam_t_[3 * c + k];
        }
        ForTracks([&](size_t j0, size_t j1) {
            for (size_t j = j0; j < j1; ++j) {
                double *p3 = map_.tracks_[tracks_[j]].point3d_.data();
                for (int k = 0; k < 3; ++k) p3[k] = points_[3 * j + k];
            }
        });
        static const bool trace = std::getenv("XRSFM_BA_TRACE_CALLS") != nullptr;      // the adapter's own share of a call (Map -> SoA and back)
        if (trace) {
            auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "[BASolver adapter] frames %zu tracks %zu obs %zu | Map -> SoA %.3f ms%s, xrsfm_ba_solve %.3f ms, SoA -> Map %.3f ms\n", frames_.size(),
                         tracks_.size(), n_obs_, ms(t_begin_, t_packed), cache_hit_ ? " (observations cached)" : "", ms(t_packed, t_solved), ms(t_solved, std::chrono::steady_clock::now()));
        }
        return rc;
    }

    // hand the observation arrays of a large call to the BASolver's cache (called by the destructor: after the solve, whatever its outcome)
    void StoreCache() {
        if (!cache_ || cache_key_ == 0 || !obs_cam_) return;
        cache_->key = cache_key_; cache_->n_frames = frames_.size(); cache_->total_feats = cache_total_feats_;
        cache_->n_map_tracks = map_.tracks_.size(); cache_->n_obs = n_obs_;
        cache_->tracks.swap(tracks_); cache_->track_slot.swap(track_slot_);
        cache_->obs_cam = std::move(obs_cam_); cache_->obs_pt = std::move(obs_pt_); cache_->obs_uv = std::move(obs_uv_);
        cache_->valid = true;
        cache_key_ = 0;
    }
    bool cache_hit() const { return cache_hit_; }

  private:
    Map &map_;
    BASolverObsCache *cache_ = nullptr;
    uint64_t cache_key_ = 0; size_t cache_total_feats_ = 0; bool cache_hit_ = false;
    std::vector<Frame *> frames_;
    std::unordered_map<int, int> frame_slot_, intr_slot_;      // (a handful of entries: frames of the call, camera ids)
    std::vector<int> track_slot_;                              // (large calls) Map::tracks_ index -> point slot of this call, -1 = not in it
    bool dense_slots_ = false;                                 // ... small calls look the slot up in the sorted tracks_ instead
    int lba_frame_id_ = -1;
    std::chrono::steady_clock::time_point t_begin_;
    std::vector<int> tracks_;
    std::vector<double> cam_q_, cam_t_, intr_params_;
    std::unique_ptr<double[]> points_;
    std::vector<uint8_t> cam_const_, point_const_;
    std::vector<int32_t> cam_intr_, intr_model_;
    std::unique_ptr<int32_t[]> obs_cam_, obs_pt_;
    std::unique_ptr<double[]> obs_uv_;
    size_t n_obs_ = 0;
};

}  // namespace

BASolver::BASolver() : obs_cache_(std::make_shared<BASolverObsCache>()) {
    // once per process, on device 0 — the device xrsfm_ba_solve runs on (the one-shot entry point has no device argument) —, without
    // size hints: the library pre-allocates nothing for this call (hinted pre-allocations are capped by the library, xrsfm_ba.hip).
    // A box without a device is reported by the first solve.
    static const int warm = xrsfm_ba_warmup(0, 0, 0, 0);
    (void)warm;
}

namespace {
xrsfm_ba_options ReferenceOptions(int max_iterations, double ftol, double ptol) {
    xrsfm_ba_options o;
    xrsfm_ba_default_options(&o);       // Ceres defaults + SPARSE_SCHUR-equivalent exact solve
    o.max_iterations = max_iterations;
    o.function_tolerance = ftol;
    o.parameter_tolerance = ptol;
    return o;
}

// Same lines as PrintSolverSummary (ba_solver.cc:14-68).
void PrintSummary(const xrsfm_ba_summary &s) {
    const char *termination = s.termination == XRSFM_BA_CONVERGENCE ? "Convergence"
                              : s.termination == XRSFM_BA_NO_CONVERGENCE ? "No convergence" : "Failure";
    const double nres = s.num_residuals > 0 ? static_cast<double>(s.num_residuals) : 1.0;
    std::cout << std::right << std::setw(16) << "Residuals : " << std::left << s.num_residuals << std::endl;
    std::cout << std::right << std::setw(16) << "Parameters : " << std::left << s.num_effective_params << std::endl;
    std::cout << std::right << std::setw(16) << "Iterations : " << std::left << s.n_successful + s.n_unsuccessful << std::endl;
    std::cout << std::right << std::setw(16) << "Time : " << std::left << s.total_time_s << " [s]" << std::endl;
    std::cout << std::right << std::setw(16) << "Initial cost : " << std::right << std::setprecision(6)
              << std::sqrt(s.initial_cost / nres) << " [px]" << std::endl;
    std::cout << std::right << std::setw(16) << "Final cost : " << std::right << std::setprecision(6)
              << std::sqrt(s.final_cost / nres) << " [px]" << std::endl;
    std::cout << std::right << std::setw(16) << "Termination : " << std::right << termination << std::endl << std::endl;
}

// Frames sharing tracks with `frame`, most covisible first.  include_self mirrors the difference between
// CovisibilityNeibors (:503-505, counts the frame itself) and FindLocalBundle (:404, skips it).
std::vector<std::pair<int, int>> CovisibleFrames(const Frame &frame, const Map &map, bool include_self, int *num_points3d) {
    std::unordered_map<int, int> shared;
    int n3d = 0;
    for (const int tid : frame.track_ids_) {
        if (tid == -1) continue;
        ++n3d;
        for (const auto &obs : map.tracks_[tid].observations_)
            if (include_self || obs.first != static_cast<int>(frame.id)) shared[obs.first] += 1;
    }
    if (num_points3d) *num_points3d = n3d;
    std::vector<std::pair<int, int>> out(shared.begin(), shared.end());
    // most covisible first; equal counts by ascending frame id.  (The reference sorts hash-table contents on the count alone,
    // :411-416 / :507-512: its order among ties is unspecified, so a fixed rule is one of the orders it can produce — and it
    // makes the selection testable against the independent numpy restatement of tests/test_adapter.py.)
    std::sort(out.begin(), out.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) {
        return a.second != b.second ? a.second > b.second : a.first < b.first;
    });
    return out;
}

// ba_solver.cc:495-521
std::vector<int> MostCovisible(int frame_id, Map &map, size_t num_images = 4) {
    std::vector<int> ids;
    for (const auto &fc : CovisibleFrames(map.frames_[frame_id], map, true, nullptr)) {
        ids.push_back(fc.first);
        if (ids.size() == num_images) break;
    }
    return ids;
}

// ba_solver.cc:393-493: the frame itself plus up to num_images-1 covisible frames that pass a ladder of
// (triangulation angle, overlap) thresholds, strictest first.
std::vector<int> LocalBundle(int frame_id, Map &map, size_t num_images = 4) {
    Frame &frame = map.frames_[frame_id];
    int num_p3d = 0;
    const auto cov = CovisibleFrames(frame, map, false, &num_p3d);
    const size_t wanted = std::min(num_images, cov.size() + 1);
    std::vector<int> ids;
    ids.reserve(wanted);
    ids.push_back(frame_id);
    if (cov.size() + 1 == wanted) {
        for (const auto &fc : cov) ids.push_back(fc.first);
        return ids;
    }
    const double base_angle = 6 * 0.01745329;
    const double angle_div[8] = {1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 5.0, 6.0};
    const double overlap_frac[8] = {0.6, 0.6, 0.5, 0.4, 0.3, 0.2, 0.1, 0.1};
    const auto centre = frame.Tcw.center();
    std::vector<double> angle(cov.size(), -1.0);
    std::vector<char> taken(cov.size(), 0);
    std::vector<decltype(frame.Tcw.center())> pts;
    for (int level = 0; level < 8 && ids.size() < wanted; ++level) {
        const double min_angle = base_angle / angle_div[level], min_overlap = overlap_frac[level] * num_p3d;
        for (size_t k = 0; k < cov.size(); ++k) {
            if (cov[k].second < min_overlap) break;
            if (taken[k]) continue;
            const Frame &other = map.frames_[cov[k].first];
            if (angle[k] < 0.0) {
                pts.clear();
                for (const int tid : frame.track_ids_)
                    if (tid != -1) pts.push_back(map.tracks_[tid].point3d_);
                auto other_pose = other.Tcw;
                angle[k] = colmap::Percentile(colmap::CalculateTriangulationAngles(centre, other_pose.center(), pts), 75);
            }
            if (angle[k] >= min_angle) {
                ids.push_back(static_cast<int>(other.id));
                taken[k] = 1;
                if (ids.size() >= wanted) break;
            }
        }
    }
    return ids;
}

} // namespace

void BASolver::GBA(Map &map, bool accurate, bool fix_all_frames) {
    FlatProblem problem(map, obs_cache_.get());
    for (auto &frame : map.frames_)
        if (frame.registered) problem.AddFrame(frame);
    if (!fix_all_frames) {
        problem.FixTranslation(map.init_id1);
        problem.FixTranslation(map.init_id2);
    } else {
        for (auto &frame : map.frames_)
            if (frame.registered) problem.FixPose(static_cast<int>(frame.id));
    }
    xrsfm_ba_options opt = accurate ? ReferenceOptions(50, 1e-5, 1e-6) : ReferenceOptions(20, 1e-4, 1e-5);
    opt.verbose = 1;   // minimizer_progress_to_stdout = true (:625)
    xrsfm_ba_summary summary;
    last_status_ = problem.Solve(opt, &summary);
    if (last_status_ == XRSFM_BA_OK) PrintSummary(summary);
}

void BASolver::KGBA(Map &map, const std::vector<int> fix_key_frame_ids, const bool is_sequential_data) {
    KeyFrameSelection(map, fix_key_frame_ids, is_sequential_data);
    int num_rf = 0, num_kf = 0;
    FlatProblem problem(map, obs_cache_.get());
    for (auto &frame : map.frames_) {
        if (!frame.registered) continue;
        ++num_rf;
        if (!frame.is_keyframe) continue;
        ++num_kf;
        problem.AddFrame(frame);
    }
    problem.FixTranslation(map.init_id1);
    problem.FixTranslation(map.init_id2);
    xrsfm_ba_options opt = ReferenceOptions(20, 1e-4, 1e-5);
    opt.initial_radius = 1e6;
    opt.verbose = 1;
    xrsfm_ba_summary summary;
    last_status_ = problem.Solve(opt, &summary);
    if (last_status_ == XRSFM_BA_OK) PrintSummary(summary);
    printf("kf: %d/%d\n", num_kf, num_rf);
    UpdateByRefFrame(map);
}

void BASolver::LBA(int frame_id, Map &map) {
    const std::vector<int> neighbours = MostCovisible(frame_id, map);
    const std::vector<int> bundle = LocalBundle(frame_id, map);
    std::set<int> local(neighbours.begin(), neighbours.end());
    local.insert(bundle.begin(), bundle.end());

    FlatProblem problem(map);
    printf("LBA: ");
    for (const int id : local) {
        printf(" %d", id);
        problem.AddFrame(map.frames_.at(id), frame_id);
    }
    printf("\n");

    // gauge: the init frames if present, else the two last frames of the local bundle / neighbour list
    int fixed = 0;
    if (local.count(map.init_id1)) { problem.FixTranslation(map.init_id1); ++fixed; }
    if (local.count(map.init_id2)) { problem.FixTranslation(map.init_id2); ++fixed; }
    if (fixed == 0) {
        const std::vector<int> *src = bundle.size() >= 2 ? &bundle : (neighbours.size() >= 2 ? &neighbours : nullptr);
        if (src) {
            problem.FixTranslation((*src)[src->size() - 1]);
            problem.FixTranslation((*src)[src->size() - 2]);
        } else {
            printf("!!!LBA only one frame\n");
            problem.FixTranslation(frame_id);
        }
    }
    xrsfm_ba_options opt = ReferenceOptions(5, 1e-4, 1e-5);
    xrsfm_ba_summary summary;
    last_status_ = problem.Solve(opt, &summary);   // the reference prints nothing for LBA (:586-591)
}

int RefineFramePose(Frame &frame, const Camera &camera, const std::vector<vector3> &points3ds,
                    const std::vector<std::pair<int, int>> &id_pair_vec, const std::vector<char> &inlier_mask) {
    const size_t n = inlier_mask.size();          // the reference loops over inlier_mask (pnp.cc:43)
    std::vector<double> P(3 * n), uv(2 * n);
    std::vector<uint8_t> mask(n);
    for (size_t id = 0; id < n; ++id) {
        mask[id] = inlier_mask[id] ? 1 : 0;
        const vector2 &p2d = frame.points[id_pair_vec[id].first];
        for (int k = 0; k < 3; ++k) P[3 * id + k] = points3ds[id].data()[k];
        uv[2 * id] = p2d.data()[0]; uv[2 * id + 1] = p2d.data()[1];
    }
    double prm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t k = 0; k < camera.params_.size() && k < 8; ++k) prm[k] = camera.params_[k];
    double q[4], t[3];
    for (int k = 0; k < 4; ++k) q[k] = frame.Tcw.q.coeffs().data()[k];
    for (int k = 0; k < 3; ++k) t[k] = frame.Tcw.t.data()[k];
    xrsfm_ba_summary s;
    const int e = xrsfm_ba_refine_pose(nullptr, static_cast<int32_t>(camera.model_id_), prm, static_cast<int32_t>(n), P.data(), uv.data(),
                                       mask.data(), q, t, &s);
    if (e != XRSFM_BA_OK) {
        fprintf(stderr, "[xrsfm_ba] pose refinement failed with code %d; pose left unchanged\n", e);
        return e;
    }
    for (int k = 0; k < 4; ++k) frame.Tcw.q.coeffs().data()[k] = q[k];
    for (int k = 0; k < 3; ++k) frame.Tcw.t.data()[k] = t[k];
    static const bool trace = std::getenv("XRSFM_BA_TRACE_CALLS") != nullptr;
    if (trace)
        std::fprintf(stderr, "[BASolver adapter] RefineFramePose frame %d: %d correspondences, LM %d+%d, %.3f -> %.3f px, %.3f ms\n", static_cast<int>(frame.id),
                     static_cast<int>(n), s.n_successful, s.n_unsuccessful, std::sqrt(s.initial_cost / s.num_residuals), std::sqrt(s.final_cost / s.num_residuals),
                     1e3 * s.total_time_s);
    std::cout << "Initial cost : " << std::setprecision(6) << std::sqrt(s.initial_cost / s.num_residuals) << " [px]" << std::endl;
    std::cout << "Final cost : " << std::setprecision(6) << std::sqrt(s.final_cost / s.num_residuals) << " [px]" << std::endl;
    return e;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pose graph with per-frame scale (loop closing).  Pose algebra on raw arrays (q = x,y,z,w; the Map's Eigen types are only
// touched through .coeffs().data() / .data(), which the reference's types and the test shim share).
namespace {
struct RawPose { double q[4] = {0, 0, 0, 1}; double t[3] = {0, 0, 0}; };

inline void QMul(const double *a, const double *b, double *o) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
inline void QInv(const double *a, double *o) {       // Eigen::Quaternion::inverse: conjugate / squaredNorm
    const double n = a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3];
    o[0] = -a[0] / n; o[1] = -a[1] / n; o[2] = -a[2] / n; o[3] = a[3] / n;
}
inline void QRot(const double *q, const double *v, double *o) {     // Eigen's q * v:  v + w*uv + q.vec x uv,  uv = 2 q.vec x v
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
inline RawPose PInv(const RawPose &p) {              // Pose::inverse (types.h:47-52)
    RawPose r;
    QInv(p.q, r.q);
    double v[3];
    QRot(r.q, p.t, v);
    for (int k = 0; k < 3; ++k) r.t[k] = -v[k];
    return r;
}
inline RawPose PMul(const RawPose &a, const RawPose &b) {   // a.mul(b) (types.h:54-58)
    RawPose r;
    double v[3];
    QRot(a.q, b.t, v);
    for (int k = 0; k < 3; ++k) r.t[k] = v[k] + a.t[k];
    QMul(a.q, b.q, r.q);
    return r;
}
template <typename P> inline RawPose Load(const P &pose) {
    RawPose r;
    for (int k = 0; k < 4; ++k) r.q[k] = pose.q.coeffs().data()[k];
    for (int k = 0; k < 3; ++k) r.t[k] = pose.t.data()[k];
    return r;
}
template <typename P> inline void Store(const RawPose &r, P &pose) {
    for (int k = 0; k < 4; ++k) pose.q.coeffs().data()[k] = r.q[k];
    for (int k = 0; k < 3; ++k) pose.t.data()[k] = r.t[k];
}
} // namespace

void BASolver::ScalePoseGraphUnorder(const LoopInfo &loop_info, Map &map, bool use_key) {
    // Step 1 (:150-196): the frame every map point is re-expressed in after the graph has moved.  Among the frames that see the
    // point (the loop frame excluded) in FRONT of the camera the order of preference is: key frame before non-key frame, then the
    // nearer one, then the lower frame id; if the point lies behind every such frame the first observing frame is kept (with
    // its negative depth, which the reference then reports).  This is what the reference's running-best loop computes.
    struct RefChoice { int frame = -1; double depth = -1.0; bool key = false; };
    const auto depth_in = [&](int frame_id, const Track &track) {
        const RawPose tcw = Load(map.frames_[frame_id].Tcw);
        double pc[3];
        QRot(tcw.q, track.point3d_.data(), pc);
        return pc[2] + tcw.t[2];
    };
    const auto preferred = [](const RefChoice &cand, const RefChoice &cur) {      // strict: an equal candidate does not replace
        if (cur.depth < 0) return true;
        if (cand.key != cur.key) return cand.key;
        return cand.depth < cur.depth;
    };
    for (auto &track : map.tracks_) {
        if (track.outlier) continue;
        RefChoice best;
        for (const auto &obs : track.observations_) {
            if (obs.first == loop_info.frame_id) continue;
            const RefChoice cand{obs.first, depth_in(obs.first, track), map.frames_[obs.first].is_keyframe};
            if (best.frame == -1 || (cand.depth >= 0 && preferred(cand, best))) best = cand;
        }
        track.ref_id = best.frame;
        track.depth = best.depth;
        if (best.depth < 0) {                          // the reference's diagnostics (stdout is observable behaviour)
            std::cout << "!!! negative depth\n";
            if (!track.observations_.empty()) {
                const auto first = track.observations_.begin();
                const int track_id = map.frames_[first->first].track_ids_[first->second];
                printf("-%d %d %lf\n", track_id, track.ref_id, track.depth);
                for (const auto &obs : track.observations_) printf("%d %d %lf\n", track_id, obs.first, depth_in(obs.first, track));
            }
        }
        if (best.frame == -1) std::cout << "!!! no frame_id\n";
    }

    const size_t num_frames = map.frames_.size();
    const size_t num_loop = loop_info.cor_frame_ids_vec.size();
    std::vector<RawPose> twc(num_frames);
    for (auto &frame : map.frames_)
        if (frame.registered) twc[frame.id] = PInv(Load(frame.Tcw));
    std::vector<double> scale(num_frames + num_loop, 1.0);       // s_vec | s_vec_loop

    double weight_o = 0.0;
    constexpr double max_th = 0.1;
    if (std::fabs(loop_info.scale_obs - 1) < max_th) {
        weight_o = 1 - std::fabs(loop_info.scale_obs - 1) / max_th;
        printf("weight scale: %lf\n", weight_o);
    }

    std::vector<int32_t> ea, eb, esa, esb;
    std::vector<double> q_mea, p_mea;
    std::vector<int> num_cov(num_frames, 0);
    auto add_edge = [&](const RawPose &pose1_mea, const RawPose &pose2, int a, int b, int sa, int sb) {
        double qi[4], q[4], d[3], p[3];
        QInv(pose1_mea.q, qi);
        QMul(qi, pose2.q, q);
        for (int k = 0; k < 3; ++k) d[k] = pose2.t[k] - pose1_mea.t[k];
        QRot(qi, d, p);
        ea.push_back(a); eb.push_back(b); esa.push_back(sa); esb.push_back(sb);
        q_mea.insert(q_mea.end(), q, q + 4); p_mea.insert(p_mea.end(), p, p + 3);
    };
    // covisibility edges, measured on the current estimate (:79-115)
    for (auto &frame : map.frames_) {
        if (!frame.registered) continue;
        if (use_key && !frame.is_keyframe) continue;
        const auto it = map.frameid2covisible_frameids_.find(frame.id);
        if (it == map.frameid2covisible_frameids_.end()) continue;
        for (const auto &cor_id : it->second) {
            if (static_cast<int>(frame.id) <= cor_id) continue;
            if (use_key && !map.frames_[cor_id].is_keyframe) continue;
            add_edge(twc[frame.id], twc[cor_id], frame.id, cor_id, frame.id, cor_id);
            num_cov[frame.id]++; num_cov[cor_id]++;
        }
    }
    // loop edges: the loop frame against every matched component, measured with that component's pose estimate (:117-145)
    for (size_t i = 0; i < num_loop; ++i) {
        const RawPose pose1_mea = Load(loop_info.twc_vec[i]);
        int count = 0;
        for (const auto &cor_id : loop_info.cor_frame_ids_vec[i]) {
            if (use_key && !map.frames_[cor_id].is_keyframe) continue;
            add_edge(pose1_mea, twc[cor_id], loop_info.frame_id, cor_id, static_cast<int>(num_frames + i), cor_id);
            count++;
            num_cov[cor_id]++; num_cov[loop_info.frame_id]++;
        }
        printf("loop_cor: %d num_edge: %d\n", static_cast<int>(i), count);
    }
    std::vector<int32_t> sca, scb;
    std::vector<double> s12;
    if (loop_info.scale_obs != -1 && num_loop >= 2) {
        printf("s12:%lf %zu %zu\n", loop_info.scale_obs, loop_info.cor_frame_ids_vec[0].size(), loop_info.cor_frame_ids_vec[1].size());
        sca.push_back(static_cast<int32_t>(num_frames)); scb.push_back(static_cast<int32_t>(num_frames + 1)); s12.push_back(loop_info.scale_obs);
    }

    // bounds and constants (:233-256): scales >= 0.2 except the loop frame's own; gauge = position and scale of the init frames
    std::vector<double> lower(num_frames + num_loop, -HUGE_VAL);
    std::vector<uint8_t> pos_const(num_frames, 0), scale_const(num_frames + num_loop, 0);
    for (auto &frame : map.frames_) {
        if (!frame.registered) continue;
        if (use_key && !frame.is_keyframe) continue;
        if (num_cov[frame.id] == 0) { printf("%d no covisiblity\n", frame.id); continue; }
        if (static_cast<int>(frame.id) != loop_info.frame_id) lower[frame.id] = 0.2;
    }
    for (size_t i = 0; i < num_loop && i < 2; ++i) lower[num_frames + i] = 0.2;
    for (const int id : {map.init_id1, map.init_id2})
        if (id >= 0 && id < static_cast<int>(num_frames)) { pos_const[id] = 1; scale_const[id] = 1; }

    std::vector<double> rot(4 * num_frames), pos(3 * num_frames);
    for (size_t i = 0; i < num_frames; ++i) {
        for (int k = 0; k < 4; ++k) rot[4 * i + k] = twc[i].q[k];
        for (int k = 0; k < 3; ++k) pos[3 * i + k] = twc[i].t[k];
    }
    xrsfm_pg_problem pg{};
    pg.n_frames = static_cast<int32_t>(num_frames); pg.n_scales = static_cast<int32_t>(num_frames + num_loop);
    pg.n_edges = static_cast<int32_t>(ea.size()); pg.n_scale_costs = static_cast<int32_t>(sca.size());
    pg.rot_q = rot.data(); pg.pos = pos.data(); pg.scale = scale.data();
    pg.pos_const = pos_const.data(); pg.scale_const = scale_const.data(); pg.scale_lower = lower.data();
    pg.edge_a = ea.data(); pg.edge_b = eb.data(); pg.edge_sa = esa.data(); pg.edge_sb = esb.data();
    pg.edge_q_mea = q_mea.data(); pg.edge_p_mea = p_mea.data(); pg.weight_o = weight_o;
    pg.sc_a = sca.data(); pg.sc_b = scb.data(); pg.sc_s12 = s12.data();
    xrsfm_pg_options opt;
    xrsfm_pg_default_options(&opt);           // 100 iterations, DOGLEG, radius 1e16 (:258-262)
    opt.verbose = 1;                          // minimizer_progress_to_stdout (:260)
    xrsfm_pg_summary summary;
    last_status_ = xrsfm_pg_solve(&opt, &pg, &summary);
    if (last_status_ != XRSFM_BA_OK) {
        fprintf(stderr, "[xrsfm_ba] pose graph failed with code %d; map left unchanged\n", last_status_);
        return;
    }
    static const char *const kTerm[] = {"", "CONVERGENCE", "CONVERGENCE", "CONVERGENCE", "CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
    std::cout << "Pose graph report: Iterations: " << summary.iterations << ", Initial cost: " << summary.initial_cost
              << ", Final cost: " << summary.final_cost << ", Termination: " << kTerm[summary.termination] << "\n";
    for (size_t i = 0; i < num_frames; ++i)
        for (int k = 0; k < 3; ++k) twc[i].t[k] = pos[3 * i + k];
    const std::vector<double> &s_vec = scale;

    const size_t stride = std::max<size_t>(1, num_frames / 10);
    if (!use_key) {
        for (size_t i = 0; i < num_frames; ++i)
            if (i % stride == 0 || s_vec[i] < 0) std::cout << i << " " << s_vec[i] << std::endl;
        for (size_t i = 0; i < num_frames; ++i) Store(PInv(twc[i]), map.frames_[i].Tcw);        // every frame (:271-273)
    } else {
        int num_keyframe = 0;
        for (const auto &frame : map.frames_)
            if (frame.registered && frame.is_keyframe) num_keyframe++;
        const int kstride = std::max(1, num_keyframe / 10);
        int count = 0;
        for (size_t i = 0; i < num_frames; ++i) {
            auto &frame = map.frames_[i];
            if (frame.registered && frame.is_keyframe) {
                if (count % kstride == 0 || s_vec[i] < 0) std::cout << i << " " << s_vec[i] << std::endl;
                count++;
            }
        }
        for (size_t i = 0; i < num_frames; ++i) {
            auto &frame = map.frames_[i];
            if (frame.registered && frame.is_keyframe) { frame.tcw_old = frame.Tcw; Store(PInv(twc[i]), frame.Tcw); }
        }
        // non-key frames follow their reference key frame, their relative pose scaled by its scale (:293-303)
        for (size_t i = 0; i < num_frames; ++i) {
            auto &frame = map.frames_[i];
            if (frame.registered && !frame.is_keyframe) {
                const auto &ref_frame = map.frames_[frame.ref_id];
                scale[i] = scale[frame.ref_id];
                RawPose tcc2 = PMul(Load(frame.Tcw), PInv(Load(ref_frame.tcw_old)));
                for (int k = 0; k < 3; ++k) tcc2.t[k] *= scale[frame.ref_id];
                Store(PMul(tcc2, Load(ref_frame.Tcw)), frame.Tcw);
            }
        }
    }
    for (size_t i = 0; i < num_loop && i < 2; ++i) std::cout << scale[num_frames + i] << std::endl;

    // re-express every map point through its reference frame: depth scaled by that frame's scale (:311-327)
    for (auto &track : map.tracks_) {
        if (track.outlier) continue;
        const int frame_id = track.ref_id;
        if (frame_id < 0) continue;
        const int p2d_id = track.observations_[frame_id];
        const RawPose tcw = Load(map.frames_[frame_id].Tcw);
        const vector2 p2d = map.GetNormalizedPoint(frame_id, p2d_id);
        const double sd = scale[frame_id] * track.depth;
        const double v[3] = {sd * p2d.data()[0] - tcw.t[0], sd * p2d.data()[1] - tcw.t[1], sd - tcw.t[2]};
        double qi[4], pw[3];
        QInv(tcw.q, qi);
        QRot(qi, v, pw);
        for (int k = 0; k < 3; ++k) track.point3d_.data()[k] = pw[k];
    }
}

} // namespace xrsfm
