// Test driver for the adapter: builds a xrsfm::Map (shim types) from a flat problem dump, calls BASolver, dumps the Map.
// usage: adapter_main <in.bin> <out.bin> <mode: gba|gba_fast|structure|kgba|lba|refine|posegraph|keyframes|refframe> [frame_id | loop.bin]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "geometry/track_filter.h"
#include "optimization/ba_solver.h"

template <typename T> static std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    auto hdr = rd<int32_t>(f, 4);
    const int nc = hdr[0], np = hdr[1], no = hdr[2], ni = hdr[3];
    auto q = rd<double>(f, 4 * (size_t)nc); auto t = rd<double>(f, 3 * (size_t)nc); auto cam_intr = rd<int32_t>(f, nc);
    auto model = rd<int32_t>(f, ni); auto prm = rd<double>(f, 8 * (size_t)ni); auto P = rd<double>(f, 3 * (size_t)np);
    auto oc = rd<int32_t>(f, no); auto op = rd<int32_t>(f, no); auto uv = rd<double>(f, 2 * (size_t)no);
    fclose(f);
    static const int nparams[5] = {3, 4, 4, 5, 8};
    xrsfm::Map map;
    for (int i = 0; i < ni; ++i) { xrsfm::Camera c; c.id_ = i; c.model_id_ = model[i]; c.params_.assign(prm.begin() + 8 * i, prm.begin() + 8 * i + nparams[model[i]]); map.camera_map_[i] = c; }
    map.frames_.resize(nc); map.tracks_.resize(np);
    for (int i = 0; i < nc; ++i) {
        auto &fr = map.frames_[i]; fr.id = i; fr.camera_id = cam_intr[i]; fr.registered = true;
        for (int k = 0; k < 4; ++k) fr.Tcw.q.coeffs().data()[k] = q[4 * i + k];
        for (int k = 0; k < 3; ++k) fr.Tcw.t.data()[k] = t[3 * i + k];
    }
    for (int j = 0; j < np; ++j) for (int k = 0; k < 3; ++k) map.tracks_[j].point3d_.data()[k] = P[3 * j + k];
    for (int i = 0; i < no; ++i) {
        auto &fr = map.frames_[oc[i]];
        xrsfm::vector2 p2; p2(0) = uv[2 * i]; p2(1) = uv[2 * i + 1];
        map.tracks_[op[i]].observations_[oc[i]] = (int)fr.points.size();
        fr.points.push_back(p2); fr.track_ids_.push_back(op[i]);
    }
    // an extra 2D feature without a track in every frame: must be skipped (track_ids_ == -1)
    for (auto &fr : map.frames_) { fr.points.push_back(xrsfm::vector2()); fr.track_ids_.push_back(-1); }
    map.init_id1 = 0; map.init_id2 = 1;
    if (std::string(argv[3]) == "lba" && argc > 6) { map.init_id1 = atoi(argv[5]); map.init_id2 = atoi(argv[6]); }     // LBA gauge cases
    xrsfm::BASolver solver;
    int refine_status = 0;
    bool extra = false, filter_dump = false;
    int filter_ret = 0;
    const std::string mode = argv[3];
    if (mode == "gba") solver.GBA(map);
    else if (mode == "gba_timed" || mode == "gba_cached") {
        // tools/adapter_timing.py: the second call is the warm one (a COPY of the map first: other storage, the adapter's observation
        // cache misses), the third repeats the second on the SAME map from the same state — the cache hit (ba_solver.cc:
        // BASolverObsCache).  gba_cached (tests/test_adapter.py): exit code 3 unless the cached call reproduces the uncached one bit for bit.
        { xrsfm::Map warm = map; solver.GBA(warm); }
        const xrsfm::Map start = map;
        solver.GBA(map);
        const xrsfm::Map first = map;
        for (size_t i = 0; i < map.frames_.size(); ++i) map.frames_[i].Tcw = start.frames_[i].Tcw;
        for (size_t j = 0; j < map.tracks_.size(); ++j) map.tracks_[j].point3d_ = start.tracks_[j].point3d_;
        solver.GBA(map);
        bool same = true;
        for (size_t i = 0; i < map.frames_.size() && same; ++i)
            same = memcmp(map.frames_[i].Tcw.q.coeffs().data(), first.frames_[i].Tcw.q.coeffs().data(), 4 * sizeof(double)) == 0 &&
                   memcmp(map.frames_[i].Tcw.t.data(), first.frames_[i].Tcw.t.data(), 3 * sizeof(double)) == 0;
        for (size_t j = 0; j < map.tracks_.size() && same; ++j)
            same = memcmp(map.tracks_[j].point3d_.data(), first.tracks_[j].point3d_.data(), 3 * sizeof(double)) == 0;
        if (!same) { fprintf(stderr, "gba_cached: the call served from the observation cache differs from the uncached one\n"); return 3; }
    }
    else if (mode == "gba_fast") solver.GBA(map, false);
    else if (mode == "structure") solver.GBA(map, true, true);
    else if (mode == "kgba") solver.KGBA(map, {3}, true);
    else if (mode == "lba") solver.LBA(argc > 4 ? atoi(argv[4]) : nc - 1, map);
    else if (mode == "refine") {
        // RegisterImage's refinement on frame 0: correspondences = its tracked features, every 7th one a RANSAC outlier
        auto &fr = map.frames_[0];
        std::vector<xrsfm::vector3> p3; std::vector<std::pair<int, int>> ids; std::vector<char> inl;
        for (size_t i = 0; i < fr.track_ids_.size(); ++i) {
            if (fr.track_ids_[i] == -1) continue;
            p3.push_back(map.tracks_[fr.track_ids_[i]].point3d_); ids.push_back({(int)i, fr.track_ids_[i]}); inl.push_back(ids.size() % 7 != 0);
        }
        refine_status = xrsfm::RefineFramePose(fr, map.Camera(fr.camera_id), p3, ids, inl);
    }
    else if (mode == "posegraph") {
        // loop.bin: i32 frame_id, i32 n_components, per component {i32 n, i32 ids[n], f64 twc[7] (q xyzw, t)}, f64 scale_obs, i32 use_key
        FILE *lf = fopen(argv[4], "rb");
        if (!lf) return 2;
        xrsfm::LoopInfo li;
        li.frame_id = rd<int32_t>(lf, 1)[0];
        const int ncomp = rd<int32_t>(lf, 1)[0];
        for (int c = 0; c < ncomp; ++c) {
            const int m = rd<int32_t>(lf, 1)[0];
            auto ids = rd<int32_t>(lf, m);
            auto pose = rd<double>(lf, 7);
            li.cor_frame_ids_vec.emplace_back(ids.begin(), ids.end());
            xrsfm::Pose tw;
            for (int k = 0; k < 4; ++k) tw.q.coeffs().data()[k] = pose[k];
            for (int k = 0; k < 3; ++k) tw.t.data()[k] = pose[4 + k];
            li.twc_vec.push_back(tw);
        }
        li.scale_obs = rd<double>(lf, 1)[0];
        const bool use_key = rd<int32_t>(lf, 1)[0] != 0;
        fclose(lf);
        // covisibility = frames sharing a track (ascending ids), key frames = even ids + init frames, ref = previous key frame
        for (auto &tr : map.tracks_)
            for (auto &o1 : tr.observations_)
                for (auto &o2 : tr.observations_)
                    if (o1.first != o2.first) {
                        auto &v = map.frameid2covisible_frameids_[o1.first];
                        if (std::find(v.begin(), v.end(), o2.first) == v.end()) v.push_back(o2.first);
                    }
        for (auto &kv : map.frameid2covisible_frameids_) std::sort(kv.second.begin(), kv.second.end());
        for (auto &fr : map.frames_) {
            fr.is_keyframe = !use_key || fr.id % 2 == 0 || (int)fr.id == map.init_id1 || (int)fr.id == map.init_id2 || (int)fr.id == li.frame_id;
            fr.ref_id = fr.is_keyframe ? (int)fr.id : (int)fr.id - 1;
        }
        solver.ScalePoseGraphUnorder(li, map, use_key);
    }
    else if (mode == "keyframes" || mode == "refframe") {
        // KeyFrameSelection / UpdateByRefFrame (the adapter's optional map_ops.cc in the _mapops binary): every frame starts as a
        // key frame, covisibility = frames sharing a track, frame argv[4] is forced (loop-matched)
        for (auto &tr : map.tracks_)
            for (auto &o1 : tr.observations_)
                for (auto &o2 : tr.observations_)
                    if (o1.first != o2.first) {
                        auto &v = map.frameid2covisible_frameids_[o1.first];
                        if (std::find(v.begin(), v.end(), o2.first) == v.end()) v.push_back(o2.first);
                    }
        for (auto &fr : map.frames_) fr.is_keyframe = true;
        xrsfm::KeyFrameSelection(map, {argc > 4 ? atoi(argv[4]) : 0}, true);
        if (mode == "refframe") {
            // move every key frame by one rigid motion of the world (Tcw' = Tcw * G): the others must follow exactly
            const double g_q[4] = {0.0, std::sin(0.05), 0.0, std::cos(0.05)}, g_t[3] = {0.3, -0.1, 0.2};
            for (auto &fr : map.frames_) {
                if (!fr.is_keyframe) continue;
                double *q = fr.Tcw.q.coeffs().data(), *t = fr.Tcw.t.data();
                const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
                // t' = R(q) g_t + t ; q' = q * g_q
                const double ux = 2 * (ay * g_t[2] - az * g_t[1]), uy = 2 * (az * g_t[0] - ax * g_t[2]), uz = 2 * (ax * g_t[1] - ay * g_t[0]);
                t[0] += g_t[0] + aw * ux + (ay * uz - az * uy); t[1] += g_t[1] + aw * uy + (az * ux - ax * uz); t[2] += g_t[2] + aw * uz + (ax * uy - ay * ux);
                q[0] = aw * g_q[0] + ax * g_q[3] + ay * g_q[2] - az * g_q[1]; q[1] = aw * g_q[1] - ax * g_q[2] + ay * g_q[3] + az * g_q[0];
                q[2] = aw * g_q[2] + ax * g_q[1] - ay * g_q[0] + az * g_q[3]; q[3] = aw * g_q[3] - ax * g_q[0] - ay * g_q[1] - az * g_q[2];
            }
            xrsfm::UpdateByRefFrame(map);
        }
        extra = true;
    }
    else if (mode == "filter") {
        // FilterPoints3dGPU(map, max_re = argv[4], deg = argv[5]); appended to the dump: per track {outlier, #observations, error, angle},
        // per frame the number of DeleteNumCorHavePoint3D calls and of features still attached to a track
        const int one_frame = argc > 6 ? atoi(argv[6]) : -1;
        filter_ret = one_frame < 0 ? xrsfm::FilterPoints3dGPU(map, atof(argv[4]), atof(argv[5]))
                                   : xrsfm::FilterPointsFrameGPU(map, one_frame, atof(argv[4]), atof(argv[5]));
        filter_dump = true;
    }
    else return 2;
    FILE *o = fopen(argv[2], "wb");
    int32_t st = mode == "refine" ? refine_status : (mode == "filter" ? filter_ret : solver.last_status());
    fwrite(&st, 4, 1, o);
    for (auto &fr : map.frames_) { fwrite(fr.Tcw.q.coeffs().data(), 8, 4, o); fwrite(fr.Tcw.t.data(), 8, 3, o); }
    for (auto &tr : map.tracks_) fwrite(tr.point3d_.data(), 8, 3, o);
    if (filter_dump) {
        for (auto &tr : map.tracks_) { double v[4] = {tr.outlier ? 1.0 : 0.0, (double)tr.observations_.size(), tr.error, tr.angle_}; fwrite(v, 8, 4, o); }
        map.shim_deleted_corr_.resize(map.frames_.size(), 0);
        for (size_t i = 0; i < map.frames_.size(); ++i) {
            int32_t attached = 0;
            for (int id : map.frames_[i].track_ids_) attached += id != -1;
            int32_t v[2] = {map.shim_deleted_corr_[i], attached};
            fwrite(v, 4, 2, o);
        }
    }
    if (extra) {
        for (auto &fr : map.frames_) { int32_t v[2] = {fr.is_keyframe ? 1 : 0, fr.ref_id}; fwrite(v, 4, 2, o); }
        for (auto &tr : map.tracks_) { int32_t v = tr.is_keypoint ? 1 : 0; fwrite(&v, 4, 1, o); }
    }
    fclose(o);
    return 0;
}
