#include "track_filter.h"

#include <cmath>
#include <cstdio>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "xrsfm_ba.h"

namespace xrsfm {

namespace {

int filter_tracks_gpu(Map &map, const std::vector<int> &track_ids, const double max_re, const double deg) {
    const double min_tri_angle_rad = deg * 0.0174532925199432954743716805978692718781530857086181640625;   // colmap::DegToRad
    // flat problem: camera = frame (index = frame id), point = non-outlier track, observations in the order of
    // Track::observations_ (ascending frame id), which is the order the reference visits them in
    const int n_frames = static_cast<int>(map.frames_.size());
    std::vector<double> cam_q(4 * (size_t)n_frames), cam_t(3 * (size_t)n_frames);
    std::vector<int32_t> cam_intr(n_frames, 0), intr_model;
    std::vector<double> intr_params;
    std::unordered_map<int, int> intr_slot;
    for (int i = 0; i < n_frames; ++i) {
        const Frame &frame = map.frames_[i];
        for (int k = 0; k < 4; ++k) cam_q[4 * (size_t)i + k] = frame.Tcw.q.coeffs().data()[k];
        for (int k = 0; k < 3; ++k) cam_t[3 * (size_t)i + k] = frame.Tcw.t.data()[k];
        auto it = intr_slot.find(static_cast<int>(frame.camera_id));
        if (it == intr_slot.end()) {
            if (map.camera_map_.count(frame.camera_id) == 0) continue;          // a frame that no track refers to
            const Camera &camera = map.Camera(frame.camera_id);
            it = intr_slot.emplace(static_cast<int>(frame.camera_id), static_cast<int>(intr_model.size())).first;
            intr_model.push_back(static_cast<int32_t>(camera.model_id_));
            for (size_t k = 0; k < 8; ++k) intr_params.push_back(k < camera.params_.size() ? camera.params_[k] : 0.0);
        }
        cam_intr[i] = it->second;
    }
    if (intr_model.empty()) { intr_model.push_back(0); intr_params.assign(8, 1.0); }
    std::vector<int> track_of_point;
    std::vector<double> points, obs_uv;
    std::vector<int32_t> obs_cam, obs_pt;
    int n_empty_outliers = 0;
    for (const int j : track_ids) {
        Track &track = map.tracks_[j];
        if (track.observations_.empty()) {
            // not reachable from the mapper (a track is born with two observations); the reference's unsigned "size() - 1"
            // sends it down the keep branch: error 0/0, angle 0, outlier iff 0 < threshold
            track.error = std::nan("");
            track.angle_ = 0;
            if (track.angle_ < min_tri_angle_rad) { track.outlier = true; ++n_empty_outliers; }
            continue;
        }
        const int pj = static_cast<int>(track_of_point.size());
        track_of_point.push_back(static_cast<int>(j));
        for (int k = 0; k < 3; ++k) points.push_back(track.point3d_.data()[k]);
        for (const auto &obs : track.observations_) {
            obs_cam.push_back(obs.first); obs_pt.push_back(pj);
            const auto &p2d = map.frames_[obs.first].points[obs.second];
            obs_uv.push_back(p2d.data()[0]); obs_uv.push_back(p2d.data()[1]);
        }
    }
    xrsfm_ba_problem p{};
    p.n_cams = n_frames; p.n_points = static_cast<int32_t>(track_of_point.size()); p.n_obs = static_cast<int32_t>(obs_cam.size());
    p.n_intr = static_cast<int32_t>(intr_model.size());
    p.cam_q = cam_q.data(); p.cam_t = cam_t.data(); p.cam_intr = cam_intr.data();
    p.intr_model = intr_model.data(); p.intr_params = intr_params.data();
    p.points = points.data(); p.obs_cam = obs_cam.data(); p.obs_pt = obs_pt.data(); p.obs_uv = obs_uv.data();
    std::vector<uint8_t> obs_delete(obs_cam.size() + 1), track_outlier(track_of_point.size() + 1);
    std::vector<double> track_error(track_of_point.size() + 1), track_angle(track_of_point.size() + 1);
    int32_t n[2] = {0, 0};
    const int e = xrsfm_ba_filter_tracks(&p, max_re, min_tri_angle_rad, obs_delete.data(), track_outlier.data(), track_error.data(),
                                         track_angle.data(), n);
    if (e != XRSFM_BA_OK) {
        fprintf(stderr, "[xrsfm_ba] track filter failed with code %d; map left unchanged\n", e);
        return e;
    }
    // apply (FilterPoint3d :300-318, SetTrackOutlier :76-82)
    size_t o = 0;
    for (size_t pj = 0; pj < track_of_point.size(); ++pj) {
        Track &track = map.tracks_[track_of_point[pj]];
        const size_t n_obs = track.observations_.size();
        auto unlink = [&](int frame_id, int p2d_id) {
            map.frames_[frame_id].track_ids_[p2d_id] = -1;
            map.DeleteNumCorHavePoint3D(frame_id, p2d_id);
        };
        if (track_outlier[pj] == 1) {                     // at most one observation would remain: the whole track goes
            track.outlier = true;
            for (const auto &obs : track.observations_) unlink(obs.first, obs.second);
            o += n_obs;
            continue;
        }
        std::vector<std::pair<int, int>> gone;
        for (const auto &obs : track.observations_) { if (obs_delete[o]) gone.push_back(obs); ++o; }
        for (const auto &g : gone) { track.observations_.erase(g.first); unlink(g.first, g.second); }
        track.error = track_error[pj];
        track.angle_ = track_angle[pj];
        if (track_outlier[pj] == 2) {                     // triangulation angle too small
            track.outlier = true;
            for (const auto &obs : track.observations_) unlink(obs.first, obs.second);
        }
    }
    n[1] += n_empty_outliers;
    printf("Outlier num1: %d Outlier num2: %d\n", n[0], n[1]);
    return n[0] + n[1];
}

} // namespace

int FilterPoints3dGPU(Map &map, const double max_re, const double deg) {
    std::vector<int> track_ids;
    for (size_t j = 0; j < map.tracks_.size(); ++j)
        if (!map.tracks_[j].outlier) track_ids.push_back(static_cast<int>(j));
    return filter_tracks_gpu(map, track_ids, max_re, deg);
}

int FilterPointsFrameGPU(Map &map, const int frame_id, const double max_re, const double deg) {
    // every track once, in the order of its first key point: a frame that lists one track at two key points (pnp.cc:84 guards
    // against it, the map format does not) must not enter the flat problem twice — the masks come back per track
    std::vector<int> track_ids;
    std::unordered_set<int> seen;
    for (const int track_id : map.frames_[frame_id].track_ids_)
        if (track_id != -1 && seen.insert(track_id).second) track_ids.push_back(track_id);
    return filter_tracks_gpu(map, track_ids, max_re, deg);
}

} // namespace xrsfm
