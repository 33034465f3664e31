#include "tag_refine_solve.h"

#include <cstdio>
#include <iostream>

#include "xrsfm_ba.h"

namespace xrsfm {

namespace {
const char *termination_text(int code) {
    switch (code) {
    case 1: return "CONVERGENCE (gradient tolerance)";
    case 2: return "CONVERGENCE (parameter tolerance)";
    case 3: return "CONVERGENCE (function tolerance)";
    case 4: return "CONVERGENCE (trust region radius)";
    case 5: return "NO_CONVERGENCE (max iterations)";
    default: return "FAILURE";
    }
}
void brief_report(const xrsfm_pg_summary &s) {      // the shape of ceres::Solver::Summary::BriefReport
    printf("xrsfm_ba Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s\n", s.iterations + 1, s.initial_cost,
           s.final_cost, termination_text(s.termination));
}
} // namespace

double RefineMapWithTags(Map &map, const std::map<int, std::map<int, std::vector<vector2>>> &tag_obs_normalized,
                         std::map<int, std::vector<vector3>> &pt_world_vec, const double tag_length, std::map<int, Pose> *tag_vec) {
    // frames and tracks by dense index (the containers are keyed by id)
    std::map<int, int> frame_slot, track_slot;
    std::vector<double> frame_q, frame_t;
    for (const auto &[id, frame] : map.frame_map_) {
        frame_slot[id] = static_cast<int>(frame_slot.size());
        for (int k = 0; k < 4; ++k) frame_q.push_back(frame.Tcw.q.coeffs().data()[k]);
        for (int k = 0; k < 3; ++k) frame_t.push_back(frame.Tcw.t.data()[k]);
    }
    std::vector<int> tag_ids;
    std::vector<double> corners, tag_obs_xy;
    std::vector<int32_t> tag_obs_tag, tag_obs_frame;
    for (const auto &[tag_id, frame_obs] : tag_obs_normalized) {
        const int k = static_cast<int>(tag_ids.size());
        tag_ids.push_back(tag_id);
        const auto &pt_world = pt_world_vec.at(tag_id);
        for (int c = 0; c < 4; ++c) for (int d = 0; d < 3; ++d) corners.push_back(pt_world[c].data()[d]);
        for (const auto &[frame_id, pts_n] : frame_obs) {
            tag_obs_tag.push_back(k); tag_obs_frame.push_back(frame_slot.at(frame_id));
            for (int c = 0; c < 4; ++c) { tag_obs_xy.push_back(pts_n[c].data()[0]); tag_obs_xy.push_back(pts_n[c].data()[1]); }
        }
    }
    // one ProjectionCost per feature of a registered frame that is attached to a non-outlier track (tag_extract.hpp:237-254)
    std::vector<double> points, obs_xy;
    std::vector<int32_t> obs_frame, obs_pt;
    std::vector<int> track_ids;
    for (const auto &[id, frame] : map.frame_map_) {
        if (!frame.registered) continue;
        for (size_t i = 0; i < frame.track_ids_.size(); ++i) {
            if (frame.track_ids_[i] == -1) continue;
            const Track &track = map.track_map_.at(frame.track_ids_[i]);
            if (track.outlier) continue;
            auto it = track_slot.find(frame.track_ids_[i]);
            if (it == track_slot.end()) {
                it = track_slot.emplace(frame.track_ids_[i], static_cast<int>(track_ids.size())).first;
                track_ids.push_back(frame.track_ids_[i]);
                for (int d = 0; d < 3; ++d) points.push_back(track.point3d_.data()[d]);
            }
            obs_frame.push_back(frame_slot.at(id)); obs_pt.push_back(it->second);
            obs_xy.push_back(frame.points_normalized[i].data()[0]); obs_xy.push_back(frame.points_normalized[i].data()[1]);
        }
    }
    const int n_tags = static_cast<int>(tag_ids.size());
    std::vector<double> tag_q(4 * (size_t)n_tags, 0.0), tag_t(3 * (size_t)n_tags, 0.0);
    for (int k = 0; k < n_tags; ++k) tag_q[4 * (size_t)k + 3] = 1.0;           // default-constructed Pose (tag_extract.hpp:209)
    xrsfm_tag_problem p{};
    p.n_frames = static_cast<int32_t>(frame_slot.size()); p.frame_q = frame_q.data(); p.frame_t = frame_t.data();
    p.n_tags = n_tags; p.tag_length = tag_length; p.tag_corners = corners.data(); p.tag_q = tag_q.data(); p.tag_t = tag_t.data();
    p.scale = 1.0; p.scale_lower = 0.2;
    p.n_tag_obs = static_cast<int32_t>(tag_obs_tag.size());
    p.tag_obs_tag = tag_obs_tag.data(); p.tag_obs_frame = tag_obs_frame.data(); p.tag_obs_xy = tag_obs_xy.data();
    p.n_points = static_cast<int32_t>(track_ids.size()); p.n_obs = static_cast<int32_t>(obs_pt.size());
    p.points = points.data(); p.obs_frame = obs_frame.data(); p.obs_pt = obs_pt.data(); p.obs_xy = obs_xy.data();
    xrsfm_pg_options opt;
    xrsfm_tag_default_options(&opt);
    opt.verbose = 1;                                   // minimizer_progress_to_stdout (tag_extract.hpp:230)
    xrsfm_pg_summary sums[2];
    const int e = xrsfm_tag_refine(&opt, &p, 2, sums);
    if (e != XRSFM_BA_OK) {
        fprintf(stderr, "[xrsfm_ba] tag refinement failed with code %d; map left unchanged\n", e);
        return static_cast<double>(e);
    }
    brief_report(sums[0]);
    brief_report(sums[1]);
    std::cout << p.scale << std::endl;
    // write back: refined corners, tag poses, track points; then the map goes to metric units (tag_extract.hpp:267-275)
    for (int k = 0; k < n_tags; ++k) {
        auto &pt_world = pt_world_vec.at(tag_ids[k]);
        for (int c = 0; c < 4; ++c) for (int d = 0; d < 3; ++d) pt_world[c].data()[d] = corners[12 * (size_t)k + 3 * c + d];
        if (tag_vec) {
            Pose &T_w_tag = (*tag_vec)[tag_ids[k]];
            for (int d = 0; d < 4; ++d) T_w_tag.q.coeffs().data()[d] = tag_q[4 * (size_t)k + d];
            for (int d = 0; d < 3; ++d) T_w_tag.t.data()[d] = tag_t[3 * (size_t)k + d];
        }
    }
    for (size_t j = 0; j < track_ids.size(); ++j) {
        Track &track = map.track_map_.at(track_ids[j]);
        for (int d = 0; d < 3; ++d) track.point3d_.data()[d] = points[3 * j + d];
    }
    const double scale = p.scale;
    for (auto &[id, frame] : map.frame_map_) for (int d = 0; d < 3; ++d) frame.Tcw.t.data()[d] /= scale;
    for (auto &[id, track] : map.track_map_) for (int d = 0; d < 3; ++d) track.point3d_.data()[d] /= scale;
    return scale;
}

} // namespace xrsfm
