// TEST HARNESS for xrsfm_amd/csrc/compat/tag/tag_refine_solve.cc against the shim Map.
// in : i32 n_frames n_tags n_tag_obs n_points n_obs; f64 tag_length; frames {q[4] t[3]}; corners [n_tags][4][3];
//      tag obs {i32 tag, i32 frame}[n_tag_obs], f64 xy [n_tag_obs][8]; points [n_points][3]; obs {i32 frame, i32 pt}[n_obs], f64 xy[n_obs][2]
// Frame ids are 10 + 3 i, track ids 100 + 2 j, tag ids 7 + 5 k (the containers are keyed by id).  One unregistered frame
// and one outlier track with observations are added: both must be ignored (tag_extract.hpp:238-246).
// out: f64 scale; frames t [n_frames][3]; points [n_points][3]; corners; tag q [n_tags][4]; tag t [n_tags][3]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tag/tag_refine_solve.h"

template <typename T> static std::vector<T> rd(FILE *f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    auto hdr = rd<int32_t>(f, 5);
    const int nf = hdr[0], nt = hdr[1], nto = hdr[2], np = hdr[3], no = hdr[4];
    const double tag_length = rd<double>(f, 1)[0];
    auto fr = rd<double>(f, 7 * (size_t)nf); auto corners = rd<double>(f, 12 * (size_t)nt);
    auto to_idx = rd<int32_t>(f, 2 * (size_t)nto); auto to_xy = rd<double>(f, 8 * (size_t)nto);
    auto P = rd<double>(f, 3 * (size_t)np); auto o_idx = rd<int32_t>(f, 2 * (size_t)no); auto o_xy = rd<double>(f, 2 * (size_t)no);
    fclose(f);
    xrsfm::Map map;
    auto fid = [](int i) { return 10 + 3 * i; };
    auto tid = [](int j) { return 100 + 2 * j; };
    for (int i = 0; i < nf; ++i) {
        auto &frame = map.frame_map_[fid(i)];
        frame.id = fid(i); frame.registered = true;
        for (int k = 0; k < 4; ++k) frame.Tcw.q.coeffs().data()[k] = fr[7 * (size_t)i + k];
        for (int k = 0; k < 3; ++k) frame.Tcw.t.data()[k] = fr[7 * (size_t)i + 4 + k];
    }
    for (int j = 0; j < np; ++j) for (int k = 0; k < 3; ++k) map.track_map_[tid(j)].point3d_.data()[k] = P[3 * (size_t)j + k];
    for (int i = 0; i < no; ++i) {
        auto &frame = map.frame_map_[fid(o_idx[2 * i])];
        xrsfm::vector2 xy; xy(0) = o_xy[2 * (size_t)i]; xy(1) = o_xy[2 * (size_t)i + 1];
        map.track_map_[tid(o_idx[2 * i + 1])].observations_[fid(o_idx[2 * i])] = (int)frame.points_normalized.size();
        frame.points_normalized.push_back(xy); frame.points.push_back(xy); frame.track_ids_.push_back(tid(o_idx[2 * i + 1]));
    }
    for (auto &[id, frame] : map.frame_map_) {        // a feature without a track, a feature on an outlier track
        frame.points_normalized.push_back(xrsfm::vector2()); frame.points.push_back(xrsfm::vector2()); frame.track_ids_.push_back(-1);
        frame.points_normalized.push_back(xrsfm::vector2()); frame.points.push_back(xrsfm::vector2()); frame.track_ids_.push_back(5);
    }
    map.track_map_[5].outlier = true;
    map.track_map_[5].point3d_(0) = 3.0;
    {   // an unregistered frame whose features point at a real track
        auto &frame = map.frame_map_[4];
        frame.id = 4; frame.registered = false; frame.Tcw.t(0) = 8.0;
        frame.points_normalized.push_back(xrsfm::vector2()); frame.points.push_back(xrsfm::vector2()); frame.track_ids_.push_back(tid(0));
    }
    std::map<int, std::map<int, std::vector<xrsfm::vector2>>> tag_obs;
    std::map<int, std::vector<xrsfm::vector3>> pt_world_vec;
    auto gid = [](int k) { return 7 + 5 * k; };
    for (int k = 0; k < nt; ++k) {
        std::vector<xrsfm::vector3> pw(4);
        for (int c = 0; c < 4; ++c) for (int d = 0; d < 3; ++d) pw[c](d) = corners[12 * (size_t)k + 3 * c + d];
        pt_world_vec[gid(k)] = pw;
    }
    for (int i = 0; i < nto; ++i) {
        std::vector<xrsfm::vector2> pts(4);
        for (int c = 0; c < 4; ++c) { pts[c](0) = to_xy[8 * (size_t)i + 2 * c]; pts[c](1) = to_xy[8 * (size_t)i + 2 * c + 1]; }
        tag_obs[gid(to_idx[2 * i])][fid(to_idx[2 * i + 1])] = pts;
    }
    std::map<int, xrsfm::Pose> tag_vec;
    const double scale = xrsfm::RefineMapWithTags(map, tag_obs, pt_world_vec, tag_length, &tag_vec);
    FILE *o = fopen(argv[2], "wb");
    fwrite(&scale, 8, 1, o);
    for (int i = 0; i < nf; ++i) fwrite(map.frame_map_[fid(i)].Tcw.t.data(), 8, 3, o);
    for (int j = 0; j < np; ++j) fwrite(map.track_map_[tid(j)].point3d_.data(), 8, 3, o);
    for (int k = 0; k < nt; ++k) for (int c = 0; c < 4; ++c) fwrite(pt_world_vec[gid(k)][c].data(), 8, 3, o);
    for (int k = 0; k < nt; ++k) fwrite(tag_vec[gid(k)].q.coeffs().data(), 8, 4, o);
    for (int k = 0; k < nt; ++k) fwrite(tag_vec[gid(k)].t.data(), 8, 3, o);
    double extra[2] = {map.frame_map_[4].Tcw.t(0), map.track_map_[5].point3d_(0)};      // rescaled like everything else
    fwrite(extra, 8, 2, o);
    fclose(o);
    return scale > 0 ? 0 : 1;
}
