// OPTIONAL part of the adapter: the two Map-level routines that bracket BASolver::KGBA, for builds that use the adapter WITHOUT
// the reference's src/base/map.cc (BA-only replays, the test harness).  In a drop-in build the reference's own definitions are
// linked and this file is left out.
//
// Behaviour restated from /root/reference/src/base/map.cc (no code shared):
//   KeyFrameSelection   :428-607   which key frames are redundant, loop-matched frames forced to key frames, reference key frame
//                                  of every other frame, Track::is_keypoint
//   UpdateByRefFrame    :642-663   non-key frames follow their reference key frame rigidly
// Two places of the reference are undefined behaviour and are given the obvious meaning here: the scan over consecutive
// covisible key frames with fewer than two of them (:476-478), and `frame.id - i > 0` on an unsigned id (:571).
#include <algorithm>
#include <climits>
#include <cstdio>
#include <iostream>
#include <map>
#include <set>
#include <vector>

#include "base/map.h"

namespace xrsfm {
namespace {
struct RawPose { double q[4]; double t[3]; };
template <typename P> RawPose Load(const P &pose) {
    RawPose r;
    for (int k = 0; k < 4; ++k) r.q[k] = pose.q.coeffs().data()[k];
    for (int k = 0; k < 3; ++k) r.t[k] = pose.t.data()[k];
    return r;
}
template <typename P> void Store(const RawPose &r, P &pose) {
    for (int k = 0; k < 4; ++k) pose.q.coeffs().data()[k] = r.q[k];
    for (int k = 0; k < 3; ++k) pose.t.data()[k] = r.t[k];
}
void QMul(const double *a, const double *b, double *o) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by; o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw; o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
void QRot(const double *q, const double *v, double *o) {
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
RawPose Inverse(const RawPose &p) {          // Pose::inverse (types.h:47-52)
    RawPose r;
    const double n = p.q[0] * p.q[0] + p.q[1] * p.q[1] + p.q[2] * p.q[2] + p.q[3] * p.q[3];
    r.q[0] = -p.q[0] / n; r.q[1] = -p.q[1] / n; r.q[2] = -p.q[2] / n; r.q[3] = p.q[3] / n;
    double v[3];
    QRot(r.q, p.t, v);
    for (int k = 0; k < 3; ++k) r.t[k] = -v[k];
    return r;
}
RawPose Mul(const RawPose &a, const RawPose &b) {   // a.mul(b) (types.h:54-58)
    RawPose r;
    double v[3];
    QRot(a.q, b.t, v);
    for (int k = 0; k < 3; ++k) r.t[k] = v[k] + a.t[k];
    QMul(a.q, b.q, r.q);
    return r;
}
int CountKeyObservers(const Map &map, const Track &track, int except_frame) {
    int n = 0;
    for (const auto &obs : track.observations_)
        if (obs.first != except_frame && map.frames_[obs.first].is_keyframe) ++n;
    return n;
}
} // namespace

void KeyFrameSelection(Map &map, std::vector<int> loop_matched_frame_id, const bool is_sequential_data) {
    constexpr int kMinOtherKeyObservers = 3, kMinRedundant = 200;
    constexpr double kMinRedundantRatio = 0.6;
    // 1. a key frame (other than the two initial ones) is dropped when most of what it sees is seen by three other key frames as
    //    well and, for sequential data, the key frames on both sides of it stay connected.  Decisions take effect at once.
    for (auto &frame : map.frames_) {
        if (!frame.registered) continue;
        frame.tcw_old = frame.Tcw;
        const int fid = static_cast<int>(frame.id);
        if (!frame.is_keyframe || fid == map.init_id1 || fid == map.init_id2) continue;
        int num_p3d = 0, num_redundant = 0;
        for (const int track_id : frame.track_ids_) {
            if (track_id == -1) continue;
            ++num_p3d;
            if (CountKeyObservers(map, map.tracks_[track_id], fid) >= kMinOtherKeyObservers) ++num_redundant;
        }
        if (num_redundant < kMinRedundant || num_redundant < kMinRedundantRatio * num_p3d) continue;
        const auto cov = map.frameid2covisible_frameids_.find(fid);
        if (cov == map.frameid2covisible_frameids_.end() || cov->second.empty()) continue;
        if (is_sequential_data) {
            std::set<int> key_neighbours;
            for (const int id : cov->second)
                if (id != fid && map.frames_[id].is_keyframe) key_neighbours.insert(id);
            int min_connect = INT_MAX;
            if (key_neighbours.size() >= 2) {
                for (auto it = key_neighbours.begin(); std::next(it) != key_neighbours.end(); ++it) {
                    const int before = *it, after = *std::next(it);
                    if (!(before < fid && after > fid)) continue;
                    int shared = 0;
                    for (const int track_id : map.frames_[before].track_ids_)
                        if (track_id != -1 && map.tracks_[track_id].observations_.count(after)) ++shared;
                    min_connect = std::min(min_connect, shared);
                }
            }
            if (min_connect < kMinRedundant) continue;
        }
        frame.is_keyframe = false;
        printf("!!! init remove: %d %d %d\n", fid, num_redundant, num_p3d);
    }
    // 2. frames matched by the loop detector become (stay) key frames
    for (const int frame_id : loop_matched_frame_id) {
        std::cout << "|" << frame_id << std::endl;
        map.frames_[frame_id].ref_id = -1;
        map.frames_[frame_id].is_keyframe = true;
    }
    // 3. every other registered frame without a reference gets the key frame it shares most (non-outlier) tracks with; without
    //    any, the nearest key frame by id
    const int n_frames = static_cast<int>(map.frames_.size());
    for (auto &frame : map.frames_) {
        if (!frame.registered || frame.is_keyframe || frame.ref_id != -1) continue;
        const int fid = static_cast<int>(frame.id);
        std::map<int, int> shared;
        for (const int track_id : frame.track_ids_) {
            if (track_id == -1) continue;
            const Track &track = map.tracks_[track_id];
            if (track.outlier) continue;
            for (const auto &obs : track.observations_)
                if (obs.first != fid && map.frames_[obs.first].is_keyframe) ++shared[obs.first];
        }
        if (shared.empty()) {
            fprintf(stderr, "no covisiblity key frame\n");
            for (int i = 1; i < n_frames; ++i) {
                if (fid + i < n_frames && map.frames_[fid + i].is_keyframe) { frame.ref_id = fid + i; break; }
                if (fid - i > 0 && map.frames_[fid - i].is_keyframe) { frame.ref_id = fid - i; break; }
            }
        } else {
            int best = -1, best_count = 0;
            for (const auto &kv : shared)            // ascending ids: the smallest id wins a tie
                if (kv.second > best_count) { best = kv.first; best_count = kv.second; }
            frame.ref_id = best;
        }
    }
    // 4. a map point seen by two key frames takes part in the key-frame BA
    for (auto &track : map.tracks_) {
        if (track.outlier) continue;
        track.is_keypoint = CountKeyObservers(map, track, -1) >= 2;
    }
}

void UpdateByRefFrame(Map &map) {
    for (auto &frame : map.frames_) {
        if (!frame.registered || frame.is_keyframe) continue;
        Frame *ref = &map.frames_[frame.ref_id];
        for (int hops = 0; !ref->is_keyframe; ++hops) {            // a reference that lost its key-frame status: follow the chain
            ref = &map.frames_[ref->ref_id];
            if (hops >= 100) { fprintf(stderr, "too many loop %d %d\n", static_cast<int>(ref->id), ref->ref_id); break; }
        }
        frame.ref_id = static_cast<int>(ref->id);
        // Tcw = tcw_old * (ref.tcw_old^-1 * ref.Tcw): the pose relative to the reference frame is kept
        Store(Mul(Load(frame.tcw_old), Mul(Inverse(Load(ref->tcw_old)), Load(ref->Tcw))), frame.Tcw);
    }
}
} // namespace xrsfm
