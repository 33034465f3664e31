// TEST SHIM — not the reference's header.  Declares only the members of xrsfm::Map / Frame / Track / Camera / Pose that
// the adapter (xrsfm_amd/csrc/compat/optimization/ba_solver.cc) touches, with a minimal stand-in
// for the Eigen types, so the adapter can be compile- and run-tested in an image without Eigen/OpenCV/glog.  It is never
// used to build the reference.
#ifndef XRSFM_TEST_SHIM_MAP_H
#define XRSFM_TEST_SHIM_MAP_H
#include <cmath>
#include <cstdint>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace shim {
template <int N> struct Vec {
    double v[N] = {0};
    double *data() { return v; }
    const double *data() const { return v; }
    double &operator()(int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
};
struct Quat {            // coefficient order x,y,z,w like Eigen::Quaterniond::coeffs()
    Vec<4> c;
    Quat() { c.v[3] = 1.0; }
    Vec<4> &coeffs() { return c; }
    const Vec<4> &coeffs() const { return c; }
};
} // namespace shim

namespace xrsfm {
using vector2 = shim::Vec<2>;
using vector3 = shim::Vec<3>;

struct Pose {
    shim::Quat q;
    vector3 t;
    inline vector3 center() {   // -(q^-1 * t) for a unit quaternion
        const double x = -q.c.v[0], y = -q.c.v[1], z = -q.c.v[2], w = q.c.v[3];
        const double ux = y * t.v[2] - z * t.v[1], uy = z * t.v[0] - x * t.v[2], uz = x * t.v[1] - y * t.v[0];
        vector3 r;
        r.v[0] = -(t.v[0] + 2 * (w * ux + y * uz - z * uy));
        r.v[1] = -(t.v[1] + 2 * (w * uy + z * ux - x * uz));
        r.v[2] = -(t.v[2] + 2 * (w * uz + x * uy - y * ux));
        return r;
    }
};

class Camera {
  public:
    uint32_t id_ = -1;
    uint32_t model_id_ = -1;
    std::vector<double> params_;
};

class Track {
  public:
    std::map<int, int> observations_;
    vector3 point3d_;
    double angle_ = -1;
    bool outlier = false;
    int ref_id = -1;
    double depth = -1;
    bool is_keypoint = false;
    double error = 0;
};

class Frame {
  public:
    uint32_t id = -1;
    uint32_t camera_id = 0;
    bool registered = false;
    bool is_keyframe = false;
    std::vector<vector2> points;
    std::vector<vector2> points_normalized;
    std::vector<int> track_ids_;
    Pose Tcw, tcw_old;
    int ref_id = -1;
};

struct LoopInfo {
    int frame_id;
    std::vector<std::set<int>> cor_frame_ids_vec;
    std::vector<Pose> twc_vec;
    std::vector<int> num_inlier_vec;
    double scale_obs = -1;
};

class Map {
  public:
    std::vector<Track> tracks_;
    std::vector<Frame> frames_;
    std::map<int, class Camera> camera_map_;
    std::map<int, Frame> frame_map_;
    std::map<int, Track> track_map_;
    int init_id1 = -1;
    int init_id2 = -1;
    inline const class Camera &Camera(int camera_id) const { return camera_map_.at(camera_id); }
    inline class Camera &Camera(int camera_id) { return camera_map_.at(camera_id); }
    std::unordered_map<int, std::vector<int>> frameid2covisible_frameids_;
    // shim: the reference walks the correspondence graph here; the harness only counts the calls per frame
    std::vector<int> shim_deleted_corr_;
    inline void DeleteNumCorHavePoint3D(int frame_id, int /*p2d_id*/) {
        if (shim_deleted_corr_.size() < frames_.size()) shim_deleted_corr_.resize(frames_.size(), 0);
        shim_deleted_corr_[frame_id]++;
    }
    // shim: undistorted pinhole normalisation with the camera's first parameters (f, cx, cy); the reference inverts the
    // full distortion model (camera_model.hpp ImageToNormalized)
    inline vector2 GetNormalizedPoint(const int frame_id, const int p2d_id) {
        const auto &frame = frames_.at(frame_id);
        const auto &cam = Camera(frame.camera_id);
        vector2 r;
        r.v[0] = (frame.points.at(p2d_id).v[0] - cam.params_[1]) / cam.params_[0];
        r.v[1] = (frame.points.at(p2d_id).v[1] - cam.params_[2]) / cam.params_[0];
        return r;
    }
};

void KeyFrameSelection(Map &map, std::vector<int> loop_matched_frame_id, const bool is_sequential_data = false);
void UpdateByRefFrame(Map &map);
} // namespace xrsfm
#endif
