// TEST SHIM implementations (deterministic stand-ins; the real ones live in the reference's src/base/map.cc etc.)
#include <cmath>
#include "base/map.h"
#include "geometry/colmap/base/triangulation.h"
#ifndef SHIM_REAL_MAP_OPS      // (the second harness binary links compat/base/map_ops.cc instead of these stand-ins)
namespace xrsfm {
void KeyFrameSelection(Map &map, std::vector<int> forced, const bool) {
    for (auto &f : map.frames_) {
        f.tcw_old = f.Tcw;
        f.is_keyframe = f.registered && (f.id % 2 == 0 || (int)f.id == map.init_id1 || (int)f.id == map.init_id2);
    }
    for (int id : forced) map.frames_[id].is_keyframe = true;
}
void UpdateByRefFrame(Map &) {}
} // namespace xrsfm
#endif
namespace colmap {
std::vector<double> CalculateTriangulationAngles(const xrsfm::vector3 &c1, const xrsfm::vector3 &c2,
                                                 const std::vector<xrsfm::vector3> &pts) {
    std::vector<double> out;
    for (const auto &p : pts) {
        double a[3], b[3], na = 0, nb = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { a[k] = p.v[k] - c1.v[k]; b[k] = p.v[k] - c2.v[k]; na += a[k] * a[k]; nb += b[k] * b[k]; dot += a[k] * b[k]; }
        const double cs = dot / std::sqrt(na * nb);
        out.push_back(std::acos(cs > 1 ? 1 : (cs < -1 ? -1 : cs)));
    }
    return out;
}
} // namespace colmap
