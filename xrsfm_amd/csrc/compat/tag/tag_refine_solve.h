// Source-level drop-in for the optimisation part of tag_refine (/root/reference/src/tag/tag_extract.hpp:193-275): everything
// between the triangulation of the tag corners and WriteColMapDataBinary2, i.e. the ceres::Problem with TagCost /
// ProjectionCost / QuatParam, its two ceres::Solve calls and the rescaling of the map.  Detection (apriltag, OpenCV), the
// normalisation of the observations and CreatePoint3dRAW stay where they are.  In tag_extract.hpp the block from
// "double scale = 1.0;" (:193) to the end of the "// resize map" loops (:275) becomes
//     std::map<int, Pose> tag_vec;
//     const double scale = RefineMapWithTags(map, tag_obs_normalized, pt_world_vec, tag_length, &tag_vec);
// and the Ceres includes of that header go away.
#ifndef XRSFM_AMD_COMPAT_TAG_REFINE_SOLVE_H
#define XRSFM_AMD_COMPAT_TAG_REFINE_SOLVE_H
#include <map>
#include <vector>

#include "base/map.h"

namespace xrsfm {
// map: frame_map_ / track_map_ as ReadColMapDataBinary fills them, Frame::points_normalized set for registered frames.
// tag_obs_normalized[tag_id][frame_id] = four normalised corners; pt_world_vec[tag_id] = four triangulated corners (refined
// in place by the second solve).  Returns the scale the map was divided by (a value <= 0: the XRSFM_BA_E* code of a failed
// call, map untouched).
double RefineMapWithTags(Map &map, const std::map<int, std::map<int, std::vector<vector2>>> &tag_obs_normalized,
                         std::map<int, std::vector<vector3>> &pt_world_vec, const double tag_length,
                         std::map<int, Pose> *tag_vec = nullptr);
} // namespace xrsfm
#endif
