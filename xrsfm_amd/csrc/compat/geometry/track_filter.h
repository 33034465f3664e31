// Source-level drop-in for the post-BA track filter: the arithmetic of
// Point3dProcessor::FilterPoints3d (/root/reference/src/geometry/track_processor.cc:321-332, with FilterPoint3d :280-319,
// UpdateTrackAngle :253-278, Reprojection_Error :19-26) runs on the GPU through xrsfm_ba_filter_tracks; this function packs the
// Map, calls it and applies the result to the Map exactly as the reference does (erase observations, SetTrackOutlier :76-82,
// Track::error, Track::angle_), printing the same "Outlier num1 / num2" line.  The maintainer replaces the body of
// Point3dProcessor::FilterPoints3d by   return FilterPoints3dGPU(map, max_re, deg);
#ifndef XRSFM_AMD_COMPAT_TRACK_FILTER_H
#define XRSFM_AMD_COMPAT_TRACK_FILTER_H
#include "base/map.h"

namespace xrsfm {
// returns num_filtered1 + num_filtered2 like the reference; a negative XRSFM_BA_E* code if the GPU call failed (Map untouched)
int FilterPoints3dGPU(Map &map, const double max_re, const double deg);
// Point3dProcessor::FilterPointsFrame (track_processor.cc:334-349, called after every LBA at incremental_mapper.cc:63-75):
// the same filter over the tracks the features of one frame are attached to
int FilterPointsFrameGPU(Map &map, const int frame_id, const double max_re, const double deg);
} // namespace xrsfm
#endif
