// Source-compatible replacement of /root/reference/src/optimization/ba_solver.h:14-30.
//
// Same class name, namespace and public signatures, so src/mapper/incremental_mapper.{h,cc}
// (member `BASolver ba_solver`, calls at incremental_mapper.cc:33,71,81), src/geometry/error_corrector.cc:220,236
// and src/run_triangulation.cc:180 compile unchanged; GBA / KGBA / LBA run on the MI355X through the C-ABI of
// include/xrsfm_ba.h instead of ceres::Solve.  No Ceres header is needed by this file.
//
// ScalePoseGraphUnorder (pose graph, DOGLEG; ba_solver.cc:147-328) is not part of the BA hot path; it is restated on top of
// the host-side xrsfm_pg_solve so that this class needs no Ceres at all.
#ifndef XRSFM_SRC_OPTIMIZATION_BA_SOLVER_H
#define XRSFM_SRC_OPTIMIZATION_BA_SOLVER_H

#include <memory>
#include <utility>
#include <vector>

#include "base/map.h"

namespace xrsfm {
// The "pose estimate [refine]" block of RegisterImage (/root/reference/src/geometry/pnp.cc:38-71) without Ceres: refines
// frame.Tcw against the inlier 2D-3D correspondences (frame.points[p2d_id] <-> points3ds[id], id_pair_vec[id] = (p2d_id,
// track_id), inlier_mask from SolvePnP_colmap) on the GPU, prints the same two "[px]" lines.  pnp.cc includes this header
// already; the block there becomes the single call  RefineFramePose(frame, camera, points3ds, id_pair_vec, inlier_mask);
// Returns 0 or a negative XRSFM_BA_E* code (the pose is left untouched on error).
int RefineFramePose(Frame &frame, const Camera &camera, const std::vector<vector3> &points3ds,
                    const std::vector<std::pair<int, int>> &id_pair_vec, const std::vector<char> &inlier_mask);

// Number of GBA / KGBA / LBA calls of this process whose solve did not run (the map was left unchanged each time).  The
// reference's call sites cannot see a status; a mapper that wants a hard error sets XRSFM_BA_ABORT_ON_FAILURE=1 or checks this at
// the end of the reconstruction.
int &BASolverFailureCount();

struct BASolverObsCache;       // observation arrays of the last global BA, reused when the same frames and tracks come again (ba_solver.cc)

class BASolver {
  public:
    // (the reference's constructor is empty, ba_solver.h:16; this one pays the one-off start-up cost of the GPU library — HIP runtime,
    //  code object, first stream — once per process, so that the mapper's first BA call does not: xrsfm_ba_warmup, include/xrsfm_ba.h)
    BASolver();

    void ScalePoseGraphUnorder(const LoopInfo &loop_info, Map &map, bool use_key = false);
    void KGBA(Map &map, const std::vector<int> fix_key_frame_ids, const bool is_sequential_data);
    void GBA(Map &map, bool accurate = true, bool fix_all_frames = false);
    void LBA(int frame_id, Map &map);

    // Result of the last GBA/KGBA/LBA call (0 = ok, negative = XRSFM_BA_E* code).  The reference ignores the
    // Ceres summary (ba_solver.cc:636-637); callers that want to know can read this.
    int last_status() const { return last_status_; }

  private:
    int last_status_ = 0;
    std::shared_ptr<BASolverObsCache> obs_cache_;      // (shared: a copied BASolver keeps working, like the reference's empty class)
};
} // namespace xrsfm
#endif // XRSFM_SRC_OPTIMIZATION_BA_SOLVER_H
