/*
 * xrsfm_ba.h — C-ABI of the MI355X-native bundle-adjustment engine.
 *
 * Drop-in boundary for the global/local BA step of openxrlab/xrsfm.  The
 * reference has no FFI for this path: it is the C++ class xrsfm::BASolver
 * (/root/reference/src/optimization/ba_solver.h:14-30) whose GBA/KGBA/LBA
 * methods build a ceres::Problem out of raw pointers into Map storage
 * (/root/reference/src/optimization/ba_solver.cc:345-347) and call
 * ceres::Solve (ba_solver.cc:591,636,672).  The entry points below are what a
 * binding for that path has to call instead of Ceres: the same parameter
 * blocks, handed over as flat FP64/int32 arrays (caller-owned, results written
 * in place like Ceres does), the same solver options the reference sets
 * (ba_solver.cc:70-77, 586-589, 626-634, 667-670), and a summary carrying the
 * quantities PrintSolverSummary prints (ba_solver.cc:14-68).
 * The source-compatible adapter on top is xrsfm_amd/csrc/compat/.
 *
 * Plain C, no torch / HIP types in any signature.  All functions return 0 on
 * success or a negative XRSFM_BA_E* code; nothing throws across the boundary.
 */
#ifndef XRSFM_BA_H
#define XRSFM_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRSFM_BA_VERSION 1

/* error codes */
#define XRSFM_BA_OK 0
#define XRSFM_BA_EINVAL (-1)   /* bad argument / inconsistent indices            */
#define XRSFM_BA_ENODEV (-2)   /* no HIP device, or the HIP runtime reported an error */
#define XRSFM_BA_ENOMEM (-3)
#define XRSFM_BA_ECOMM (-4)    /* RCCL unavailable or a collective failed         */
#define XRSFM_BA_ESTATE (-5)   /* call order violated                            */
#define XRSFM_BA_ETOOBIG (-6)  /* explicit reduced camera matrix requested (CHOLESKY) but it does not fit: use AUTO or PCG */
#define XRSFM_BA_EINTERNAL (-7) /* an unexpected C++ exception was stopped at the boundary (never the termination code +2) */

/* camera models: ids of /root/reference/src/base/camera_model.hpp:93-209 */
#define XRSFM_BA_SIMPLE_PINHOLE 0 /* {f,cx,cy}             uv = 2f*xn + c (reference quirk, :102-105) */
#define XRSFM_BA_PINHOLE 1        /* {fx,fy,cx,cy}         same quirk (:121-124)                       */
#define XRSFM_BA_SIMPLE_RADIAL 2  /* {f,cx,cy,k}                                                      */
#define XRSFM_BA_RADIAL 3         /* {fx,fy,cx,cy,k}       single k (:155-177)                         */
#define XRSFM_BA_OPENCV 4         /* {fx,fy,cx,cy,k1,k2,p1,p2}                                         */

/* cam_const bits: which parameter blocks of a frame are held constant
 * (problem.SetParameterBlockConstant, ba_solver.cc:611-621) */
#define XRSFM_BA_CONST_Q 1u
#define XRSFM_BA_CONST_T 2u
/* bal9 mode (BASELINE.json north_star "2x9 camera blocks") — NOT something the reference does: it
 * always holds the intrinsics block of ReProjectionCost (cost_factor_ceres.h:42-46, kNumParams) constant (ba_solver.cc:
 * 602-606, 655-659, 389).  A camera with this bit keeps its intrinsics VARIABLE and contributes a 9-wide block {rotation 3,
 * translation 3, f, k1, k2}.  Requirements (XRSFM_BA_EINVAL otherwise): camera model 5 below, one intrinsics entry per such
 * camera; exact solver on one rank.  The refined {f, k1, k2} come back in problem->intr_params (xrsfm_ba_solve) /
 * xrsfm_ba_download_intrinsics.  The adapter never sets it. */
#define XRSFM_BA_INTR_VARIABLE 4u
/* Camera models: 0..4 = the reference's (camera_model.hpp:93-209); 5 = extension for BAL-style problems: params {f, k1, k2},
 * no principal point, uv = f (1 + k1 r^2 + k2 r^4) xy with the reference's sign convention xy = pc.hnormalized(). */
#define XRSFM_BA_MODEL_BAL 5

/* One BA call = one ceres::Problem of the reference (ba_solver.cc:596,645,536).
 * Observation order is free (the reference's is frame-major, ba_solver.cc:598-601). */
typedef struct xrsfm_ba_problem {
    int32_t n_cams;   /* frames added by SetUp/SetUpLBA (ba_solver.cc:330-391)          */
    int32_t n_points; /* tracks referenced by those frames                             */
    int32_t n_obs;    /* residual blocks = ReProjectionCost instances                  */
    int32_t n_intr;   /* distinct camera_id's (intrinsics always constant, :602-606)   */
    double *cam_q;              /* [n_cams][4]  Tcw.q.coeffs() = x,y,z,w   in/out        */
    double *cam_t;              /* [n_cams][3]  Tcw.t                      in/out        */
    const uint8_t *cam_const;   /* [n_cams]     XRSFM_BA_CONST_* bits, NULL = all free   */
    const int32_t *cam_intr;    /* [n_cams]     index into intr_*                        */
    const int32_t *intr_model;  /* [n_intr]     XRSFM_BA_<MODEL>                         */
    double *intr_params;        /* [n_intr][8]  camera.params_ (zero padded); read only, except in bal9 mode (XRSFM_BA_INTR_VARIABLE): in/out */
    double *points;             /* [n_points][3] track.point3d_            in/out        */
    const uint8_t *point_const; /* [n_points]   non-zero = constant (SetUpLBA :380-382), NULL = all free */
    const int32_t *obs_cam;     /* [n_obs] */
    const int32_t *obs_pt;      /* [n_obs] */
    const double *obs_uv;       /* [n_obs][2]   frame.points[i]                          */
} xrsfm_ba_problem;

#define XRSFM_BA_SOLVER_PCG 0      /* implicit-Schur PCG on the reduced camera system (any size)            */
#define XRSFM_BA_SOLVER_CHOLESKY 1 /* explicit reduced camera matrix + tile Cholesky (exact, what Ceres SPARSE_SCHUR
                                      computes).  Any camera graph up to 6*n_cams <= 12288; beyond that only band /
                                      ring graphs (sequential data: shallow elimination tree) while the tile storage
                                      fits (XRSFM_BA_ETOOBIG otherwise)                                      */
#define XRSFM_BA_SOLVER_AUTO 2     /* CHOLESKY whenever the rule above allows it, else PCG                  */

typedef struct xrsfm_ba_options {
    int32_t max_iterations;      /* GBA accurate 50 / fast 20 / KGBA 20 / LBA 5        */
    double function_tolerance;   /* 1e-5 / 1e-4                                       */
    double parameter_tolerance;  /* 1e-6 / 1e-5                                       */
    double gradient_tolerance;   /* Ceres default 1e-10                               */
    double initial_radius;       /* Ceres default 1e4, KGBA 1e6 (ba_solver.cc:667)    */
    double huber_a;              /* 5.99 (ba_solver.cc:343,374)                       */
    int32_t linear_solver;       /* XRSFM_BA_SOLVER_*                                 */
    double pcg_tolerance;        /* |r|_2 <= tol*|b|_2; 1e-12 follows the exact solve */
    int32_t pcg_max_iterations;
    int32_t profile;             /* !=0: HIP-event timing of the dominant kernel      */
    int32_t verbose;             /* !=0: Ceres-style progress table on stdout         */
} xrsfm_ba_options;

/* termination codes (ceres::TerminationType as printed by ba_solver.cc:41-66) */
#define XRSFM_BA_CONVERGENCE 0
#define XRSFM_BA_NO_CONVERGENCE 1
#define XRSFM_BA_FAILURE 2

typedef struct xrsfm_ba_summary {
    double initial_cost;   /* 1/2 sum rho(|r|^2) at entry                              */
    double final_cost;     /* ... at the last accepted state                           */
    int32_t num_residuals;          /* num_residuals_reduced = 2*n_obs                 */
    int32_t num_effective_params;   /* num_effective_parameters_reduced                */
    int32_t n_successful;           /* LM steps accepted.  Iteration 0 (the evaluation at the initial point) is NOT counted: a RECALLED
                                       detail of Ceres' num_successful_steps (UNPINNED, oracle/ba_oracle.py ALT_DETAILS
                                       "iteration_zero_counted"): if Ceres counts it, the reference's "Iterations :" line
                                       (ba_solver.cc:22-25) reads one more than n_successful + n_unsuccessful here */
    int32_t n_unsuccessful;         /* LM steps rejected or invalid                    */
    int32_t termination;            /* XRSFM_BA_CONVERGENCE / ...                      */
    int32_t termination_reason;     /* 1 gradient, 2 parameter, 3 function tolerance, 4 min radius, 5 max iterations, 6 invalid steps */
    int32_t pcg_iterations;         /* total over all LM steps                         */
    int32_t lm_steps_attempted;     /* incl. the step that triggered a tolerance exit  */
    double total_time_s;            /* wall time of the solve, device-resident inputs  */
    double dom_kernel_ms;           /* profile!=0: sum of HIP-event durations of the costliest kernel */
    int32_t dom_kernel_launches;    /* profile!=0: launches counted in dom_kernel_ms   */
    int32_t dom_kernel_id;          /* profile!=0: index for xrsfm_ba_profile_entry    */
    int32_t linear_solver_used;     /* XRSFM_BA_SOLVER_PCG or _CHOLESKY                */
    int32_t reserved;
} xrsfm_ba_summary;

typedef struct xrsfm_ba_context xrsfm_ba_context; /* opaque: device buffers, stream, communicator */

/* Fill `opt` with the reference's GBA(accurate=true) settings + Ceres defaults. */
void xrsfm_ba_default_options(xrsfm_ba_options *opt);

/* Library / device probe: returns XRSFM_BA_VERSION, *n_devices = visible HIP devices (0 if none). */
int xrsfm_ba_version(int *n_devices);

/* Optional: pay the one-off costs of the first call of a process NOW — HIP runtime start-up, loading the library's code
 * object, a stream with its pinned scalar block, the kernels' dynamic-LDS attributes and, with n_obs_hint > 0, the device
 * buffers of a problem of about that many observations / n_points_hint points / n_cams_hint cameras (they go to the
 * allocation cache the next xrsfm_ba_create draws from; capped at a quarter of the device memory that is free at the time of
 * the call).  The adapter calls it from BASolver's constructor (the reference
 * pays the equivalent when ceres::Problem is first used).  Returns XRSFM_BA_ENODEV without a device; never required. */
int xrsfm_ba_warmup(int device, int64_t n_obs_hint, int64_t n_points_hint, int64_t n_cams_hint);

/* Build a device-resident problem on HIP device `device`: validates indices,
 * orders tracks, uploads everything.  The host arrays are only read. */
int xrsfm_ba_create(const xrsfm_ba_problem *problem, int device, xrsfm_ba_context **out);

/* Multi-GPU (points sharded by rank, cameras replicated): attach an RCCL
 * communicator.  `unique_id` is the 128-byte ncclUniqueId obtained from
 * xrsfm_ba_comm_unique_id on rank 0 and distributed by the caller. */
int xrsfm_ba_comm_unique_id(unsigned char id[128]);
int xrsfm_ba_comm_init(xrsfm_ba_context *ctx, int n_ranks, int rank, const unsigned char id[128]);
/* Watchdog of multi-rank contexts (environment XRSFM_BA_WATCHDOG_S = seconds, default 300 with several ranks and OFF on one
 * rank; an explicit value applies to every context, 0 switches it off): a rank that waits that long for its device without
 * progress — an all-reduce a peer never joined — gets XRSFM_BA_ECOMM (XRSFM_BA_ENODEV on one rank) from xrsfm_ba_run.  The
 * context is then POISONED: run / reset / download return XRSFM_BA_ESTATE, and xrsfm_ba_destroy aborts the communicator and
 * releases the host side only (the stream and the device buffers the stuck work may still touch are leaked on purpose, never
 * waited for). */

/* TEST HOOK: replace the RCCL all-reduce of this context by a caller-supplied one working on a HOST copy of the buffer
 * (op 0 = sum, 1 = max; return 0 on success).  Lets several ranks share ONE GPU — RCCL refuses two ranks on one device — so the
 * multi-rank logic (sharded points, replicated cameras, union block pattern, identical LM decisions on every rank) can be
 * verified on a 1-GPU box with any host transport (the tests use torch.distributed/gloo).  Not a production path. */
typedef int (*xrsfm_ba_allreduce_fn)(void *user, double *host_buf, uint64_t n, int op);
int xrsfm_ba_debug_comm_hook(xrsfm_ba_context *ctx, int n_ranks, int rank, xrsfm_ba_allreduce_fn fn, void *user);

/* Run Levenberg-Marquardt on the device-resident state (blocking). */
int xrsfm_ba_run(xrsfm_ba_context *ctx, const xrsfm_ba_options *opt, xrsfm_ba_summary *summary);

/* Restore the device-resident state to the values uploaded by xrsfm_ba_create. */
int xrsfm_ba_reset(xrsfm_ba_context *ctx);

/* Copy the current state back into caller arrays laid out like the problem
 * (cam_q [n_cams][4], cam_t [n_cams][3], points [n_points][3]); NULL skips one. */
int xrsfm_ba_download(xrsfm_ba_context *ctx, double *cam_q, double *cam_t, double *points);

/* bal9 mode: intrinsics {f, k1, k2} of the cameras with XRSFM_BA_INTR_VARIABLE into intr_params [n_intr][8] (other rows and
 * columns untouched); a no-op for ordinary problems. */
int xrsfm_ba_download_intrinsics(xrsfm_ba_context *ctx, double *intr_params);

void xrsfm_ba_destroy(xrsfm_ba_context *ctx);

/* The library keeps process-wide caches between calls (the reference builds a fresh ceres::Problem per call,
 * ba_solver.cc:596,645,536; BASolver::LBA runs once per registered frame): device blocks and streams returned by
 * xrsfm_ba_destroy, and — for contexts above 200k observations — the release of their host arrays on one library-owned
 * thread.  xrsfm_ba_quiesce() waits for every deferred release and frees the cached device memory; it also returns the
 * bytes that were cached (device) through *cached_bytes (may be NULL).  Never required for correctness: the thread is joined
 * when the library is unloaded (dlclose / process exit), so no library code runs after the unload; cached device blocks
 * that were not released through this call stay with the process until it exits (the library does not call into the HIP
 * runtime from a static destructor).  An embedder that dlcloses the library should call this first. */
int xrsfm_ba_quiesce(uint64_t *cached_bytes);

/* Free and total memory of HIP device `device` as the runtime reports them (hipMemGetInfo): for embedders that watch the
 * footprint of a long mapping session (tests/test_mapper_replay.py asserts that replaying a reconstruction twice leaves it
 * unchanged).  XRSFM_BA_ENODEV without a device. */
int xrsfm_ba_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* One-shot convenience = create + run + download into problem->{cam_q,cam_t,points} + destroy:
 * the call that replaces ceres::Solve(options, &problem, &summary). */
int xrsfm_ba_solve(const xrsfm_ba_options *opt, xrsfm_ba_problem *problem, xrsfm_ba_summary *summary);

/* Pose-only refinement of one frame against fixed 3-D points: the "pose estimate [refine]" block of RegisterImage
 * (/root/reference/src/geometry/pnp.cc:38-71): one ReProjectionCost + HuberLoss(5.99) per inlier correspondence, points
 * and intrinsics constant, EigenQuaternionParameterization on q, ceres::Solver::Options defaults with
 * max_num_iterations = 10.  xrsfm_ba_refine_pose_options fills exactly those settings (function_tolerance 1e-6,
 * parameter_tolerance 1e-8, gradient_tolerance 1e-10, initial radius 1e4).
 *   model, intr_params[8]   camera model id 0..4 and its parameters (unused tail ignored)
 *   points3d [n][3], uv [n][2], inlier_mask [n] (NULL = all inliers; only non-zero entries enter, like pnp.cc:43-45)
 *   q[4] (x,y,z,w), t[3]    in: the RANSAC pose; out: the refined pose
 * summary->initial_cost / final_cost with num_residuals give the two "[px]" values the reference prints (pnp.cc:63-70).
 * One persistent workgroup runs the whole LM loop on the device (xrsfm_amd/csrc/ba_refine.h: normal equations by block reduction,
 * 6x6 damped solve and trust-region bookkeeping on one lane, candidate linearised in the pass that yields its cost): one
 * upload, one launch, one read-back, ~0.12 ms per call.  Same arithmetic and loop semantics as xrsfm_ba_solve on the
 * one-camera problem, which stays available as the cross-check (environment variable XRSFM_BA_REFINE_ENGINE=1). */
void xrsfm_ba_refine_pose_options(xrsfm_ba_options *opt);
int xrsfm_ba_refine_pose(const xrsfm_ba_options *opt, int32_t model, const double *intr_params, int32_t n,
                         const double *points3d, const double *uv, const uint8_t *inlier_mask, double *q, double *t,
                         xrsfm_ba_summary *summary);
/* The same for several frames in one upload / launch / read-back (one workgroup per frame): e.g. the candidate next frames of
 * incremental_mapper.cc:41-46 or the frames of a loop correction.  Frame f owns the correspondences corr_ptr[f] ..
 * corr_ptr[f+1] of the concatenated arrays (corr_ptr[0] = 0); models [n_frames], intr_params [n_frames][8], q [n_frames][4],
 * t [n_frames][3], summaries [n_frames].  Every frame gets exactly the result of its own xrsfm_ba_refine_pose call. */
int xrsfm_ba_refine_poses(const xrsfm_ba_options *opt, int32_t n_frames, const int32_t *models, const double *intr_params,
                          const int32_t *corr_ptr, const double *points3d, const double *uv, const uint8_t *inlier_mask,
                          double *q, double *t, xrsfm_ba_summary *summaries);

/* ---- Scaled pose graph of BASolver::ScalePoseGraphUnorder (/root/reference/src/optimization/ba_solver.cc:147-328; SURVEY 8f
 * row f4).  HOST code (O(frames) unknowns; no GPU needed or used): it completes the BASolver interface without Ceres.
 * Poses are T_wc (twc_vec of the reference).  Rotations are constant (ba_solver.cc:248-249), positions and scales are the
 * unknowns.  An edge is one PoseGraphCost(q_mea, p_mea, weight_o) residual block (cost_factor_ceres.h:117-198) between
 * pose1 = frame edge_a (scale edge_sa) and pose2 = frame edge_b (scale edge_sb); a scale cost is one ScaleCost(s12)
 * (cost_factor_ceres.h:200-221).  n_scales >= n_frames: scale i < n_frames belongs to frame i, the rest are the loop
 * scales (s_vec_loop). */
typedef struct xrsfm_pg_problem {
    int32_t n_frames, n_scales, n_edges, n_scale_costs;
    const double *rot_q;        /* [n_frames][4] x,y,z,w, constant */
    double *pos;                /* [n_frames][3] in/out */
    double *scale;              /* [n_scales]    in/out */
    const uint8_t *pos_const;   /* [n_frames] 1 = constant (NULL: none) */
    const uint8_t *scale_const; /* [n_scales] */
    const double *scale_lower;  /* [n_scales] lower bounds (-HUGE_VAL = none); NULL = unconstrained problem */
    const int32_t *edge_a, *edge_b, *edge_sa, *edge_sb;   /* [n_edges] */
    const double *edge_q_mea;   /* [n_edges][4] x,y,z,w */
    const double *edge_p_mea;   /* [n_edges][3] */
    double weight_o;            /* weight of the scale prior row (ba_solver.cc:221-229) */
    const int32_t *sc_a, *sc_b; /* [n_scale_costs] scale indices */
    const double *sc_s12;       /* [n_scale_costs] */
} xrsfm_pg_problem;

typedef struct xrsfm_pg_options {
    int32_t max_iterations;     /* 100 (InitSolverOptions, ba_solver.cc:73) */
    double function_tolerance, parameter_tolerance, gradient_tolerance;   /* Ceres defaults 1e-6, 1e-8, 1e-10 */
    double initial_radius;      /* 1e16 (ba_solver.cc:261) */
    int32_t verbose;
    /* 0 (default): bounded parameters are handled as Ceres handles them — projection in Plus + projected Armijo line search on
     * every step, nothing else (trust_region_minimizer.cc); with an active bound the loop may stop above the constrained minimum,
     * exactly as upstream does.  1: additionally hold a parameter that sits on its bound while the gradient pushes it outwards
     * (projected-Newton active set) — a deliberate DEVIATION from the reference that reaches the constrained minimum. */
    int32_t bounds_active_set;
} xrsfm_pg_options;

typedef struct xrsfm_pg_summary {
    double initial_cost, final_cost;
    int32_t iterations, n_successful, n_unsuccessful;
    int32_t termination;        /* 1 gradient, 2 parameter, 3 function tolerance, 4 radius, 5 max iterations, 6 failure (linear solver, non-finite input) */
} xrsfm_pg_summary;

void xrsfm_pg_default_options(xrsfm_pg_options *opt);
int xrsfm_pg_solve(const xrsfm_pg_options *opt, xrsfm_pg_problem *problem, xrsfm_pg_summary *summary);

/* ---- Metric-scale refinement against AprilTag corners: the two ceres::Solve calls of tag_refine
 * (/root/reference/src/tag/tag_extract.hpp:193-265).  HOST code like the pose graph.  Detection
 * (apriltag/OpenCV) and the RANSAC triangulation of the corners (CreatePoint3dRAW, tag_extract.hpp:176-192) stay with the
 * caller; this entry point takes the normalised observations and the triangulated corners.
 *   stage 1 (tag_extract.hpp:197-234): per tag 4 x TagCost(get_tag(tag_length)[i], 1.0) on (tag_q, tag_t, scale) with the
 *            corners constant, QuatParam on tag_q, scale >= scale_lower; max_num_iterations 500, other options default
 *   stage 2 (tag_extract.hpp:236-265): the corners become variable and carry one ProjectionCost per observing frame; every
 *            track point carries one ProjectionCost per observation; all frame poses stay constant
 * The caller then divides frame translations and points by the returned scale (tag_extract.hpp:267-275). */
typedef struct xrsfm_tag_problem {
    int32_t n_frames;
    const double *frame_q;      /* [n_frames][4] x,y,z,w  Tcw, constant */
    const double *frame_t;      /* [n_frames][3] */
    int32_t n_tags;
    double tag_length;
    double *tag_corners;        /* [n_tags][4][3] in: triangulated world corners (pt_world_vec); out (stage 2): refined */
    double *tag_q, *tag_t;      /* [n_tags][4], [n_tags][3]  T_w_tag in/out (the reference starts at identity / zero) */
    double scale;               /* in/out (the reference starts at 1.0) */
    double scale_lower;         /* 0.2 (tag_extract.hpp:227) */
    int32_t n_tag_obs;          /* (tag, frame) pairs */
    const int32_t *tag_obs_tag, *tag_obs_frame;
    const double *tag_obs_xy;   /* [n_tag_obs][4][2] normalised image coordinates of the four corners */
    int32_t n_points, n_obs;    /* tracks of the map (used by stage 2 only; n_obs may be 0) */
    double *points;             /* [n_points][3] in/out */
    const int32_t *obs_frame, *obs_pt;
    const double *obs_xy;       /* [n_obs][2] normalised image coordinates (Frame::points_normalized) */
} xrsfm_tag_problem;

void xrsfm_tag_default_options(xrsfm_pg_options *opt);   /* 500 iterations, radius 1e4, tolerances 1e-6 / 1e-8 / 1e-10 */
/* stages = 1: only the first solve; 2: both, like the reference.  summaries[stages]. termination codes as xrsfm_pg_summary. */
int xrsfm_tag_refine(const xrsfm_pg_options *opt, xrsfm_tag_problem *problem, int32_t stages, xrsfm_pg_summary *summaries);

/* Post-BA track filter on the same flat arrays (Point3dProcessor::FilterPoints3d,
 * /root/reference/src/geometry/track_processor.cc:280-332, called after every KGBA at incremental_mapper.cc:83-85).
 * The problem here is the whole map: every registered frame and every observation of every non-outlier track.
 *   obs_delete[i]    1: observation i has reprojection error > max_reproj_error or depth outside [1e-3, 1e3]
 *   track_outlier[j] 0 keep; 1: at most one observation would remain; 2: max pairwise triangulation angle of the kept
 *                    observations < min_tri_angle_rad (for 1 every observation of the track goes, like SetTrackOutlier)
 *   track_error[j]   mean reprojection error of the kept observations (-1 if outlier 1 / no observation)   (may be NULL)
 *   track_angle[j]   Track::angle_ as UpdateTrackAngle leaves it (early exit above the threshold)          (may be NULL)
 *   num_filtered[2]  the two counters the reference prints ("Outlier num1 / num2")                          (may be NULL) */
int xrsfm_ba_filter_tracks(const xrsfm_ba_problem *problem, double max_reproj_error, double min_tri_angle_rad,
                           uint8_t *obs_delete, uint8_t *track_outlier, double *track_error, double *track_angle,
                           int32_t *num_filtered);

/* profile != 0 in the last xrsfm_ba_run: per-kernel totals measured with HIP events on the
 * library's stream.  Returns 0 and fills the outputs for index < number of kernel classes,
 * XRSFM_BA_EINVAL past the end. */
int xrsfm_ba_profile_entry(xrsfm_ba_context *ctx, int index, const char **name, double *total_ms, int *launches);

/* ---- test/diagnostic entry points (kernel-level parity against the oracle) ---- */

/* Linearise at the current state with Jacobi scaling `use_scaling` (0: scale = 1).
 * Outputs are in the caller's observation / point / camera order; any may be NULL.
 *   r [n_obs][2], Jc [n_obs][2][6], Jp [n_obs][2][3]   robustified, scaled
 *   Hpp [n_points][6] (upper: 00 01 02 11 12 22), gp [n_points][3]
 *   Hcc_diag [n_cams][6], gc [n_cams][6], *cost */
int xrsfm_ba_debug_linearize(xrsfm_ba_context *ctx, double huber_a, int use_scaling, double *r, double *Jc,
                             double *Jp, double *Hpp, double *gp, double *Hcc_diag, double *gc, double *cost);

/* After debug_linearize: y = S(radius) * x for a caller vector x [n_cams][6]; also returns rhs b [n_cams][6]. */
int xrsfm_ba_debug_schur_product(xrsfm_ba_context *ctx, double radius, const double *x, double *y, double *b);

/* Host-side packing only (no GPU needed): runs the track-tile packing of xrsfm_ba_create and reports
 * stats[0] tiles, [1] slots (= 64*tiles), [2] work items, [3] regular tiles, [4] long items (tracks > 64 observations),
 * [5] camera-major partial entries, [6] longest track, [7] active points.  slot_obs (may be NULL) receives, for each of
 * the `slots` slots, the caller's observation index stored there or -1 for padding; it must hold n_obs + 64*(n_points+1)
 * entries at most (upper bound of the slot count). */
int xrsfm_ba_debug_pack(const xrsfm_ba_problem *problem, int32_t stats[8], int32_t *slot_obs);

/* The S-assembly side of the packing (no GPU needed): Gram tiles and the item classes of k_schur_pairs.
 * stats[0] Gram tiles, [1] cells of their C x C destination tables, [2] camera-major entries of the S assembly (one per
 * distinct camera of a Gram tile), [3] largest C, [4]/[5]/[6] items in the small-LDS Gram class / big-LDS Gram class /
 * per-pair + long class, [7] partial blocks written per pass.  tile_ncam (may be NULL): [tiles]; slot_cidx (may be NULL):
 * [slots], 255 outside Gram tiles; slot_campos_g (may be NULL): [slots], -1 for non-writers. */
int xrsfm_ba_debug_pack_gram(const xrsfm_ba_problem *problem, int32_t stats[8], int32_t *tile_ncam, uint8_t *slot_cidx,
                             int32_t *slot_campos_g);

/* TEST ENTRY (no GPU needed): the schedule of 4x4 result blocks by which k_schur_pairs forms the camera-pair blocks of a Gram tile
 * of n_cams cameras (6 operand rows per camera, groups of 4 rows; csrc/ba_chol.h: gram_tile4).  entries [4 * *n_inst]: four blocks
 * per matrix instruction, row group | column group << 8, padded with block (0, 0).  *n_inst = 0 when the tile takes the 16x16 form
 * (more than 6 instructions).  Returns XRSFM_BA_EINVAL for n_cams outside 1..10; entries must hold 128 values. */
int xrsfm_ba_debug_gram_schedule(int n_cams, int32_t *n_inst, int32_t *n_inst_all, uint16_t *entries);

/* Host-side plan of the Cholesky path (no GPU needed): stats[0] tiles T, [1] elimination-tree levels, [2] ordering
 * (0 natural, 1 nested dissection of a band/ring, 2 reverse Cuthill-McKee of an unordered collection, 3 nested dissection of an unordered collection's camera graph), [3] hub cameras, [4] band width in cameras, [5] off-diagonal blocks,
 * [6] schedule BITS: bit 0 (value 1) = level schedule (one launch per elimination-tree level; clear = the panel schedule of
 * deep elimination trees: dense / unordered patterns), bit 1 (value 2) = look-ahead panel schedule (one launch per tile
 * column, k_panel_slot) — test the bits, not the value: a look-ahead plan reports 2, [7] structurally
 * non-zero tiles after fill.  cam_offset (may be NULL):
 * [n_cams] first row of each camera in the elimination order. */
int xrsfm_ba_debug_chol_plan(const xrsfm_ba_problem *problem, int32_t stats[8], int32_t *cam_offset);

/* TEST ENTRY: device-side packing (large problems: xrsfm_ba_create sorts and lays out the observations on the GPU) against the host
 * packing it replaces: packs `problem` both ways and compares every array.  Returns 0 with *field = 0 when they are identical,
 * *field > 0 = the first array that differs (see xrsfm_ba.hip), *field = -100 when the device path declines the problem. */
int xrsfm_ba_debug_device_pack_check(const xrsfm_ba_problem *problem, int32_t *field, int32_t *index);

/* Multi-GPU emulation for tests: supply the union of all ranks' off-diagonal camera pairs (row > col) before the first
 * Cholesky solve, exactly what xrsfm_ba_run obtains with an all-reduce when n_ranks > 1. */
int xrsfm_ba_debug_set_block_pattern(xrsfm_ba_context *ctx, int n_pairs, const int32_t *row_col);

/* After debug_linearize: solve S(radius) y = b with the Cholesky path; y [n_cams][6].  If S_dense != NULL it
 * receives the assembled reduced camera matrix before factorisation, [6 n_cams][6 n_cams] row-major, symmetric. */
int xrsfm_ba_debug_cholesky_solve(xrsfm_ba_context *ctx, double radius, double *y, double *S_dense);

/* bal9 contexts: linearise at the current state with Jacobi scaling (cost; per observation r [n_obs][2], Jc [n_obs][2][9],
 * Jp [n_obs][2][3]; per camera diag(Hcc), g_c [n_cams][9]) and, if y != NULL, solve the reduced system at `radius`
 * (y [n_cams][9], scaled coordinates).  Any output pointer may be NULL. */
int xrsfm_ba_debug_wide(xrsfm_ba_context *ctx, double huber_a, double radius, double *cost, double *r, double *Jc, double *Jp,
                        double *Hcc_diag, double *gc, double *y);

/* After debug_cholesky_solve: one back-substitution (k_backsub) from the camera solution it left.  Outputs in the
 * library's PACKED order (for comparing two builds of the library on the same problem, tools/backsub_waves_probe.py):
 * per work item the model-decrease and squared point-step partials [n_items] (n_items = debug_pack stats), candidate
 * points and scaled point steps [n_points_packed][3], candidate cameras [n_cams][4] / [n_cams][3].  Any pointer may be NULL. */
int xrsfm_ba_debug_backsub(xrsfm_ba_context *ctx, double *part_model, double *part_step2, double *cand_points,
                           double *point_step, double *cand_cam_q, double *cand_cam_t);

#ifdef __cplusplus
}
#endif
#endif /* XRSFM_BA_H */
