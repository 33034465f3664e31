"""Deterministic synthetic "BAL-style" BA problems.

Shapes follow BASELINE.json configs 2 and 4: one shared SIMPLE_RADIAL camera
with the KITTI-00 intrinsics the reference hard-codes
(/root/reference/src/rec_kitti.cc:25), `k_obs` observations per point, 0.5 px
Gaussian noise, 2 % gross outliers, perturbed initial state; frames 0 and 1
play the role of ``map.init_id1/init_id2`` (their translation is held constant,
/root/reference/src/optimization/ba_solver.cc:611-614).

The arrays returned are exactly the flat SoA the C-ABI takes
(include/xrsfm_ba.h); observation order is frame-major like the reference's
problem construction (ba_solver.cc:598-601, 336-349).
"""
from __future__ import annotations

import numpy as np

KITTI_INTR = (718.856, 607.1928, 185.27157, 0.0)   # rec_kitti.cc:25 -> {f, cx, cy, k}
IMG_W, IMG_H = 1241.0, 376.0


def _rot_from_quat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    M = np.empty(q.shape[:-1] + (3, 3))
    M[..., 0, 0] = 1 - 2 * (y * y + z * z); M[..., 0, 1] = 2 * (x * y - w * z); M[..., 0, 2] = 2 * (x * z + w * y)
    M[..., 1, 0] = 2 * (x * y + w * z); M[..., 1, 1] = 1 - 2 * (x * x + z * z); M[..., 1, 2] = 2 * (y * z - w * x)
    M[..., 2, 0] = 2 * (x * z - w * y); M[..., 2, 1] = 2 * (y * z + w * x); M[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return M


def _quat_from_rot(R):
    """Rotation matrices [n,3,3] -> unit quaternions xyzw (w >= 0)."""
    n = R.shape[0]
    q = np.empty((n, 4))
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    for i in range(n):
        m = R[i]
        if tr[i] > 0:
            s = np.sqrt(tr[i] + 1.0) * 2
            q[i] = ((m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s)
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = np.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q[i] = (0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s)
        elif m[1, 1] > m[2, 2]:
            s = np.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q[i] = ((m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s)
        else:
            s = np.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q[i] = ((m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[q[:, 3] < 0] *= -1
    return q


def _quat_plus(q, d):
    n = np.linalg.norm(d, axis=-1)
    safe = np.where(n > 0, n, 1.0)
    s = np.where(n > 0, np.sin(safe) / safe, 0.0)
    av = d * s[..., None]; aw = np.where(n > 0, np.cos(n), 1.0)
    bv = q[..., :3]; bw = q[..., 3]
    out = np.empty_like(q)
    out[..., 3] = aw * bw - np.sum(av * bv, axis=-1)
    out[..., :3] = aw[..., None] * bv + bw[..., None] * av + np.cross(av, bv)
    return out


def _project_simple_radial(q, t, P, intr):
    f, cx, cy, k = intr
    Pc = np.einsum("nij,nj->ni", _rot_from_quat(q), P) + t
    z = Pc[:, 2]
    zs = np.where(np.abs(z) > 1e-12, z, 1e-12)
    xn = Pc[:, 0] / zs; yn = Pc[:, 1] / zs
    r2 = xn * xn + yn * yn
    u = f * (xn + xn * k * r2) + cx
    v = f * (yn + yn * k * r2) + cy
    return np.stack([u, v], axis=1), z


def make_problem(n_cams: int, n_points: int, k_obs: int = 4, seed: int = 0,
                 mode: str = "sequential", k_dist: float = 0.0, noise: float = 0.5,
                 outlier_frac: float = 0.02, perturb=(0.01, 0.05, 0.10),
                 min_tri_angle_deg: float = 2.0, dropout: float = 0.0, point_seed: int | None = None,
                 literal_appendix_d: bool = False, n_hubs: int = 0, hub_tracks: int = 0) -> dict:
    """Return a dict of flat arrays (keys = fields of ``xrsfm_ba_problem``) + ground truth.
    ``dropout`` > 0 removes each observation with that probability (at least two per point stay): ragged tracks with many
    distinct camera tuples, like a real reconstruction with missed detections.
    ``point_seed``: None = one random stream for everything (the historical problems).  An integer makes the cameras
    (ground truth AND perturbed initial state) a function of ``seed`` alone and draws the points, observations and their
    noise from a stream of their own: calls that differ only in ``point_seed`` are disjoint point shards of one larger
    problem over the same cameras (bench.py --scaling weak, one shard per rank).
    ``literal_appendix_d``: SURVEY.md Appendix D read literally — a radius-40 ring whatever the camera count (0.25-unit
    baselines at 1000 cameras), no triangulation-angle filter, and the perturbation added to t instead of the camera centre
    (config L0: reported once in profiles/, it is a much harder problem than the headline workload)."""
    assert n_cams >= k_obs >= 1
    # distant landmarks (config LP): the last n_far points of the problem, see below
    n_far = n_hubs * hub_tracks if (n_hubs >= k_obs and mode == "sequential") else 0
    assert n_far < n_points
    n_points_total, n_points = n_points, n_points - n_far
    rng = np.random.Generator(np.random.PCG64(seed))
    rng_cam0 = rng if point_seed is None else np.random.Generator(np.random.PCG64([seed, 999983]))
    intr = (KITTI_INTR[0], KITTI_INTR[1], KITTI_INTR[2], k_dist)
    # Closed loop with a constant inter-camera spacing of 2*pi*40/100 = 2.51 units (the spacing of
    # BASELINE.json config 2: 100 cameras on a radius-40 ring; a KITTI-like per-frame baseline), so the
    # ring grows with the camera count (radius 400 at 1k cameras) instead of the parallax shrinking;
    # below 100 cameras an open arc of the radius-40 ring with the same spacing.
    radius = 40.0 if literal_appendix_d else 40.0 * max(n_cams, 100) / 100.0
    ang = 2 * np.pi * np.arange(n_cams) / (n_cams if literal_appendix_d else max(n_cams, 100))
    if literal_appendix_d:
        min_tri_angle_deg = 0.0
    centre = np.stack([radius * np.cos(ang), rng.normal(0, 0.1, n_cams), radius * np.sin(ang)], axis=1)
    if mode == "sequential":
        fwd = np.stack([-np.sin(ang), np.zeros(n_cams), np.cos(ang)], axis=1)      # tangent
    elif mode == "unordered":
        fwd = -centre / np.linalg.norm(centre, axis=1, keepdims=True)              # look at origin
    else:
        raise ValueError(mode)
    yaw = np.deg2rad(rng.uniform(-2, 2, n_cams)); pitch = np.deg2rad(rng.uniform(-2, 2, n_cams))
    up = np.array([0.0, -1.0, 0.0])      # image y points down
    Rcw = np.empty((n_cams, 3, 3))
    for i in range(n_cams):
        zc = fwd[i] / np.linalg.norm(fwd[i])
        xc = np.cross(-up, zc); xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R0 = np.stack([xc, yc, zc], axis=0)          # rows = camera axes in world
        cy_, sy_ = np.cos(yaw[i]), np.sin(yaw[i]); cp_, sp_ = np.cos(pitch[i]), np.sin(pitch[i])
        Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        Rx = np.array([[1, 0, 0], [0, cp_, -sp_], [0, sp_, cp_]])
        Rcw[i] = Rx @ Ry @ R0
    q_gt = _quat_from_rot(Rcw)
    Rq = _rot_from_quat(q_gt)
    t_gt = -np.einsum("nij,nj->ni", Rq, centre)

    if point_seed is not None:
        rng = np.random.Generator(np.random.PCG64([seed, 1000003 + int(point_seed)]))
    # points: base camera c0, placed in the frustum of camera c0 + k_obs/2, seen by k_obs cameras
    P_gt = np.empty((n_points, 3))
    cams_of = np.empty((n_points, k_obs), dtype=np.int64)
    todo = np.arange(n_points)
    f, cx, cy, _ = intr
    rounds = 0
    while todo.size:
        rounds += 1
        if rounds > 200:
            raise RuntimeError("synthetic generator failed to place all points")
        m = todo.size
        if mode == "sequential":
            c0 = rng.integers(0, n_cams, m)
            cams = (c0[:, None] + np.arange(k_obs)[None, :]) % n_cams
        else:
            if 4 * k_obs > n_cams:        # dense visibility: sample without replacement
                cams = np.argsort(rng.random((m, n_cams)), axis=1)[:, :k_obs]
                bad_dup = np.zeros(m, bool)
            else:
                cams = rng.integers(0, n_cams, (m, k_obs))
                srt = np.sort(cams, axis=1)
                bad_dup = (np.diff(srt, axis=1) == 0).any(axis=1) if k_obs > 1 else np.zeros(m, bool)
        cmid = cams[:, k_obs // 2]
        depth = rng.uniform(5.0, 40.0, m)
        u = rng.uniform(0.05 * IMG_W, 0.95 * IMG_W, m); v = rng.uniform(0.05 * IMG_H, 0.95 * IMG_H, m)
        pc = np.stack([(u - cx) / f * depth, (v - cy) / f * depth, depth], axis=1)
        if mode == "unordered":
            # scene around the origin so that many ring cameras see it
            pw = rng.uniform(-8.0, 8.0, (m, 3))
        else:
            pw = np.einsum("nji,nj->ni", Rq[cmid], pc - t_gt[cmid])       # R^T (pc - t)
        ok = np.ones(m, bool)
        if mode == "unordered":
            ok &= ~bad_dup
        for j in range(k_obs):
            uv, z = _project_simple_radial(q_gt[cams[:, j]], t_gt[cams[:, j]], pw, intr)
            ok &= (z > 1.0) & (uv[:, 0] >= 0) & (uv[:, 0] < IMG_W) & (uv[:, 1] >= 0) & (uv[:, 1] < IMG_H)
        # keep only points the mapper would keep: max pairwise triangulation angle above the
        # GBA filter threshold (FilterPoints3d, /root/reference/src/geometry/track_processor.cc:321-332)
        if min_tri_angle_deg > 0 and k_obs > 1:
            rays = pw[:, None, :] - centre[cams]                       # [m,k,3]
            rays /= np.linalg.norm(rays, axis=2, keepdims=True)
            cosang = np.einsum("mik,mjk->mij", rays, rays).min(axis=(1, 2))
            ok &= cosang < np.cos(np.deg2rad(min_tri_angle_deg))
        P_gt[todo[ok]] = pw[ok]; cams_of[todo[ok]] = cams[ok]
        todo = todo[~ok]

    if n_far > 0:
        # ``n_hubs`` / ``hub_tracks``: DISTANT LANDMARKS.  n_hubs cameras evenly spaced along the loop ("hub" frames); every
        # window of k_obs consecutive hubs sees hub_tracks far points (0.6 .. 2.5 ring radii away: a skyline seen over an
        # arc of the loop).  A sequential map whose tracks span 4 frames determines its global shape only through a chain of
        # 1000 relative poses; these few long-baseline tracks tie the loop together (the weakest Gauss-Newton eigenvalue rises
        # by orders of magnitude) WITHOUT leaving the code path of the sequential configurations: the plan treats the hub
        # frames as the hub cameras of a band order (ba_plan.h: long-range pairs, <= max(24, N_c/16) hubs, eliminated last).
        rng_far = np.random.Generator(np.random.PCG64([seed, 7919]))
        hubs = (np.arange(n_hubs) * n_cams) // n_hubs
        P_far = np.empty((n_far, 3)); cams_far = np.empty((n_far, k_obs), dtype=np.int64)
        for g in range(n_hubs):
            hc = hubs[(g + np.arange(k_obs)) % n_hubs]
            got = 0
            tries = 0
            while got < hub_tracks:
                tries += 1
                if tries > 400:
                    raise RuntimeError("synthetic generator failed to place the distant landmarks")
                m = 4 * hub_tracks
                cm = hc[k_obs // 2 - 1] if k_obs > 1 else hc[0]
                depth = rng_far.uniform(0.6, 2.5, m) * radius
                u = rng_far.uniform(0.05 * IMG_W, 0.95 * IMG_W, m); v = rng_far.uniform(0.2 * IMG_H, 0.8 * IMG_H, m)
                pc = np.stack([(u - cx) / f * depth, (v - cy) / f * depth, depth], axis=1)
                pw = np.einsum("ji,nj->ni", Rq[cm], pc - t_gt[cm])
                ok = np.ones(m, bool)
                for j in range(k_obs):
                    qj = np.broadcast_to(q_gt[hc[j]], (m, 4)); tj = np.broadcast_to(t_gt[hc[j]], (m, 3))
                    uvj, z = _project_simple_radial(qj, tj, pw, intr)
                    ok &= (z > 1.0) & (uvj[:, 0] >= 0) & (uvj[:, 0] < IMG_W) & (uvj[:, 1] >= 0) & (uvj[:, 1] < IMG_H)
                rays = pw[:, None, :] - centre[hc][None, :, :]
                rays /= np.linalg.norm(rays, axis=2, keepdims=True)
                ok &= np.einsum("mik,mjk->mij", rays, rays).min(axis=(1, 2)) < np.cos(np.deg2rad(max(min_tri_angle_deg, 2.0)))
                idx = np.nonzero(ok)[0][:hub_tracks - got]
                P_far[g * hub_tracks + got:g * hub_tracks + got + idx.size] = pw[idx]
                cams_far[g * hub_tracks + got:g * hub_tracks + got + idx.size] = hc[None, :]
                got += idx.size
        P_gt = np.concatenate([P_gt, P_far]); cams_of = np.concatenate([cams_of, cams_far])
        n_points = n_points_total

    # observations, frame-major order
    obs_pt = np.repeat(np.arange(n_points), k_obs)
    obs_cam = cams_of.reshape(-1)
    if dropout > 0.0 and k_obs > 2:
        keep = rng.random((n_points, k_obs)) >= dropout
        short = keep.sum(1) < 2
        keep[short, 0] = True; keep[short, k_obs - 1] = True      # the widest baseline of the window
        keep = keep.reshape(-1)
        obs_pt, obs_cam = obs_pt[keep], obs_cam[keep]
    order = np.lexsort((obs_pt, obs_cam))
    obs_pt = obs_pt[order].astype(np.int32); obs_cam = obs_cam[order].astype(np.int32)
    uv, _ = _project_simple_radial(q_gt[obs_cam], t_gt[obs_cam], P_gt[obs_pt], intr)
    uv = uv + rng.normal(0, noise, uv.shape)
    n_obs = uv.shape[0]
    out = rng.random(n_obs) < outlier_frac
    uv[out] += rng.uniform(-30, 30, (int(out.sum()), 2))

    # perturbed initial state; frames 0/1 keep their translation (gauge)
    s_rot, s_t, s_p = perturb
    # rotation and camera CENTRE are perturbed (t = -R c follows); the gauge frames 0/1 stay exact
    drot = rng_cam0.normal(0, s_rot, (n_cams, 3)); drot[0:2] = 0.0
    dcen = rng_cam0.normal(0, s_t, (n_cams, 3)); dcen[0:2] = 0.0
    q0 = _quat_plus(q_gt, drot)
    t0 = -np.einsum("nij,nj->ni", _rot_from_quat(q0), centre + dcen)
    if literal_appendix_d:
        t0 = t_gt + dcen
    t0[0:2] = t_gt[0:2]; q0[0:2] = q_gt[0:2]
    P0 = P_gt + rng.normal(0, s_p, (n_points, 3))
    cam_const = np.zeros(n_cams, np.uint8); cam_const[0:2] = 2        # bit1: t constant
    return dict(
        cam_q=np.ascontiguousarray(q0), cam_t=np.ascontiguousarray(t0), cam_const=cam_const,
        cam_intr=np.zeros(n_cams, np.int32),
        intr_model=np.array([2], np.int32),
        intr_params=np.array([[intr[0], intr[1], intr[2], intr[3], 0, 0, 0, 0]], np.float64),
        points=np.ascontiguousarray(P0), point_const=np.zeros(n_points, np.uint8),
        obs_cam=obs_cam, obs_pt=obs_pt, obs_uv=np.ascontiguousarray(uv),
        gt_q=q_gt, gt_t=t_gt, gt_points=P_gt,
    )


CONFIGS = {
    # BASELINE.json configs 2 and 4
    "S": dict(n_cams=100, n_points=50_000, k_obs=4, seed=2),
    "L": dict(n_cams=1000, n_points=500_000, k_obs=4, seed=4),
    # shape of BASELINE.json config 3 (KITTI-00 key-frame global BA, SURVEY 8 estimate): sequential visibility
    "K": dict(n_cams=2000, n_points=1_000_000, k_obs=4, seed=3),
    # shape of BASELINE.json config 5 (1DSfM internet collection): cameras around a scene, random visibility ->
    # a dense reduced camera matrix, no regular tiles
    # a long sequential trajectory: 30 000 camera unknowns, still on the exact (band-ordered) Cholesky path
    "X": dict(n_cams=5000, n_points=1_000_000, k_obs=4, seed=6),
    # ragged tracks (windows of 8 frames, 35 % missed detections): ~15 000 distinct camera tuples, few regular tiles
    "R": dict(n_cams=1000, n_points=400_000, k_obs=8, seed=7, dropout=0.35),
    # the same beyond the dense limit (18 000 camera unknowns, unordered): implicit-Schur PCG path
    "V": dict(n_cams=3000, n_points=300_000, k_obs=5, seed=8, mode="unordered"),
    "U": dict(n_cams=500, n_points=100_000, k_obs=5, seed=5, mode="unordered"),
    # the dense limit of the exact path: 12 000 camera unknowns, unordered -> right-looking tile Cholesky of a full S
    # (the configuration the MFMA utilisation of the reduced solve is quoted on)
    "D": dict(n_cams=2000, n_points=200_000, k_obs=5, seed=9, mode="unordered"),
    # BASELINE.json config 4 as a PARITY workload: the same sizes and the same code path as L (band
    # order, level schedule, regular 4-camera tiles), plus 24 hub frames that share 50 distant landmarks per window of four hubs
    # (1200 of the 500 000 tracks).  They tie the 1000-frame loop together, so that absolute camera parameters are determined to
    # better than the north star's 1e-5 and two exact solvers can be compared on them literally (L itself: 1e-3 of gauge drift).
    "LP": dict(n_cams=1000, n_points=500_000, k_obs=4, seed=4, n_hubs=24, hub_tracks=50),
    # BASELINE.json config 4 with SURVEY.md Appendix D read literally (radius-40 ring, no angle filter)
    "L0": dict(n_cams=1000, n_points=500_000, k_obs=4, seed=4, literal_appendix_d=True),
}


def _next_prime(n: int) -> int:
    p = max(2, int(n))
    while any(p % q == 0 for q in range(2, int(p ** 0.5) + 1)):
        p += 1
    return p


def make_collection(n_cams: int, n_points: int, seed: int = 0, cams_per_cluster: int = 150, track_alpha: float = 1.45,
                    max_track: int = 80, cross_cluster: float = 0.15, shuffle_ids: bool = True, noise: float = 0.5,
                    outlier_frac: float = 0.02, perturb=(0.01, 0.05, 0.10), min_tri_angle_deg: float = 2.0) -> dict:
    """Synthetic *unordered photo collection* with viewpoint clusters (the shape of BASELINE.json config 5, 1DSfM Trafalgar:
    /root/reference/src/rec_1dsfm.cc:66-98 reconstructs ~7.5k internet photos of one plaza; the data set itself is not
    available offline, so this is a generator of that SIZE and SHAPE, not Trafalgar).
      * landmarks (clusters) on a ring, 25 units apart; the photos of a landmark are taken from 12-35 units away within
        +-60 degrees of its outward direction and look at it (+ jitter);
      * the points of a landmark lie on its "facade" (4 units wide, 5 high, 1 deep);
      * track lengths follow a power law (2 + Pareto(track_alpha), capped at max_track): most points are seen by 2-4 photos, a
        few by dozens; a track draws its photos from the point's landmark and, with probability `cross_cluster` per
        observation, from the two neighbouring landmarks, among the photos that really see the point (in frame, depth > 1);
      * camera ids are shuffled (an internet collection has no temporal order), so the reduced camera matrix has NO band: each
        landmark is a dense diagonal block, neighbouring landmarks are coupled by the cross-cluster observations.
    Same flat arrays, noise model, perturbation and gauge convention (frames 0 and 1 keep their translation) as make_problem()."""
    rng = np.random.Generator(np.random.PCG64([seed, 424243]))
    K = max(3, int(round(n_cams / cams_per_cluster)))
    intr = KITTI_INTR
    f, cx, cy, _ = intr
    a = 2 * np.pi * np.arange(K) / K
    R = 25.0 * K / (2 * np.pi)
    lm = np.stack([R * np.cos(a), np.zeros(K), R * np.sin(a)], axis=1)
    cl_of_cam = np.sort(rng.integers(0, K, n_cams))
    th = a[cl_of_cam] + np.deg2rad(rng.uniform(-60, 60, n_cams))
    rho = rng.uniform(12.0, 35.0, n_cams)
    centre = lm[cl_of_cam] + np.stack([rho * np.cos(th), rng.normal(0, 0.3, n_cams), rho * np.sin(th)], axis=1)
    target = lm[cl_of_cam] + rng.normal(0, 1.5, (n_cams, 3)) * np.array([1.0, 0.3, 1.0])
    fwd = target - centre
    fwd /= np.linalg.norm(fwd, axis=1, keepdims=True)
    up = np.array([0.0, -1.0, 0.0])
    xc = np.cross(-up[None, :], fwd); xc /= np.linalg.norm(xc, axis=1, keepdims=True)
    yc = np.cross(fwd, xc)
    Rcw = np.stack([xc, yc, fwd], axis=1)                 # rows = camera axes in world
    q_gt = _quat_from_rot(Rcw)
    Rq = _rot_from_quat(q_gt)
    t_gt = -np.einsum("nij,nj->ni", Rq, centre)
    cam_first = np.searchsorted(cl_of_cam, np.arange(K + 1))

    cl_of_pt = np.sort(rng.integers(0, K, n_points))
    pt_first = np.searchsorted(cl_of_pt, np.arange(K + 1))
    P_gt = np.empty((n_points, 3))
    want = np.minimum(max_track, 2 + np.floor(rng.pareto(track_alpha, n_points) * 1.6)).astype(np.int64)
    obs_cam_l, obs_pt_l = [], []
    cos_min = np.cos(np.deg2rad(min_tri_angle_deg))
    for k in range(K):
        p0, p1 = pt_first[k], pt_first[k + 1]
        if p1 == p0:
            continue
        near = np.concatenate([np.arange(cam_first[kk % K], cam_first[kk % K + 1]) for kk in (k - 1, k, k + 1)]) if K >= 3 else np.arange(n_cams)
        near = np.unique(near)
        own = cl_of_cam[near] == k
        tang = np.array([-np.sin(a[k]), 0.0, np.cos(a[k])]); outw = np.array([np.cos(a[k]), 0.0, np.sin(a[k])])
        # short tracks (the bulk) draw from 48 random candidate photos, long ones from every photo around the landmark
        allp = np.arange(p0, p1)
        for todo, ncand in ((allp[want[allp] <= 12], 48), (allp[want[allp] > 12], near.size)):
            ncand = min(ncand, near.size)
            for _round in range(80):
                if todo.size == 0:
                    break
                m = todo.size
                pw = lm[k] + rng.normal(0, 2.0, (m, 1)) * tang + rng.uniform(-2.5, 2.5, (m, 1)) * np.array([0, 1.0, 0]) + rng.normal(0, 0.5, (m, 1)) * outw
                if ncand < near.size:        # distinct candidates: an arithmetic progression modulo a prime >= near.size, out-of-range entries masked
                    start = rng.integers(0, near.size, (m, 1)); stride = rng.integers(1, near.size, (m, 1))
                    pos = (start + stride * np.arange(ncand)[None, :]) % _next_prime(near.size)
                    inr = pos < near.size
                    pos = np.where(inr, pos, 0)
                else:
                    pos = np.broadcast_to(np.arange(near.size)[None, :], (m, near.size)); inr = np.ones((m, near.size), bool)
                cid = near[pos]                                                   # [m, ncand]
                Pc = np.einsum("mcij,mj->mci", Rq[cid], pw) + t_gt[cid]
                z = Pc[..., 2]
                zs = np.where(z > 1e-9, z, 1e-9)
                u = f * Pc[..., 0] / zs + cx; v = f * Pc[..., 1] / zs + cy
                vis = inr & (z > 1.0) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
                # random keys: photos of the own landmark first, a photo of a neighbour is preferred with probability cross_cluster
                key = rng.random((m, ncand))
                key = np.where(own[pos] | (rng.random((m, ncand)) < cross_cluster), key, key + 1.0)
                key = np.where(vis, key, np.inf)
                Lw = want[todo]
                srt = np.argsort(key, axis=1)
                cid = np.take_along_axis(cid, srt, axis=1); kk = np.take_along_axis(key, srt, axis=1)
                take = (np.arange(ncand)[None, :] < Lw[:, None]) & np.isfinite(kk)
                ok = take.sum(1) >= 2
                if min_tri_angle_deg > 0:
                    c8 = cid[:, :8]                         # widest pair among the first (up to) 8 photos of the track
                    rays = pw[:, None, :] - centre[c8]
                    rays /= np.linalg.norm(rays, axis=2, keepdims=True)
                    cosang = np.einsum("mik,mjk->mij", rays, rays)
                    valid8 = take[:, :8]
                    cosang = np.where(valid8[:, :, None] & valid8[:, None, :], cosang, 1.0)
                    ok &= cosang.min(axis=(1, 2)) < cos_min
                good = np.nonzero(ok)[0]
                P_gt[todo[good]] = pw[good]
                rows, cols = np.nonzero(take[good])
                obs_pt_l.append(todo[good][rows]); obs_cam_l.append(cid[good][rows, cols])
                todo = todo[~ok]
            if todo.size:
                raise RuntimeError("make_collection: could not place all points of a landmark")
    obs_pt = np.concatenate(obs_pt_l); obs_cam = np.concatenate(obs_cam_l)
    if shuffle_ids:
        perm = rng.permutation(n_cams)            # new id of old camera c = perm[c]
        inv = np.empty(n_cams, np.int64); inv[perm] = np.arange(n_cams)
        obs_cam = perm[obs_cam]
        q_gt, t_gt, centre, cl_of_cam = q_gt[inv], t_gt[inv], centre[inv], cl_of_cam[inv]
    pperm = rng.permutation(n_points)             # points in no particular order either
    pinv = np.empty(n_points, np.int64); pinv[pperm] = np.arange(n_points)
    obs_pt = pperm[obs_pt]; P_gt = P_gt[pinv]
    order = np.lexsort((obs_pt, obs_cam))
    obs_pt = obs_pt[order].astype(np.int32); obs_cam = obs_cam[order].astype(np.int32)
    uv, _ = _project_simple_radial(q_gt[obs_cam], t_gt[obs_cam], P_gt[obs_pt], intr)
    uv = uv + rng.normal(0, noise, uv.shape)
    n_obs = uv.shape[0]
    out = rng.random(n_obs) < outlier_frac
    uv[out] += rng.uniform(-30, 30, (int(out.sum()), 2))
    s_rot, s_t, s_p = perturb
    drot = rng.normal(0, s_rot, (n_cams, 3)); drot[0:2] = 0.0
    dcen = rng.normal(0, s_t, (n_cams, 3)); dcen[0:2] = 0.0
    q0 = _quat_plus(q_gt, drot)
    t0 = -np.einsum("nij,nj->ni", _rot_from_quat(q0), centre + dcen)
    t0[0:2] = t_gt[0:2]; q0[0:2] = q_gt[0:2]
    P0 = P_gt + rng.normal(0, s_p, (n_points, 3))
    cam_const = np.zeros(n_cams, np.uint8); cam_const[0:2] = 2
    return dict(
        cam_q=np.ascontiguousarray(q0), cam_t=np.ascontiguousarray(t0), cam_const=cam_const,
        cam_intr=np.zeros(n_cams, np.int32), intr_model=np.array([2], np.int32),
        intr_params=np.array([[intr[0], intr[1], intr[2], intr[3], 0, 0, 0, 0]], np.float64),
        points=np.ascontiguousarray(P0), point_const=np.zeros(n_points, np.uint8),
        obs_cam=obs_cam, obs_pt=obs_pt, obs_uv=np.ascontiguousarray(uv),
        gt_q=q_gt, gt_t=t_gt, gt_points=P_gt, cluster_of_cam=cl_of_cam,
    )


# BASELINE.json config 5 at its size (docs/en/benchmark.md:93,111: ~7.5k registered frames; ~10M observations estimated)
CONFIGS["T"] = dict(n_cams=7500, n_points=1_800_000, seed=12)
# BASELINE.json config 4 in bal9 mode (SURVEY.md section 8(d) "optional bal9 mode"): to_bal9(make_problem(**CONFIGS["L"]))
CONFIGS["Lb9"] = CONFIGS["L"]


def to_bal9(arr: dict, seed: int = 0) -> dict:
    """The same geometry and observations with one intrinsics entry of the extension model 5 {f, k1, k2} per camera, kept
    VARIABLE (cam_const bit 2 of include/xrsfm_ba.h): 9-wide camera blocks.  The principal point of the pinhole model the
    problem was generated with is subtracted from the observations (model 5 has none); the intrinsics start 1 % / 0.01 off
    the generating ones, so they have somewhere to go."""
    out = dict(arr)
    n_cams = arr["cam_q"].shape[0]
    rng = np.random.default_rng(9000 + seed)
    prm0 = np.asarray(arr["intr_params"], np.float64)[0]
    f, cx, cy = prm0[0], prm0[1], prm0[2]
    out["obs_uv"] = np.ascontiguousarray(np.asarray(arr["obs_uv"], np.float64) - np.array([cx, cy]))
    out["cam_intr"] = np.arange(n_cams, dtype=np.int32)
    out["intr_model"] = np.full(n_cams, 5, np.int32)
    prm = np.zeros((n_cams, 8))
    prm[:, 0] = f * (1 + rng.normal(0, 0.01, n_cams)); prm[:, 1] = rng.normal(0, 0.01, n_cams); prm[:, 2] = rng.normal(0, 0.01, n_cams)
    out["intr_params"] = prm
    out["cam_const"] = (np.asarray(arr["cam_const"], np.uint8) | 4).astype(np.uint8)
    return out
