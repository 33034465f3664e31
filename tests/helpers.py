"""Shared test helpers: synthetic problems in both the product and the oracle representation."""
import numpy as np

from oracle import ba_oracle as bo
from xrsfm_amd import capi, synth

FIELDS = capi.ProblemArrays.FIELDS


def make(n_cams, n_points, k_obs=4, seed=100, **kw):
    d = synth.make_problem(n_cams, n_points, k_obs, seed=seed, **kw)
    return {k: d[k] for k in FIELDS}


def to_oracle(arr) -> bo.Problem:
    return bo.Problem(**{k: np.array(arr[k], copy=True) for k in FIELDS})


def to_product(arr) -> capi.ProblemArrays:
    return capi.ProblemArrays(**{k: np.array(arr[k], copy=True) for k in FIELDS})


def with_models(arr, seed=0):
    """Give every camera its own intrinsics, cycling through the 5 reference camera models."""
    rng = np.random.default_rng(seed)
    n = arr["cam_q"].shape[0]
    f, cx, cy = 718.856, 607.1928, 185.27157
    model = (np.arange(n) % 5).astype(np.int32)
    prm = np.zeros((n, 8))
    for i, m in enumerate(model):
        k = rng.normal(0, 0.02)
        if m == 0: prm[i, :3] = (f / 2, cx, cy)            # 2f quirk: halve f so the scene stays in view
        elif m == 1: prm[i, :4] = (f / 2, f / 2 * 1.01, cx, cy)
        elif m == 2: prm[i, :4] = (f, cx, cy, k)
        elif m == 3: prm[i, :5] = (f, f * 1.01, cx, cy, k)
        else: prm[i, :8] = (f, f * 0.99, cx, cy, k, rng.normal(0, 0.005), rng.normal(0, 1e-3), rng.normal(0, 1e-3))
    out = dict(arr)
    out["cam_intr"] = np.arange(n, dtype=np.int32)
    out["intr_model"] = model
    out["intr_params"] = prm
    return out


def rel_err(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def make_pose_problem(n=200, seed=0, model=2, outlier_frac=0.1, noise=0.5):
    """One frame against fixed 3-D points: the input of the reference's pose refinement (pnp.cc:38-71).  Returns the flat
    problem dict (one camera, every point constant) with a perturbed initial pose."""
    rng = np.random.default_rng(seed)
    f, cx, cy = 718.856, 607.1928, 185.27157
    prm = {0: [f / 2, cx, cy], 1: [f / 2, f / 2, cx, cy], 2: [f, cx, cy, -0.01], 3: [f, f, cx, cy, 0.01],
           4: [f, f, cx, cy, 0.01, -0.005, 1e-4, -2e-4]}[model]
    intr = np.zeros((1, 8)); intr[0, :len(prm)] = prm
    axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
    ang = 0.4
    q_true = np.concatenate([np.sin(ang / 2) * axis, [np.cos(ang / 2)]])
    t_true = rng.normal(0, 1.0, 3)
    R = bo.rotation_from_quat(q_true[None])[0]
    # points in front of the camera: sample in the camera frame, move to the world frame
    pc = np.stack([rng.uniform(-8, 8, n), rng.uniform(-2.5, 2.5, n), rng.uniform(8, 40, n)], 1)
    pw = (pc - t_true) @ R          # R^T (pc - t)
    arr = dict(cam_q=q_true[None].copy(), cam_t=t_true[None].copy(), cam_const=np.zeros(1, np.uint8), cam_intr=np.zeros(1, np.int32),
               intr_model=np.array([model], np.int32), intr_params=intr, points=pw, point_const=np.ones(n, np.uint8),
               obs_cam=np.zeros(n, np.int32), obs_pt=np.arange(n, dtype=np.int32), obs_uv=np.zeros((n, 2)))
    r0, _ = bo.project(to_oracle(arr), want_jac=False)       # residual = uv_est - uv with uv = 0  ->  uv_est
    uv = r0 + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outlier_frac
    uv[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2))
    arr["obs_uv"] = uv
    # RANSAC-quality initial pose
    dq = np.concatenate([rng.normal(0, 0.01, 3), [1.0]]); dq /= np.linalg.norm(dq)
    x1, y1, z1, w1 = dq; x2, y2, z2, w2 = q_true
    arr["cam_q"] = np.array([[w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                              w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]])
    arr["cam_t"] = (t_true + rng.normal(0, 0.05, 3))[None]
    return arr


def relabel_points(arr, seed=0):
    """The same problem with the points (and so the residual blocks of every camera) in another order: the summation order of
    every per-camera sum changes, the mathematics does not.  Returns (problem, perm) with new point j = old point perm[j]."""
    rng = np.random.default_rng(seed)
    n = arr["points"].shape[0]
    perm = rng.permutation(n)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    out = dict(arr)
    out["points"] = np.ascontiguousarray(arr["points"][perm]); out["point_const"] = np.ascontiguousarray(arr["point_const"][perm])
    op = inv[arr["obs_pt"]].astype(np.int32)
    order = np.lexsort((op, arr["obs_cam"]))                      # frame-major like the reference (Appendix B)
    out["obs_cam"] = np.ascontiguousarray(arr["obs_cam"][order]); out["obs_pt"] = np.ascontiguousarray(op[order])
    out["obs_uv"] = np.ascontiguousarray(arr["obs_uv"][order])
    return out, perm


def gn_energy(arr_state, dq_tangent, dP):
    """Gauss-Newton energy 1/2 |J dx|^2 of a parameter difference dx = (camera tangent [Nc,6], points [Np,3]) at the state
    `arr_state` (robustified Jacobian of the oracle): how much of the objective's quadratic model separates two results."""
    pr = to_oracle(arr_state)
    cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    Jd = np.einsum("nij,nj->ni", np.asarray(Fc).reshape(-1, 2, 6), dq_tangent[pr.obs_cam]) \
        + np.einsum("nij,nj->ni", np.asarray(Ep).reshape(-1, 2, 3), dP[pr.obs_pt])
    return 0.5 * float((Jd ** 2).sum()), cost


# ------------------------------------------------------------------------------------------------ shaped problems
def make_tracks(n_cams, tracks, seed=0, noise=0.5, outlier_frac=0.02):
    """A problem whose tracks are given explicitly: tracks[j] = ascending camera indices that observe point j.  Cameras on the
    ring of synth.make_problem(mode="unordered") (all look at a scene around the origin, so any camera subset is a valid
    track); observations = projections of the ground-truth points + noise, frame-major order like the reference's problem
    construction (ba_solver.cc:598-601)."""
    n_pts = len(tracks)
    d = synth.make_problem(n_cams, n_pts, 2, seed=seed, mode="unordered", min_tri_angle_deg=0.0)
    rng = np.random.default_rng(77000 + seed)
    obs_cam = np.concatenate([np.asarray(t, np.int64) for t in tracks])
    obs_pt = np.concatenate([np.full(len(t), j, np.int64) for j, t in enumerate(tracks)])
    order = np.lexsort((obs_pt, obs_cam))
    obs_cam, obs_pt = obs_cam[order].astype(np.int32), obs_pt[order].astype(np.int32)
    intr = tuple(d["intr_params"][0, :4])
    uv, z = synth._project_simple_radial(d["gt_q"][obs_cam], d["gt_t"][obs_cam], d["gt_points"][obs_pt], intr)
    assert (z > 1.0).all()
    uv = uv + rng.normal(0, noise, uv.shape)
    out = rng.random(uv.shape[0]) < outlier_frac
    uv[out] += rng.uniform(-30, 30, (int(out.sum()), 2))
    arr = {k: d[k] for k in FIELDS}
    arr["obs_cam"], arr["obs_pt"], arr["obs_uv"] = obs_cam, obs_pt, np.ascontiguousarray(uv)
    return arr


def shape_group(base, C, T, ragged, rng, span=10):
    """Tracks of one tile under test + the group that closes it.  The tile: T tracks over the cameras [base, base + C), every
    camera used, at most 64 observations; `ragged` = the tracks see different subsets (a dense tile = every track sees all C).
    Then 16 tracks with the tuple (base+span-1 .. base+span+2): they fill a tile by themselves, so the packing starts them on
    a tile boundary (ba_pack.h: big_group_start) and the tracks before them keep their tile to themselves; the tuple also
    ties this group's cameras to the next group's.  Returns (tracks, cameras used up to)."""
    cams = np.arange(base, base + C)
    tracks = []
    if not ragged:
        assert T * C <= 64
        tracks = [cams.copy() for _ in range(T)]
    else:
        assert C >= 2 and 2 * T <= 64
        # lengths: as even as possible with sum <= 64, each >= 2 and <= C; then make sure every camera is used and not all tuples are equal
        lmax = min(C, 64 // T)
        ln = np.full(T, int(min(lmax, max(2, round(0.6 * C)))))
        while ln.sum() < C and (ln < lmax).any():
            i = int(np.argmin(ln)); ln[i] += 1
        assert ln.sum() <= 64 and ln.sum() >= C
        for attempt in range(200):
            # cover: deal the cameras to the tracks (no track beyond its length), then fill every track up with other cameras
            sets = [set() for _ in range(T)]
            for c in rng.permutation(cams):
                room = [q for q in range(T) if len(sets[q]) < ln[q]]
                sets[int(rng.choice(room))].add(int(c))
            for q in range(T):
                rest = [int(c) for c in cams if int(c) not in sets[q]]
                need = int(ln[q]) - len(sets[q])
                if need > 0: sets[q].update(int(c) for c in rng.choice(rest, need, replace=False))
            tr = [np.array(sorted(x)) for x in sets]
            if all(len(t) >= 2 for t in tr) and (len({tuple(t) for t in tr}) > 1 or C == 2):
                break
        else:
            raise ValueError(f"no ragged tile for C={C} T={T}")
        tracks = tr
    closer = np.arange(base + span - 1, base + span + 3)
    tracks += [closer.copy() for _ in range(16)]
    return tracks, base + span + 3


def shape_cells():
    """The (C, T, ragged) cells a one-tile shape can realise: C distinct cameras (2..10 Gram tiles; 11..40 per-pair tiles),
    T tracks, dense (every track sees all C: T*C <= 64) or ragged (>= 2 per track, every camera used, <= 64 observations)."""
    cells = []
    for C in list(range(2, 11)) + [11, 13, 17, 24, 32, 40]:
        for T in (1, 2, 3, 5, 16, 21):
            if T * C <= 64 and (T > 1 or C <= 64):
                cells.append((C, T, False))
            # ragged: T tracks of >= 2 cameras, all C cameras used: needs sum of lengths >= C with lengths <= min(C, 64/T)
            lmax = min(C, 64 // T)
            if T >= 2 and lmax >= 2 and T * lmax >= C and C >= 3:
                cells.append((C, T, True))
    return cells


def shape_problems(groups_per_problem=6, seed=0):
    """Problems that together contain every cell of shape_cells() as a tile of its own.  Yields (arr, cells_in_it)."""
    cells = shape_cells()
    rng = np.random.default_rng(4242 + seed)
    out = []
    i = 0
    while i < len(cells):
        shape_tracks, mine = [], []
        base = 5
        while i < len(cells) and len(mine) < groups_per_problem:
            C, T, ragged = cells[i]
            span = max(10, C + 1)
            tr, nxt = shape_group(base, C, T, ragged, rng, span=span)
            shape_tracks += tr; base = nxt - 3    # the closer's last three cameras open the next group's range
            mine.append(cells[i]); i += 1
        n_cams = base + 3
        # Support: every camera shares six 3-view points with the gauge cameras 0 and 1 (tuples (0,1,c) sort before every
        # shape track, so they never share a tile with one), then a fence group (1,2,3,4) x 16 that fills a tile by itself:
        # the first shape group starts on a tile boundary.  Without it a camera of a 40-camera one-track tile would have a
        # single observation and the problem would be determined by the LM damping alone.
        tracks = [np.array([0, 1, c]) for c in range(2, n_cams) for _ in range(6)]
        tracks += [np.arange(1, 5) for _ in range(16)]
        tracks += shape_tracks
        out.append((make_tracks(n_cams, tracks, seed=100 + len(out)), mine))
    return out


def reduced_system_oracle(arr, radius, use_scaling=True):
    """Dense reduced camera matrix S(radius) and right-hand side b of the oracle's linearisation at the state `arr` (Jacobi
    scaling from the column norms like iteration 0 of a solve): S = Hcc + Dc^2 - sum_tracks W Hpp^-1 W^T."""
    pr = to_oracle(arr)
    cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    ci, pi = pr.obs_cam, pr.obs_pt
    n_cams, n_pts = pr.cam_q.shape[0], pr.points.shape[0]
    if use_scaling:
        sc_c = 1 / (1 + np.sqrt(bo._scatter_add(n_cams, ci, np.sum(Fc * Fc, axis=1))))
        sc_p = 1 / (1 + np.sqrt(bo._scatter_add(n_pts, pi, np.sum(Ep * Ep, axis=1))))
        Fc = Fc * sc_c[ci][:, None, :]; Ep = Ep * sc_p[pi][:, None, :]
    lin = bo._Linearization(pr, rt, Fc, Ep)
    Dc2 = np.clip(np.einsum("nii->ni", lin.Hcc), 1e-6, 1e32) / radius
    Dp2 = np.clip(np.einsum("nii->ni", lin.Hpp), 1e-6, 1e32) / radius
    Hinv = np.linalg.inv(lin.Hpp + np.einsum("ni,ij->nij", Dp2, np.eye(3)))
    n = 6 * n_cams
    S = np.zeros((n, n))
    WH = np.einsum("nij,njk->nik", lin.W, Hinv[pi])
    for c in range(n_cams):
        S[6 * c:6 * c + 6, 6 * c:6 * c + 6] = lin.Hcc[c] + np.diag(Dc2[c])
    order = np.argsort(pi, kind="stable")
    ptr = np.searchsorted(pi[order], np.arange(n_pts + 1))
    for j in range(n_pts):
        ids = order[ptr[j]:ptr[j + 1]]
        Wj = lin.W[ids]                                   # [k,6,3]
        blk = np.einsum("aij,bkj->aibk", WH[ids], Wj)     # [k,6,k,6]
        cj = ci[ids]
        for x, ca in enumerate(cj):
            for y, cb in enumerate(cj):
                S[6 * ca:6 * ca + 6, 6 * cb:6 * cb + 6] -= blk[x, :, y, :]
    b = lin.gc - bo._scatter_add(n_cams, ci, np.einsum("nij,nj->ni", WH, lin.gp[pi]))
    return S, b


def make_bal9(n_cams=12, n_pts=600, k_obs=4, seed=5, **kw):
    """bal9 mode (BASELINE north_star "2x9 camera blocks"): every camera has its own intrinsics of the extension
    model 5 {f, k1, k2} (no principal point) and keeps them VARIABLE (cam_const bit 2): 9-wide camera blocks.  Same geometry and
    observations as make() (the KITTI principal point is subtracted from the observations), intrinsics start 1 % / 0.01 off."""
    return synth.to_bal9(make(n_cams, n_pts, k_obs, seed=seed, **kw), seed)
