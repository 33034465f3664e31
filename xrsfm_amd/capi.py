"""ctypes binding of the C-ABI in include/xrsfm_ba.h (the product path: HIP only).

There is deliberately no CPU fallback here: if the shared library is missing or
no HIP device is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

_c_double_p = C.POINTER(C.c_double)
_c_int32_p = C.POINTER(C.c_int32)
_c_uint8_p = C.POINTER(C.c_uint8)


class CProblem(C.Structure):
    _fields_ = [
        ("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("n_intr", C.c_int32),
        ("cam_q", _c_double_p), ("cam_t", _c_double_p), ("cam_const", _c_uint8_p), ("cam_intr", _c_int32_p),
        ("intr_model", _c_int32_p), ("intr_params", _c_double_p),
        ("points", _c_double_p), ("point_const", _c_uint8_p),
        ("obs_cam", _c_int32_p), ("obs_pt", _c_int32_p), ("obs_uv", _c_double_p),
    ]


class CPgProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_scales", C.c_int32), ("n_edges", C.c_int32), ("n_scale_costs", C.c_int32),
                ("rot_q", _c_double_p), ("pos", _c_double_p), ("scale", _c_double_p), ("pos_const", _c_uint8_p),
                ("scale_const", _c_uint8_p), ("scale_lower", _c_double_p), ("edge_a", _c_int32_p), ("edge_b", _c_int32_p),
                ("edge_sa", _c_int32_p), ("edge_sb", _c_int32_p), ("edge_q_mea", _c_double_p), ("edge_p_mea", _c_double_p),
                ("weight_o", C.c_double), ("sc_a", _c_int32_p), ("sc_b", _c_int32_p), ("sc_s12", _c_double_p)]


class CPgOptions(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("initial_radius", C.c_double), ("verbose", C.c_int32),
                ("bounds_active_set", C.c_int32)]


class CPgSummary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32), ("n_successful", C.c_int32),
                ("n_unsuccessful", C.c_int32), ("termination", C.c_int32)]


class CTagProblem(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("frame_q", _c_double_p), ("frame_t", _c_double_p), ("n_tags", C.c_int32),
                ("tag_length", C.c_double), ("tag_corners", _c_double_p), ("tag_q", _c_double_p), ("tag_t", _c_double_p),
                ("scale", C.c_double), ("scale_lower", C.c_double), ("n_tag_obs", C.c_int32), ("tag_obs_tag", _c_int32_p),
                ("tag_obs_frame", _c_int32_p), ("tag_obs_xy", _c_double_p), ("n_points", C.c_int32), ("n_obs", C.c_int32),
                ("points", _c_double_p), ("obs_frame", _c_int32_p), ("obs_pt", _c_int32_p), ("obs_xy", _c_double_p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.c_int)


class COptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double), ("initial_radius", C.c_double), ("huber_a", C.c_double),
        ("linear_solver", C.c_int32), ("pcg_tolerance", C.c_double), ("pcg_max_iterations", C.c_int32),
        ("profile", C.c_int32), ("verbose", C.c_int32),
    ]


class CSummary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("num_residuals", C.c_int32), ("num_effective_params", C.c_int32),
        ("n_successful", C.c_int32), ("n_unsuccessful", C.c_int32),
        ("termination", C.c_int32), ("termination_reason", C.c_int32),
        ("pcg_iterations", C.c_int32), ("lm_steps_attempted", C.c_int32),
        ("total_time_s", C.c_double), ("dom_kernel_ms", C.c_double),
        ("dom_kernel_launches", C.c_int32), ("dom_kernel_id", C.c_int32),
        ("linear_solver_used", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


# every symbol include/xrsfm_ba.h declares
EXPORTS = [
    "xrsfm_ba_default_options", "xrsfm_ba_version", "xrsfm_ba_create", "xrsfm_ba_comm_unique_id",
    "xrsfm_ba_comm_init", "xrsfm_ba_run", "xrsfm_ba_reset", "xrsfm_ba_download", "xrsfm_ba_destroy",
    "xrsfm_ba_solve", "xrsfm_ba_filter_tracks", "xrsfm_ba_profile_entry", "xrsfm_ba_debug_linearize", "xrsfm_ba_debug_schur_product",
    "xrsfm_ba_debug_cholesky_solve", "xrsfm_ba_debug_set_block_pattern", "xrsfm_ba_debug_pack", "xrsfm_ba_debug_chol_plan", "xrsfm_ba_refine_pose", "xrsfm_ba_refine_pose_options", "xrsfm_ba_debug_comm_hook", "xrsfm_pg_default_options", "xrsfm_pg_solve", "xrsfm_ba_debug_pack_gram", "xrsfm_ba_debug_gram_schedule",
    "xrsfm_tag_default_options", "xrsfm_tag_refine", "xrsfm_ba_refine_poses", "xrsfm_ba_quiesce", "xrsfm_ba_debug_backsub", "xrsfm_ba_device_memory", "xrsfm_ba_download_intrinsics", "xrsfm_ba_debug_wide",
    "xrsfm_ba_debug_device_pack_check", "xrsfm_ba_warmup",
]

SOLVER_PCG, SOLVER_CHOLESKY, SOLVER_AUTO = 0, 1, 2

ERRORS = {-1: "EINVAL", -2: "ENODEV (no HIP device / HIP error; there is no CPU fallback)", -3: "ENOMEM",
          -4: "ECOMM", -5: "ESTATE", -6: "ETOOBIG", -7: "EINTERNAL (unexpected exception inside the library)"}

_lib = None


def load(path: str | None = None):
    """dlopen libxrsfm_ba.so.  torch (if used in this process) must be imported
    first so both share one HIP runtime (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("XRSFM_BA_LIB") or _build.LIB      # XRSFM_BA_LIB: developer override (ablation builds)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "the BA path has no fallback implementation")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.xrsfm_ba_default_options.argtypes = [C.POINTER(COptions)]; lib.xrsfm_ba_default_options.restype = None
    lib.xrsfm_ba_version.argtypes = [C.POINTER(C.c_int)]; lib.xrsfm_ba_version.restype = C.c_int
    lib.xrsfm_ba_create.argtypes = [C.POINTER(CProblem), C.c_int, C.POINTER(vp)]; lib.xrsfm_ba_create.restype = C.c_int
    lib.xrsfm_ba_comm_unique_id.argtypes = [C.c_char_p]; lib.xrsfm_ba_comm_unique_id.restype = C.c_int
    lib.xrsfm_ba_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]; lib.xrsfm_ba_comm_init.restype = C.c_int
    lib.xrsfm_ba_run.argtypes = [vp, C.POINTER(COptions), C.POINTER(CSummary)]; lib.xrsfm_ba_run.restype = C.c_int
    lib.xrsfm_ba_reset.argtypes = [vp]; lib.xrsfm_ba_reset.restype = C.c_int
    lib.xrsfm_ba_download.argtypes = [vp, _c_double_p, _c_double_p, _c_double_p]; lib.xrsfm_ba_download.restype = C.c_int
    lib.xrsfm_ba_destroy.argtypes = [vp]; lib.xrsfm_ba_destroy.restype = None
    lib.xrsfm_ba_solve.argtypes = [C.POINTER(COptions), C.POINTER(CProblem), C.POINTER(CSummary)]; lib.xrsfm_ba_solve.restype = C.c_int
    lib.xrsfm_ba_filter_tracks.argtypes = [C.POINTER(CProblem), C.c_double, C.c_double, _c_uint8_p, _c_uint8_p, _c_double_p,
                                            _c_double_p, _c_int32_p]
    lib.xrsfm_ba_filter_tracks.restype = C.c_int
    lib.xrsfm_ba_debug_linearize.argtypes = [vp, C.c_double, C.c_int] + [_c_double_p] * 8
    lib.xrsfm_ba_debug_linearize.restype = C.c_int
    lib.xrsfm_ba_debug_schur_product.argtypes = [vp, C.c_double, _c_double_p, _c_double_p, _c_double_p]
    lib.xrsfm_ba_debug_schur_product.restype = C.c_int
    lib.xrsfm_ba_debug_cholesky_solve.argtypes = [vp, C.c_double, _c_double_p, _c_double_p]
    lib.xrsfm_ba_debug_cholesky_solve.restype = C.c_int
    if hasattr(lib, "xrsfm_ba_debug_backsub"):      # (absent in the round-2 repro builds of tools/backsub_waves_probe.py)
        lib.xrsfm_ba_debug_backsub.argtypes = [vp] + [_c_double_p] * 6
        lib.xrsfm_ba_debug_backsub.restype = C.c_int
    lib.xrsfm_ba_debug_set_block_pattern.argtypes = [vp, C.c_int, _c_int32_p]
    lib.xrsfm_ba_debug_set_block_pattern.restype = C.c_int
    lib.xrsfm_pg_default_options.argtypes = [C.POINTER(CPgOptions)]
    lib.xrsfm_pg_default_options.restype = None
    lib.xrsfm_ba_refine_poses.argtypes = [C.POINTER(COptions), C.c_int32, _c_int32_p, _c_double_p, _c_int32_p, _c_double_p, _c_double_p,
                                          _c_uint8_p, _c_double_p, _c_double_p, C.POINTER(CSummary)]
    lib.xrsfm_ba_refine_poses.restype = C.c_int
    lib.xrsfm_pg_solve.argtypes = [C.POINTER(CPgOptions), C.POINTER(CPgProblem), C.POINTER(CPgSummary)]
    lib.xrsfm_pg_solve.restype = C.c_int
    lib.xrsfm_tag_default_options.argtypes = [C.POINTER(CPgOptions)]
    lib.xrsfm_tag_default_options.restype = None
    lib.xrsfm_tag_refine.argtypes = [C.POINTER(CPgOptions), C.POINTER(CTagProblem), C.c_int32, C.POINTER(CPgSummary)]
    lib.xrsfm_tag_refine.restype = C.c_int
    lib.xrsfm_ba_debug_comm_hook.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
    lib.xrsfm_ba_debug_comm_hook.restype = C.c_int
    lib.xrsfm_ba_refine_pose_options.argtypes = [C.POINTER(COptions)]
    lib.xrsfm_ba_refine_pose_options.restype = None
    lib.xrsfm_ba_refine_pose.argtypes = [C.POINTER(COptions), C.c_int32, _c_double_p, C.c_int32, _c_double_p, _c_double_p, _c_uint8_p,
                                         _c_double_p, _c_double_p, C.POINTER(CSummary)]
    lib.xrsfm_ba_refine_pose.restype = C.c_int
    lib.xrsfm_ba_debug_pack_gram.argtypes = [C.POINTER(CProblem), _c_int32_p, _c_int32_p, _c_uint8_p, _c_int32_p]
    lib.xrsfm_ba_debug_pack_gram.restype = C.c_int
    lib.xrsfm_ba_debug_chol_plan.argtypes = [C.POINTER(CProblem), _c_int32_p, _c_int32_p]
    lib.xrsfm_ba_debug_chol_plan.restype = C.c_int
    lib.xrsfm_ba_debug_pack.argtypes = [C.POINTER(CProblem), _c_int32_p, _c_int32_p]
    lib.xrsfm_ba_debug_pack.restype = C.c_int
    lib.xrsfm_ba_profile_entry.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), _c_double_p, C.POINTER(C.c_int)]
    lib.xrsfm_ba_profile_entry.restype = C.c_int
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed: {code} {ERRORS.get(code, '')}")


def _dp(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


def device_count() -> int:
    n = C.c_int(0)
    load().xrsfm_ba_version(C.byref(n))
    return n.value


def quiesce() -> int:
    """xrsfm_ba_quiesce: wait for deferred context releases, free the cached device blocks; returns the bytes that were cached."""
    n = C.c_uint64(0)
    lib = load()
    lib.xrsfm_ba_quiesce.argtypes = [C.POINTER(C.c_uint64)]
    lib.xrsfm_ba_quiesce.restype = C.c_int
    check(lib.xrsfm_ba_quiesce(C.byref(n)), "xrsfm_ba_quiesce")
    return int(n.value)


def default_options(**kw) -> COptions:
    o = COptions()
    load().xrsfm_ba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class ProblemArrays:
    """Owns contiguous numpy arrays of one BA call and the matching C struct."""
    FIELDS = ("cam_q", "cam_t", "cam_const", "cam_intr", "intr_model", "intr_params",
              "points", "point_const", "obs_cam", "obs_pt", "obs_uv")

    def __init__(self, **arr):
        f64 = lambda a, s: np.ascontiguousarray(np.asarray(a, np.float64).reshape(s))
        self.cam_q = f64(arr["cam_q"], (-1, 4)); self.cam_t = f64(arr["cam_t"], (-1, 3))
        n_c = self.cam_q.shape[0]
        cc = arr.get("cam_const")
        self.cam_const = np.ascontiguousarray(np.zeros(n_c, np.uint8) if cc is None else np.asarray(cc, np.uint8))
        self.cam_intr = np.ascontiguousarray(np.asarray(arr["cam_intr"], np.int32))
        self.intr_model = np.ascontiguousarray(np.asarray(arr["intr_model"], np.int32))
        self.intr_params = f64(arr["intr_params"], (-1, 8))
        self.points = f64(arr["points"], (-1, 3))
        pc = arr.get("point_const")
        self.point_const = np.ascontiguousarray(np.zeros(self.points.shape[0], np.uint8) if pc is None else np.asarray(pc, np.uint8))
        self.obs_cam = np.ascontiguousarray(np.asarray(arr["obs_cam"], np.int32))
        self.obs_pt = np.ascontiguousarray(np.asarray(arr["obs_pt"], np.int32))
        self.obs_uv = f64(arr["obs_uv"], (-1, 2))
        if not (self.cam_t.shape[0] == n_c and self.cam_intr.shape[0] == n_c and self.cam_const.shape[0] == n_c):
            raise ValueError("cam_q / cam_t / cam_intr / cam_const disagree on the number of cameras")
        if not (self.obs_pt.shape[0] == self.obs_cam.shape[0] == self.obs_uv.shape[0]):
            raise ValueError("obs_cam / obs_pt / obs_uv disagree on the number of observations")
        if self.point_const.shape[0] != self.points.shape[0] or self.intr_model.shape[0] != self.intr_params.shape[0]:
            raise ValueError("point_const / points or intr_model / intr_params disagree in length")

    @property
    def n_cams(self): return self.cam_q.shape[0]
    @property
    def n_points(self): return self.points.shape[0]
    @property
    def n_obs(self): return self.obs_cam.shape[0]

    def as_dict(self):
        return {k: getattr(self, k) for k in self.FIELDS}

    def c_struct(self) -> CProblem:
        i32 = lambda a: a.ctypes.data_as(_c_int32_p)
        u8 = lambda a: a.ctypes.data_as(_c_uint8_p)
        return CProblem(self.n_cams, self.n_points, self.n_obs, self.intr_model.shape[0],
                        _dp(self.cam_q), _dp(self.cam_t), u8(self.cam_const), i32(self.cam_intr),
                        i32(self.intr_model), _dp(self.intr_params), _dp(self.points), u8(self.point_const),
                        i32(self.obs_cam), i32(self.obs_pt), _dp(self.obs_uv))


class Context:
    """Device-resident BA problem (xrsfm_ba_create ... xrsfm_ba_destroy)."""

    def __init__(self, problem: ProblemArrays, device: int = 0):
        self.lib = load()
        self.problem = problem
        self._h = C.c_void_p()
        cs = problem.c_struct()
        check(self.lib.xrsfm_ba_create(C.byref(cs), device, C.byref(self._h)), "xrsfm_ba_create")

    def close(self):
        if self._h:
            self.lib.xrsfm_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def comm_init(self, n_ranks: int, rank: int, unique_id: bytes):
        check(self.lib.xrsfm_ba_comm_init(self._h, n_ranks, rank, unique_id), "xrsfm_ba_comm_init")

    def comm_hook(self, n_ranks: int, rank: int, allreduce):
        """TEST transport: `allreduce(array, op)` reduces a host float64 array in place across the ranks (op 0 sum, 1 max)."""
        def _cb(user, buf, n, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(n),)), int(op))
                return 0
            except Exception:      # noqa: BLE001  (must not unwind through the C frame)
                return 1
        self._hook = ALLREDUCE_FN(_cb)       # keep the trampoline alive as long as the context
        check(self.lib.xrsfm_ba_debug_comm_hook(self._h, n_ranks, rank, self._hook, None), "xrsfm_ba_debug_comm_hook")

    def run(self, options: COptions | None = None) -> CSummary:
        options = options or default_options()
        s = CSummary()
        check(self.lib.xrsfm_ba_run(self._h, C.byref(options), C.byref(s)), "xrsfm_ba_run")
        return s

    def reset(self):
        check(self.lib.xrsfm_ba_reset(self._h), "xrsfm_ba_reset")

    def download(self, out=None):
        """out: optional (cam_q, cam_t, points) arrays to write into (the C++ adapter writes into the map's own storage; fresh
        numpy arrays of tens of MB cost their first-touch page faults on top of the copy)."""
        p = self.problem
        if out is None:
            q = np.empty_like(p.cam_q); t = np.empty_like(p.cam_t); P = p.points.copy()
        else:
            q, t, P = out
            # (not `assert`: stripped under python -O, and the C function writes through these pointers unchecked)
            if not (q.shape == p.cam_q.shape and t.shape == p.cam_t.shape and P.shape == p.points.shape):
                raise ValueError("download(out=...): shapes must equal the problem's cam_q / cam_t / points")
            if not all(a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable for a in (q, t, P)):
                raise ValueError("download(out=...): arrays must be writable C-contiguous float64")
            if P is not p.points:
                P[...] = p.points      # points that take no part in the problem keep their input value
        check(self.lib.xrsfm_ba_download(self._h, _dp(q), _dp(t), _dp(P)), "xrsfm_ba_download")
        return q, t, P

    def debug_linearize(self, huber_a: float = 5.99, use_scaling: bool = False):
        p = self.problem
        out = dict(r=np.zeros((p.n_obs, 2)), Jc=np.zeros((p.n_obs, 2, 6)), Jp=np.zeros((p.n_obs, 2, 3)),
                   Hpp=np.zeros((p.n_points, 6)), gp=np.zeros((p.n_points, 3)),
                   Hcc_diag=np.zeros((p.n_cams, 6)), gc=np.zeros((p.n_cams, 6)))
        cost = C.c_double(0)
        check(self.lib.xrsfm_ba_debug_linearize(self._h, huber_a, int(use_scaling), _dp(out["r"]), _dp(out["Jc"]),
                                                _dp(out["Jp"]), _dp(out["Hpp"]), _dp(out["gp"]), _dp(out["Hcc_diag"]),
                                                _dp(out["gc"]), C.cast(C.byref(cost), _c_double_p)), "debug_linearize")
        out["cost"] = cost.value
        return out

    def debug_schur_product(self, radius: float, x: np.ndarray):
        x = np.ascontiguousarray(x, np.float64)
        y = np.zeros_like(x); b = np.zeros_like(x)
        check(self.lib.xrsfm_ba_debug_schur_product(self._h, radius, _dp(x), _dp(y), _dp(b)), "debug_schur_product")
        return y, b


    def debug_cholesky_solve(self, radius: float, want_S: bool = False):
        n = 6 * self.problem.n_cams
        y = np.zeros((self.problem.n_cams, 6))
        S = np.zeros((n, n)) if want_S else None
        check(self.lib.xrsfm_ba_debug_cholesky_solve(self._h, radius, _dp(y), _dp(S)), "debug_cholesky_solve")
        return y, S

    def download_intrinsics(self) -> np.ndarray:
        """bal9 mode: intr_params with the refined {f, k1, k2} of the cameras that keep their intrinsics variable."""
        out = np.array(self.problem.intr_params, copy=True)
        self.lib.xrsfm_ba_download_intrinsics.argtypes = [C.c_void_p, _c_double_p]
        self.lib.xrsfm_ba_download_intrinsics.restype = C.c_int
        check(self.lib.xrsfm_ba_download_intrinsics(self._h, _dp(out)), "xrsfm_ba_download_intrinsics")
        return out

    def debug_wide(self, huber_a: float = 5.99, radius: float | None = None) -> dict:
        """bal9 contexts: scaled linearisation (and, with a radius, the reduced-system step) in caller order."""
        p = self.problem
        out = dict(r=np.zeros((p.n_obs, 2)), Jc=np.zeros((p.n_obs, 2, 9)), Jp=np.zeros((p.n_obs, 2, 3)), Hcc_diag=np.zeros((p.n_cams, 9)),
                   gc=np.zeros((p.n_cams, 9)))
        y = np.zeros((p.n_cams, 9)) if radius is not None else None
        cost = C.c_double(0)
        self.lib.xrsfm_ba_debug_wide.argtypes = [C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_double)] + [_c_double_p] * 6
        self.lib.xrsfm_ba_debug_wide.restype = C.c_int
        check(self.lib.xrsfm_ba_debug_wide(self._h, huber_a, float(radius or 1.0), C.byref(cost), _dp(out["r"]), _dp(out["Jc"]), _dp(out["Jp"]),
                                          _dp(out["Hcc_diag"]), _dp(out["gc"]), _dp(y)), "xrsfm_ba_debug_wide")
        out["cost"] = cost.value
        if y is not None:
            out["y"] = y
        return out

    def debug_backsub(self) -> dict:
        """After debug_cholesky_solve: one k_backsub launch; per-item partials and the candidate state in PACKED order."""
        st = debug_pack(self.problem)
        ni, npk, nc = st["items"], st["active_points"], self.problem.n_cams
        out = dict(part_model=np.zeros(ni), part_step2=np.zeros(ni), cand_points=np.zeros((npk, 3)), point_step=np.zeros((npk, 3)),
                   cand_cam_q=np.zeros((nc, 4)), cand_cam_t=np.zeros((nc, 3)))
        check(self.lib.xrsfm_ba_debug_backsub(self._h, *[_dp(out[k]) for k in ("part_model", "part_step2", "cand_points", "point_step",
                                                                                  "cand_cam_q", "cand_cam_t")]), "debug_backsub")
        return out

    def debug_set_block_pattern(self, row_col: np.ndarray):
        rc = np.ascontiguousarray(row_col, np.int32).reshape(-1, 2)
        check(self.lib.xrsfm_ba_debug_set_block_pattern(self._h, rc.shape[0], rc.ctypes.data_as(_c_int32_p)), "debug_set_block_pattern")

    def profile(self) -> dict:
        """Per-kernel HIP-event totals of the last run with options.profile != 0: name -> (ms, launches)."""
        out = {}
        i = 0
        while True:
            name = C.c_char_p(); ms = C.c_double(0); n = C.c_int(0)
            if self.lib.xrsfm_ba_profile_entry(self._h, i, C.byref(name), C.cast(C.byref(ms), _c_double_p), C.byref(n)) != 0:
                break
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(load().xrsfm_ba_comm_unique_id(buf), "xrsfm_ba_comm_unique_id")
    return buf.raw


def solve(problem: ProblemArrays, options: COptions | None = None) -> CSummary:
    """One-shot xrsfm_ba_solve: results are written into problem.cam_q / cam_t / points."""
    options = options or default_options()
    s = CSummary()
    cs = problem.c_struct()
    check(load().xrsfm_ba_solve(C.byref(options), C.byref(cs), C.byref(s)), "xrsfm_ba_solve")
    return s


def filter_tracks(problem: ProblemArrays, max_reproj_error: float, min_tri_angle_rad: float) -> dict:
    """xrsfm_ba_filter_tracks: masks and per-track statistics of the reference's FilterPoints3d."""
    obs_del = np.zeros(problem.n_obs, np.uint8); out = np.zeros(problem.n_points, np.uint8)
    err = np.zeros(problem.n_points); ang = np.zeros(problem.n_points); cnt = np.zeros(2, np.int32)
    cs = problem.c_struct()
    check(load().xrsfm_ba_filter_tracks(C.byref(cs), max_reproj_error, min_tri_angle_rad, obs_del.ctypes.data_as(_c_uint8_p),
                                        out.ctypes.data_as(_c_uint8_p), _dp(err), _dp(ang), cnt.ctypes.data_as(_c_int32_p)),
          "xrsfm_ba_filter_tracks")
    return dict(obs_delete=obs_del, track_outlier=out, track_error=err, track_angle=ang, num_filtered=cnt)


def debug_pack(problem: ProblemArrays) -> dict:
    """Host-side packing statistics and the slot -> observation map (works without a GPU)."""
    stats = np.zeros(8, np.int32)
    slot_obs = np.full(problem.n_obs + 64 * (problem.n_points + 1), -2, np.int32)
    cs = problem.c_struct()
    check(load().xrsfm_ba_debug_pack(C.byref(cs), stats.ctypes.data_as(_c_int32_p), slot_obs.ctypes.data_as(_c_int32_p)), "xrsfm_ba_debug_pack")
    keys = ("tiles", "slots", "items", "regular_tiles", "long_items", "cam_entries", "longest_track", "active_points")
    out = dict(zip(keys, (int(v) for v in stats)))
    out["slot_obs"] = slot_obs[:out["slots"]].copy()
    return out


def debug_device_pack_check(problem: ProblemArrays) -> tuple:
    """Device-side packing (large problems) against the host packing: (field, index) of the first difference, (0, -1) when every
    array is identical, (-100, -1) when the device path declines the problem.  Needs a GPU."""
    lib = load()
    lib.xrsfm_ba_debug_device_pack_check.argtypes = [C.POINTER(CProblem), _c_int32_p, _c_int32_p]
    lib.xrsfm_ba_debug_device_pack_check.restype = C.c_int
    f = np.zeros(1, np.int32); i = np.zeros(1, np.int32)
    cs = problem.c_struct()
    check(lib.xrsfm_ba_debug_device_pack_check(C.byref(cs), f.ctypes.data_as(_c_int32_p), i.ctypes.data_as(_c_int32_p)), "xrsfm_ba_debug_device_pack_check")
    return int(f[0]), int(i[0])


def debug_chol_plan(problem: ProblemArrays) -> dict:
    """Host-side plan of the Cholesky path: tiles, elimination-tree levels, ordering (works without a GPU)."""
    stats = np.zeros(8, np.int32)
    off = np.zeros(max(problem.n_cams, 1), np.int32)
    cs = problem.c_struct()
    check(load().xrsfm_ba_debug_chol_plan(C.byref(cs), stats.ctypes.data_as(_c_int32_p), off.ctypes.data_as(_c_int32_p)), "xrsfm_ba_debug_chol_plan")
    keys = ("tiles", "levels", "ordering", "hubs", "band", "blocks", "level_schedule", "tiles_nz")
    out = dict(zip(keys, (int(v) for v in stats)))
    out["lookahead"] = (out["level_schedule"] >> 1) & 1      # panel schedule with partial products on a second stream (ba_plan.h)
    out["level_schedule"] &= 1
    out["cam_offset"] = off[:problem.n_cams].copy()
    return out


def refine_pose_options(**overrides) -> "COptions":
    """Settings of the reference's pose refinement (pnp.cc:57-60): Ceres defaults, 10 iterations."""
    o = COptions()
    load().xrsfm_ba_refine_pose_options(C.byref(o))
    for k, v in overrides.items():
        setattr(o, k, v)
    return o


def refine_pose(model: int, intr_params, points3d, uv, q, t, inlier_mask=None, options=None):
    """xrsfm_ba_refine_pose: returns (q, t, summary); inputs are not modified."""
    prm = np.zeros(8); prm[:len(intr_params)] = np.asarray(intr_params, float)[:8]
    P = np.ascontiguousarray(points3d, float).reshape(-1, 3); UV = np.ascontiguousarray(uv, float).reshape(-1, 2)
    q = np.array(q, float).reshape(4).copy(); t = np.array(t, float).reshape(3).copy()
    mask = None if inlier_mask is None else np.ascontiguousarray(inlier_mask, np.uint8)
    s = CSummary()
    check(load().xrsfm_ba_refine_pose(C.byref(options) if options is not None else None, int(model), _dp(prm), P.shape[0], _dp(P), _dp(UV),
                                      mask.ctypes.data_as(_c_uint8_p) if mask is not None else None, _dp(q), _dp(t), C.byref(s)),
          "xrsfm_ba_refine_pose")
    return q, t, s


def refine_poses(models, intr_params, frames, q, t, options=None):
    """xrsfm_ba_refine_poses: `frames` is a list of (points3d [n,3], uv [n,2], inlier_mask or None) per frame; models
    [n_frames], intr_params [n_frames][<=8], q [n_frames][4], t [n_frames][3].  Returns (q, t, [summary per frame])."""
    nf = len(frames)
    prm = np.zeros((nf, 8))
    for f in range(nf):
        v = np.asarray(intr_params[f], float)[:8]; prm[f, :len(v)] = v
    corr_ptr = np.zeros(nf + 1, np.int32)
    for f in range(nf):
        corr_ptr[f + 1] = corr_ptr[f] + np.asarray(frames[f][0]).reshape(-1, 3).shape[0]
    P = np.ascontiguousarray(np.vstack([np.asarray(fr[0], float).reshape(-1, 3) for fr in frames]) if nf else np.zeros((0, 3)))
    UV = np.ascontiguousarray(np.vstack([np.asarray(fr[1], float).reshape(-1, 2) for fr in frames]) if nf else np.zeros((0, 2)))
    any_mask = any(fr[2] is not None for fr in frames)
    mask = None
    if any_mask:
        mask = np.ascontiguousarray(np.concatenate([np.ones(np.asarray(fr[0]).reshape(-1, 3).shape[0], np.uint8) if fr[2] is None
                                                    else np.asarray(fr[2], np.uint8) for fr in frames]))
    q = np.array(q, float).reshape(nf, 4).copy(); t = np.array(t, float).reshape(nf, 3).copy()
    mdl = np.ascontiguousarray(models, np.int32)
    sums = (CSummary * max(nf, 1))()
    check(load().xrsfm_ba_refine_poses(C.byref(options) if options is not None else None, nf, mdl.ctypes.data_as(_c_int32_p), _dp(prm),
                                       corr_ptr.ctypes.data_as(_c_int32_p), _dp(P), _dp(UV),
                                       mask.ctypes.data_as(_c_uint8_p) if mask is not None else None, _dp(q), _dp(t), sums),
          "xrsfm_ba_refine_poses")
    return q, t, [sums[f] for f in range(nf)]


def pose_graph_solve(rot_q, pos, scale, edges, weight_o=0.0, scale_costs=(), pos_const=None, scale_const=None, scale_lower=None,
                     **opt_overrides):
    """xrsfm_pg_solve (host code).  edges: dict(a, b, sa, sb, q_mea [n,4], p_mea [n,3]); scale_costs:
    iterable of (sa, sb, s12).  Returns (pos, scale, summary); inputs are not modified."""
    rot_q = np.ascontiguousarray(rot_q, float); pos = np.array(pos, float, copy=True); scale = np.array(scale, float, copy=True)
    keep = [rot_q, pos, scale]
    p = CPgProblem()
    p.n_frames, p.n_scales = rot_q.shape[0], scale.shape[0]
    p.rot_q, p.pos, p.scale = _dp(rot_q), _dp(pos), _dp(scale)

    def u8(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, np.uint8); keep.append(a); return a.ctypes.data_as(_c_uint8_p)

    def i32(a):
        a = np.ascontiguousarray(a, np.int32); keep.append(a); return a.ctypes.data_as(_c_int32_p)

    def f64(a):
        if a is None:
            return None
        a = np.ascontiguousarray(a, float); keep.append(a); return _dp(a)

    p.pos_const, p.scale_const, p.scale_lower = u8(pos_const), u8(scale_const), f64(scale_lower)
    p.n_edges = len(edges["a"])
    p.edge_a, p.edge_b, p.edge_sa, p.edge_sb = i32(edges["a"]), i32(edges["b"]), i32(edges["sa"]), i32(edges["sb"])
    p.edge_q_mea, p.edge_p_mea = f64(edges["q_mea"]), f64(edges["p_mea"])
    p.weight_o = float(weight_o)
    sc = list(scale_costs)
    p.n_scale_costs = len(sc)
    p.sc_a, p.sc_b, p.sc_s12 = i32([c[0] for c in sc]), i32([c[1] for c in sc]), f64([c[2] for c in sc])
    o = CPgOptions()
    load().xrsfm_pg_default_options(C.byref(o))
    for k, v in opt_overrides.items():
        setattr(o, k, v)
    s = CPgSummary()
    check(load().xrsfm_pg_solve(C.byref(o), C.byref(p), C.byref(s)), "xrsfm_pg_solve")
    return pos, scale, s


def tag_refine(frame_q, frame_t, tag_corners, tag_obs_tag, tag_obs_frame, tag_obs_xy, tag_length, points=None, obs_frame=None,
               obs_pt=None, obs_xy=None, stages=2, tag_q=None, tag_t=None, scale=1.0, scale_lower=0.2, **opt_overrides):
    """xrsfm_tag_refine (host code): the two solves of tag_refine (tag_extract.hpp:193-265).
    Returns dict(scale, tag_q, tag_t, tag_corners, points, summaries); inputs are not modified."""
    keep = []

    def f64(a, copy=False):
        a = np.array(a, float, copy=True) if copy else np.ascontiguousarray(a, float)
        keep.append(a); return a

    def i32(a):
        a = np.ascontiguousarray(a, np.int32); keep.append(a); return a

    frame_q, frame_t = f64(frame_q), f64(frame_t)
    corners = f64(tag_corners, copy=True).reshape(-1, 4, 3)
    n_tags = corners.shape[0]
    tq = f64(np.tile([0.0, 0.0, 0.0, 1.0], (n_tags, 1)) if tag_q is None else tag_q, copy=True)
    tt = f64(np.zeros((n_tags, 3)) if tag_t is None else tag_t, copy=True)
    p = CTagProblem()
    p.n_frames, p.frame_q, p.frame_t = frame_q.shape[0], _dp(frame_q), _dp(frame_t)
    p.n_tags, p.tag_length, p.tag_corners, p.tag_q, p.tag_t = n_tags, float(tag_length), _dp(corners), _dp(tq), _dp(tt)
    p.scale, p.scale_lower = float(scale), float(scale_lower)
    ot, of, oxy = i32(tag_obs_tag), i32(tag_obs_frame), f64(tag_obs_xy)
    p.n_tag_obs, p.tag_obs_tag, p.tag_obs_frame, p.tag_obs_xy = ot.shape[0], ot.ctypes.data_as(_c_int32_p), of.ctypes.data_as(_c_int32_p), _dp(oxy)
    pts = None
    if points is not None:
        pts = f64(points, copy=True)
        a, b, c = i32(obs_frame), i32(obs_pt), f64(obs_xy)
        p.n_points, p.n_obs, p.points = pts.shape[0], a.shape[0], _dp(pts)
        p.obs_frame, p.obs_pt, p.obs_xy = a.ctypes.data_as(_c_int32_p), b.ctypes.data_as(_c_int32_p), _dp(c)
    o = CPgOptions()
    load().xrsfm_tag_default_options(C.byref(o))
    for k, v in opt_overrides.items():
        setattr(o, k, v)
    sums = (CPgSummary * 2)()
    check(load().xrsfm_tag_refine(C.byref(o), C.byref(p), int(stages), sums), "xrsfm_tag_refine")
    return dict(scale=p.scale, tag_q=tq, tag_t=tt, tag_corners=corners, points=pts, summaries=[sums[i] for i in range(stages)])


def debug_gram_schedule(n_cams: int) -> dict:
    """Schedule of 4x4 result blocks of a Gram tile of n_cams cameras (works without a GPU): instructions used by the 4x4 form (0 = the tile
    takes 16x16 result tiles), instructions of the full schedule, and its (row group, column group) blocks, four per instruction."""
    n = C.c_int32(0); na = C.c_int32(0)
    ent = np.zeros(128, np.uint16)
    f = load().xrsfm_ba_debug_gram_schedule
    f.argtypes = [C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint16)]
    f.restype = C.c_int
    check(f(int(n_cams), C.byref(n), C.byref(na), ent.ctypes.data_as(C.POINTER(C.c_uint16))), "xrsfm_ba_debug_gram_schedule")
    e = ent[:4 * na.value]
    return dict(n_inst=n.value, n_inst_all=na.value, blocks=[(int(v & 255), int(v >> 8)) for v in e])


def debug_pack_gram(problem: ProblemArrays) -> dict:
    """Gram tiles and S-assembly item classes of the host-side packing (works without a GPU)."""
    st = debug_pack(problem)
    stats = np.zeros(8, np.int32)
    ncam = np.zeros(max(st["tiles"], 1), np.int32); cidx = np.zeros(max(st["slots"], 1), np.uint8); cpg = np.zeros(max(st["slots"], 1), np.int32)
    cs = problem.c_struct()
    check(load().xrsfm_ba_debug_pack_gram(C.byref(cs), stats.ctypes.data_as(_c_int32_p), ncam.ctypes.data_as(_c_int32_p),
                                          cidx.ctypes.data_as(_c_uint8_p), cpg.ctypes.data_as(_c_int32_p)), "xrsfm_ba_debug_pack_gram")
    keys = ("gram_tiles", "table_cells", "cam_entries_g", "max_cams", "items_small", "items_big", "items_other", "block_writes")
    out = dict(zip(keys, (int(v) for v in stats)))
    out.update(tile_ncam=ncam[:st["tiles"]], slot_cidx=cidx[:st["slots"]], slot_campos_g=cpg[:st["slots"]], slot_obs=st["slot_obs"],
               items=st["items"], long_items=st["long_items"])
    return out
