"""ctypes wrapper of oracle/ba_cpu.c (TEST INFRASTRUCTURE: CPU baseline + at-scale checker)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libba_cpu.so")
_lib = None


class _Problem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_obs", C.c_int32), ("n_intr", C.c_int32),
                ("cam_q", C.c_void_p), ("cam_t", C.c_void_p), ("cam_const", C.c_void_p), ("cam_intr", C.c_void_p),
                ("intr_model", C.c_void_p), ("intr_params", C.c_void_p), ("points", C.c_void_p),
                ("point_const", C.c_void_p), ("obs_cam", C.c_void_p), ("obs_pt", C.c_void_p), ("obs_uv", C.c_void_p)]


class _Options(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("function_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("gradient_tolerance", C.c_double), ("initial_radius", C.c_double), ("huber_a", C.c_double),
                ("linear_solver", C.c_int32), ("pcg_tolerance", C.c_double), ("pcg_max_iterations", C.c_int32),
                ("profile", C.c_int32), ("verbose", C.c_int32)]


class _Summary(C.Structure):
    _fields_ = [("initial_cost", C.c_double), ("final_cost", C.c_double), ("n_successful", C.c_int),
                ("n_unsuccessful", C.c_int), ("termination", C.c_int), ("reason", C.c_int),
                ("num_effective_params", C.c_int), ("linearize_s", C.c_double), ("solve_s", C.c_double),
                ("total_s", C.c_double)]


def available() -> bool:
    return os.path.exists(_LIB)


def build():
    import subprocess
    subprocess.run(["make", "-C", _HERE], check=True)


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_LIB)
        _lib.ba_cpu_solve.argtypes = [C.POINTER(_Options), C.POINTER(_Problem), C.POINTER(_Summary), C.c_int]
        _lib.ba_cpu_solve.restype = C.c_int
    return _lib


def solve(prob: dict, max_iterations=50, function_tolerance=1e-5, parameter_tolerance=1e-6,
          initial_radius=1e4, huber_a=5.99, threads=0) -> dict:
    """Runs LM in place on the arrays of `prob` (keys = fields of xrsfm_ba_problem)."""
    a = {}
    for k, dt in (("cam_q", np.float64), ("cam_t", np.float64), ("cam_const", np.uint8), ("cam_intr", np.int32),
                  ("intr_model", np.int32), ("intr_params", np.float64), ("points", np.float64),
                  ("point_const", np.uint8), ("obs_cam", np.int32), ("obs_pt", np.int32), ("obs_uv", np.float64)):
        a[k] = np.ascontiguousarray(prob[k], dtype=dt)
    p = _Problem(a["cam_q"].shape[0], a["points"].shape[0], a["obs_cam"].shape[0], a["intr_model"].shape[0],
                 *[a[k].ctypes.data for k in ("cam_q", "cam_t", "cam_const", "cam_intr", "intr_model", "intr_params",
                                              "points", "point_const", "obs_cam", "obs_pt", "obs_uv")])
    o = _Options(max_iterations, function_tolerance, parameter_tolerance, 1e-10, initial_radius, huber_a, 0, 0.0, 0, 0, 0)
    s = _Summary()
    rc = _load().ba_cpu_solve(C.byref(o), C.byref(p), C.byref(s), threads)
    if rc != 0:
        raise RuntimeError(f"ba_cpu_solve failed: {rc}")
    for k in ("cam_q", "cam_t", "points", "intr_params"):        # intr_params: changed in bal9 mode only (variable {f, k1, k2})
        prob[k][...] = a[k].reshape(np.asarray(prob[k]).shape)
    return {f: getattr(s, f) for f, _ in s._fields_}
