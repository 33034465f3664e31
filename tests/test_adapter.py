"""The source-compatible xrsfm::BASolver adapter (xrsfm_amd/csrc/compat), compiled against the TEST SHIM of base/map.h
(tests/shim: Eigen/OpenCV/glog are not in this image) and linked with libxrsfm_ba.so.

CPU: it builds, links, runs, and reports ENODEV without touching the map.  GPU: GBA / structure-only / KGBA / LBA through
the adapter equal the C-ABI called directly on the equivalent flat problem (same frames, constants and options)."""
import os
import re
import struct
import subprocess

import numpy as np
import pytest

from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim")
EXE = os.path.join(SHIM, "_build", "adapter_main")


@pytest.fixture(scope="module")
def exe(lib):
    subprocess.run(["make", "-C", SHIM], check=True, capture_output=True)
    return EXE


def _dump(arr, path):
    with open(path, "wb") as f:
        f.write(struct.pack("4i", arr["cam_q"].shape[0], arr["points"].shape[0], arr["obs_cam"].shape[0], arr["intr_model"].shape[0]))
        for k, dt in (("cam_q", "f8"), ("cam_t", "f8"), ("cam_intr", "i4"), ("intr_model", "i4"), ("intr_params", "f8"),
                      ("points", "f8"), ("obs_cam", "i4"), ("obs_pt", "i4"), ("obs_uv", "f8")):
            f.write(np.ascontiguousarray(arr[k], dtype=dt).tobytes())


def _run(exe, arr, tmp_path, mode, *extra):
    inp, out = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    _dump(arr, inp)
    p = subprocess.run([exe, inp, out, mode, *map(str, extra)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    raw = open(out, "rb").read()
    status = struct.unpack("i", raw[:4])[0]
    nc, npt = arr["cam_q"].shape[0], arr["points"].shape[0]
    cams = np.frombuffer(raw, dtype="f8", count=7 * nc, offset=4).reshape(nc, 7)
    pts = np.frombuffer(raw, dtype="f8", count=3 * npt, offset=4 + 56 * nc).reshape(npt, 3)
    return status, cams[:, :4].copy(), cams[:, 4:].copy(), pts.copy(), p.stdout, p.stderr


def test_adapter_builds_and_fails_loudly_without_gpu(exe, tmp_path):
    import torch
    from xrsfm_amd import capi
    if torch.cuda.is_available() and capi.device_count() > 0:
        pytest.skip("a GPU is present")
    arr = H.make(6, 40, 3, seed=130)
    status, q, t, P, out, err = _run(exe, arr, tmp_path, "gba")
    assert status == -2 and "no CPU fallback" in err
    assert np.array_equal(q, arr["cam_q"]) and np.array_equal(t, arr["cam_t"]) and np.array_equal(P, arr["points"])


def _subproblem(arr, frames, lba_frame=None):
    """Flat problem of the frames `frames` (ascending ids), like FlatProblem::AddFrame builds it."""
    frames = list(frames)
    fmap = {f: i for i, f in enumerate(frames)}
    keep = np.isin(arr["obs_cam"], frames)
    order = np.lexsort((np.arange(keep.sum()), arr["obs_cam"][keep]))      # frame-major, original order inside a frame
    oc = arr["obs_cam"][keep][order]; op = arr["obs_pt"][keep][order]; uv = arr["obs_uv"][keep][order]
    pts, inv = np.unique(op, return_inverse=True)
    sub = dict(cam_q=arr["cam_q"][frames], cam_t=arr["cam_t"][frames], cam_const=np.zeros(len(frames), np.uint8),
               cam_intr=arr["cam_intr"][frames], intr_model=arr["intr_model"], intr_params=arr["intr_params"],
               points=arr["points"][pts], point_const=np.zeros(len(pts), np.uint8),
               obs_cam=np.array([fmap[c] for c in oc], np.int32), obs_pt=inv.astype(np.int32), obs_uv=uv)
    if lba_frame is not None:
        seen = np.zeros(arr["points"].shape[0], bool); seen[arr["obs_pt"][arr["obs_cam"] == lba_frame]] = True
        sub["point_const"] = (~seen[pts]).astype(np.uint8)
    return sub, frames, pts


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gba", "gba_fast", "structure", "kgba", "lba"])
def test_adapter_equals_direct_c_abi(exe, tmp_path, mode):
    from xrsfm_amd import capi
    arr = H.with_models(H.make(9, 260, 4, seed=131), seed=2)
    nc = arr["cam_q"].shape[0]
    lba_frame = 2
    status, q, t, P, out, err = _run(exe, arr, tmp_path, mode, lba_frame)
    assert status == 0, err
    if mode in ("gba", "gba_fast", "structure"):
        frames = list(range(nc)); lba = None
        kw = dict(max_iterations=50, function_tolerance=1e-5, parameter_tolerance=1e-6) if mode != "gba_fast" else \
            dict(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5)
    elif mode == "kgba":
        frames = [i for i in range(nc) if i % 2 == 0 or i in (0, 1, 3)]; lba = None      # shim KeyFrameSelection + forced {3}
        kw = dict(max_iterations=20, function_tolerance=1e-4, parameter_tolerance=1e-5, initial_radius=1e6)
        assert f"kf: {len(frames)}/{nc}" in out
    else:
        m = re.search(r"LBA:\s*((?:\d+ ?)+)", out)
        frames = sorted(int(x) for x in m.group(1).split()); lba = lba_frame
        assert lba in frames and len(frames) <= 7
        kw = dict(max_iterations=5, function_tolerance=1e-4, parameter_tolerance=1e-5)
    sub, frames, pts = _subproblem(arr, frames, lba)
    if mode == "structure":
        sub["cam_const"][:] = 3
    else:
        fixed = [f for f in (0, 1) if f in frames]
        if mode == "lba" and not fixed:
            fixed = None   # fallback gauge depends on the bundle order; covered by the cost check below only
        if fixed:
            for f in fixed: sub["cam_const"][frames.index(f)] |= 2
    if mode == "lba" and not [f for f in (0, 1) if f in frames]:
        pytest.skip("local set without init frames: gauge fallback order is implementation defined")
    prod = H.to_product(sub)
    s = capi.solve(prod, capi.default_options(**kw))
    assert np.abs(q[frames] - prod.cam_q).max() < 1e-9 and np.abs(t[frames] - prod.cam_t).max() < 1e-9
    assert np.abs(P[pts] - prod.points).max() < 1e-7
    others = [i for i in range(nc) if i not in frames]
    assert np.array_equal(q[others], arr["cam_q"][others])            # frames outside the problem are untouched
    if mode in ("gba", "kgba"):
        assert "Residuals : " in out and "Termination : " in out and f"{2 * sub['obs_cam'].shape[0]}" in out


@pytest.mark.gpu
def test_adapter_refine_pose_equals_c_abi(exe, tmp_path):
    """RefineFramePose (compat header; replaces the Ceres block of pnp.cc:38-71) = xrsfm_ba_refine_pose on the same inliers."""
    from xrsfm_amd import capi
    arr = H.make_pose_problem(120, seed=77, model=2)
    status, q, t, P, out, err = _run(exe, arr, tmp_path, "refine")
    assert status == 0, err
    mask = (np.arange(1, 121) % 7 != 0).astype(np.uint8)
    q2, t2, s = capi.refine_pose(2, arr["intr_params"][0], arr["points"], arr["obs_uv"], arr["cam_q"][0], arr["cam_t"][0], inlier_mask=mask)
    assert np.array_equal(q[0], q2) and np.array_equal(t[0], t2)
    assert np.array_equal(P, arr["points"])                     # points are constant
    m = re.search(r"Initial cost : ([0-9.eE+-]+) \[px\]\nFinal cost : ([0-9.eE+-]+) \[px\]", out)
    assert m, out
    assert abs(float(m.group(2)) - np.sqrt(s.final_cost / s.num_residuals)) < 1e-4 * max(1.0, float(m.group(2)))
