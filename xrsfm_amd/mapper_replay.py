"""Driver of tests/shim/mapper_main (the mapper-shaped call sequence of IncrementalMapper::Reconstruct,
/root/reference/src/mapper/incremental_mapper.cc:33-88, through the source-compatible BASolver adapter on the test shim of
base/map.h).  Used by tests/test_mapper_replay.py and bench.py --config M; measurement / test helper, not part of the library."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim")
EXE = os.path.join(SHIM, "_build", "mapper_main")
CLASSES = ("GBA", "LBA", "KGBA", "refine", "frame_filter", "map_filter")
FILTER_CLASSES = ("refine", "frame_filter", "map_filter")      # "filters+refine" of rounds 3-5 = their sum


def build() -> str:
    subprocess.run(["make", "-C", SHIM, "_build/mapper_main"], check=True, capture_output=True)
    return EXE


def dump(arr: dict, path: str) -> None:
    with open(path, "wb") as f:
        f.write(struct.pack("4i", arr["cam_q"].shape[0], arr["points"].shape[0], arr["obs_cam"].shape[0], arr["intr_model"].shape[0]))
        for k, dt in (("cam_q", "f8"), ("cam_t", "f8"), ("cam_intr", "i4"), ("intr_model", "i4"), ("intr_params", "f8"),
                      ("points", "f8"), ("obs_cam", "i4"), ("obs_pt", "i4"), ("obs_uv", "f8")):
            f.write(np.ascontiguousarray(arr[k], dtype=dt).tobytes())


def sequence_problem(n_frames: int = 300, n_points: int = 45000, k_obs: int = 8, seed: int = 21, dropout: float = 0.25) -> dict:
    """A sequential reconstruction: `n_frames` frames on the generator's trajectory, every point seen in a window of `k_obs`
    consecutive frames with missed detections (ragged tracks).  Frame poses and points start a few pixels off (rotation
    0.002 rad, centre 0.01, points 0.02 units: what PnP and a two-view triangulation leave; the generator's default
    perturbation — 19 px median — is beyond the 16 px of the per-frame outlier filter, th_rpe_lba, and would have every new
    track filtered before its first LBA).  ~n_points * k_obs * (1 - dropout) / n_frames features per frame (900 at the defaults)."""
    from xrsfm_amd import capi, synth
    d = synth.make_problem(n_frames, n_points, k_obs, seed=seed, dropout=dropout, min_tri_angle_deg=1.0, perturb=(0.002, 0.01, 0.02))
    return {k: d[k] for k in capi.ProblemArrays.FIELDS}


def run(arr: dict, repeats: int = 1, timeout: float = 1800.0) -> dict:
    """Replays the reconstruction `repeats` times in ONE process.  Returns dict(status, same_end_state, cam_q, cam_t, points,
    n_outlier_tracks, replays=[{classes: {name: {count,total_ms,p50,p90,p99,max}}, wall_ms, free_bytes}])."""
    exe = build()
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        dump(arr, inp)
        p = subprocess.run([exe, inp, out, str(repeats)], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout)
        if not os.path.exists(out):
            raise RuntimeError(f"mapper_main failed (rc {p.returncode}): {p.stderr[-2000:]}")
        raw = open(out, "rb").read()
    nc, npt = arr["cam_q"].shape[0], arr["points"].shape[0]
    status, same = struct.unpack("2i", raw[:8])
    off = 8
    cams = np.frombuffer(raw, dtype="f8", count=7 * nc, offset=off).reshape(nc, 7); off += 56 * nc
    pts = np.frombuffer(raw, dtype="f8", count=3 * npt, offset=off).reshape(npt, 3); off += 24 * npt
    n_out, n_never = struct.unpack("2i", raw[off:off + 8]); off += 8
    replays = []
    for _ in range(repeats):
        nrec = 6 * len(CLASSES) + 3
        rec = np.frombuffer(raw, dtype="f8", count=nrec, offset=off); off += nrec * 8
        classes = {}
        for c, name in enumerate(CLASSES):
            v = rec[6 * c:6 * c + 6]
            classes[name] = dict(count=int(v[0]), total_ms=float(v[1]), p50=float(v[2]), p90=float(v[3]), p99=float(v[4]), max=float(v[5]))
        fr = [classes[n] for n in FILTER_CLASSES]
        classes["filters+refine"] = dict(count=sum(c["count"] for c in fr), total_ms=sum(c["total_ms"] for c in fr), p50=float("nan"), p90=float("nan"),
                                         p99=float("nan"), max=max(c["max"] for c in fr))
        replays.append(dict(classes=classes, wall_ms=float(rec[6 * len(CLASSES)]), free_bytes=int(rec[6 * len(CLASSES) + 1])))
    return dict(status=status, same_end_state=bool(same), cam_q=cams[:, :4].copy(), cam_t=cams[:, 4:].copy(), points=pts.copy(),
                n_outlier_tracks=n_out - n_never, n_never_triangulated=n_never, replays=replays, returncode=p.returncode, stderr=p.stderr[-2000:])
