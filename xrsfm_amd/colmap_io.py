"""Python mirror of xrsfm_amd/csrc/io/colmap_model.h (COLMAP-binary model as the reference reads/writes it,
/root/reference/src/utility/io_ecim.cc:9-84, 145-235): used by the tests to build and check replay inputs/outputs."""
from __future__ import annotations

import os
import struct

import numpy as np

NUM_PARAMS = (3, 4, 4, 5, 8)
NO_TRACK = 0xFFFFFFFFFFFFFFFF


def write_model(path: str, arr: dict, names=None):
    """arr: flat BA problem (keys of xrsfm_ba_problem); one image per camera, one camera file entry per intrinsics set."""
    os.makedirs(path, exist_ok=True)
    n_c, n_p = arr["cam_q"].shape[0], arr["points"].shape[0]
    with open(os.path.join(path, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", arr["intr_model"].shape[0]))
        for i, (mid, prm) in enumerate(zip(arr["intr_model"], arr["intr_params"])):
            f.write(struct.pack("<IIQQ", i, int(mid), 1241, 376))
            f.write(np.asarray(prm[:NUM_PARAMS[int(mid)]], "<f8").tobytes())
    order = np.lexsort((arr["obs_pt"], arr["obs_cam"]))
    oc, op, uv = arr["obs_cam"][order], arr["obs_pt"][order], arr["obs_uv"][order]
    ptr = np.searchsorted(oc, np.arange(n_c + 1))
    p2d_index = (np.arange(oc.shape[0]) - ptr[oc]).astype(np.int64)      # index of the 2D feature inside its image
    rec2d = np.dtype([("x", "<f8"), ("y", "<f8"), ("track", "<u8")])
    with open(os.path.join(path, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", n_c))
        for c in range(n_c):
            q = arr["cam_q"][c]
            f.write(struct.pack("<I", c)); f.write(np.array([q[3], q[0], q[1], q[2]], "<f8").tobytes())
            f.write(np.asarray(arr["cam_t"][c], "<f8").tobytes()); f.write(struct.pack("<I", int(arr["cam_intr"][c])))
            f.write(((names[c] if names else f"img{c:05d}.png") + "\0").encode())
            n2 = ptr[c + 1] - ptr[c]
            f.write(struct.pack("<Q", n2 + 1))
            p2 = np.empty(n2 + 1, rec2d)
            p2["x"][:n2] = uv[ptr[c]:ptr[c + 1], 0]; p2["y"][:n2] = uv[ptr[c]:ptr[c + 1], 1]; p2["track"][:n2] = op[ptr[c]:ptr[c + 1]]
            p2["x"][n2] = 1.0; p2["y"][n2] = 2.0; p2["track"][n2] = NO_TRACK          # a 2D feature without a track
            f.write(p2.tobytes())
    by_pt = np.lexsort((oc, op))
    pptr = np.searchsorted(op[by_pt], np.arange(n_p + 1))
    oc_s = oc[by_pt].astype("<i4"); p2_s = p2d_index[by_pt].astype("<i4")
    pairs = np.stack([oc_s, p2_s], 1)                                      # (frame, 2D feature) of every observation, track-major
    pts = np.asarray(arr["points"], "<f8")
    with open(os.path.join(path, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", n_p))
        chunks = []
        for j in range(n_p):
            chunks.append(struct.pack("<Qddd3Bd", j, pts[j, 0], pts[j, 1], pts[j, 2], 0, 0, 0, -1.0) + struct.pack("<Q", int(pptr[j + 1] - pptr[j]))
                          + pairs[pptr[j]:pptr[j + 1]].tobytes())
            if len(chunks) >= 65536:
                f.write(b"".join(chunks)); chunks = []
        f.write(b"".join(chunks))


def read_model(path: str, with_points: bool = True) -> dict:
    """with_points=False skips points3D.bin (a million-point model takes a while in a Python loop)."""
    cams, images, points = {}, {}, {}
    with open(os.path.join(path, "cameras.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, mid, w, h = struct.unpack("<IIQQ", f.read(24))
            cams[cid] = (mid, np.frombuffer(f.read(8 * NUM_PARAMS[mid]), "<f8").copy())
    with open(os.path.join(path, "images.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            (iid,) = struct.unpack("<I", f.read(4))
            q = np.frombuffer(f.read(32), "<f8").copy(); t = np.frombuffer(f.read(24), "<f8").copy()
            (cam,) = struct.unpack("<I", f.read(4))
            name = b""
            while True:
                ch = f.read(1)
                if ch == b"\0":
                    break
                name += ch
            (n2,) = struct.unpack("<Q", f.read(8))
            p2 = np.frombuffer(f.read(24 * n2), dtype=[("x", "<f8"), ("y", "<f8"), ("track", "<u8")]).copy()
            images[iid] = dict(q_wxyz=q, t=t, camera=cam, name=name.decode(), points=p2)
    with open(os.path.join(path, "points3D.bin"), "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n if with_points else 0):
            (pid,) = struct.unpack("<Q", f.read(8))
            xyz = np.frombuffer(f.read(24), "<f8").copy(); f.read(3)
            (err,) = struct.unpack("<d", f.read(8)); (no,) = struct.unpack("<Q", f.read(8))
            obs = np.frombuffer(f.read(8 * no), "<i4").reshape(no, 2).copy()
            points[pid] = dict(xyz=xyz, error=err, obs=obs)
    return dict(cameras=cams, images=images, points=points)
