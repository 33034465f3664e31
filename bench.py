#!/usr/bin/env python3
"""bench.py — BA iterations/sec x (cams+points) on synthetic BAL-style problems (BASELINE.json).

A "step" is one complete GBA-shaped solve (reference settings of
/root/reference/src/optimization/ba_solver.cc:626-629: <=50 LM iterations,
ftol 1e-5, ptol 1e-6, Huber 5.99) from the same perturbed initial state, with
the problem already resident in HBM (xrsfm_ba_create is outside the timed
region; xrsfm_ba_reset restores the state between steps).  value = LM
iterations (successful + unsuccessful, as the reference counts them,
ba_solver.cc:22-25) x (cams + points) / second, whole job.

N > 1: one process per GPU (torch.distributed.run); the tracks are sharded by
point over the ranks, cameras replicated, per-camera sums and the reduced camera
blocks all-reduced with RCCL inside the library (xrsfm_ba_comm_init).
  --scaling strong (default): the SAME problem is split over the ranks:
      BASELINE.json config 4 read literally (1k cams / 500k points / 2M obs,
      points sharded N ways).  The exact factorisation of the reduced camera
      system is replicated on every rank, so this is bounded by Amdahl
      (DESIGN.md section 6 has the projection from measured kernel times);
  --scaling weak: the point set grows with N (N x 500k points over the same
      1000 cameras at config L; every rank generates and holds one config-sized
      shard) -- the regime the sharding is for (maps that outgrow one GPU).

The JSON line carries `roofline` (dominant HBM-streaming kernel — of those within 10 % of the largest total the one furthest below its roofline —, algorithmic
bytes of SURVEY.md section 8d / DESIGN.md section 5, duration from HIP events
recorded by the library on its own stream; `roofline.iteration` = the whole LM
iteration against BASELINE.md section 4's B_iter), `cpu_baseline` (oracle/ C
restatement timed on the host cores, rank 0, N = 1 only), `host_inclusive`
(the one-shot xrsfm_ba_solve the BASolver adapter calls: packing + upload +
solve + download) and `mfma_utilisation` (FP64 matrix-core rate of the dense
reduced-camera factorisation, measured on config D in the same job).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def shard_problem(arr: dict, rank: int, world: int) -> dict:
    """The shard of `rank`: points assigned by the length-aware greedy of SURVEY.md section 8(e) (xrsfm_amd/sharding.py:
    observations per rank within 1 % whatever the track-length distribution), their observations, all cameras replicated."""
    from xrsfm_amd import sharding
    return sharding.shard_problem(arr, rank, world)


def weak_scaled_shard(cfg: dict, rank: int, world: int) -> dict:
    """Weak scaling: the problem of `world` ranks is the union of `world` configuration-sized point shards over the same
    cameras (synth.make_problem(point_seed=rank): cameras from the seed alone, points from a stream per shard), so every rank
    generates only its own shard.  world == 1 is the configuration itself (the historical single-stream problem)."""
    from xrsfm_amd import synth
    return synth.make_problem(**cfg) if world == 1 else synth.make_problem(**cfg, point_seed=rank)


def make_config(name: str) -> dict:
    """The synthetic problem of a named configuration (xrsfm_amd/synth.py: CONFIGS).  T = BASELINE.json config 5 at its size
    (unordered photo collection with viewpoint clusters), everything else the BAL-style generator of SURVEY.md Appendix D."""
    from xrsfm_amd import synth
    cfg = dict(synth.CONFIGS[name])
    if name == "Lb9":            # config 4 with 9-wide camera blocks: every camera's {f, k1, k2} is a variable (bal9 mode)
        return synth.to_bal9(synth.make_problem(**cfg))
    return synth.make_collection(**cfg) if name == "T" else synth.make_problem(**cfg)


def algorithmic_bytes(kernel: str, n_obs: int, n_pts: int, n_cams: int, nnzb: int = 0, width: int = 6):
    """ALGORITHMIC HBM bytes of one launch, J-stored accounting of SURVEY.md section 8(d) / DESIGN.md section 5
    (FP64 values, int32 indices).  None for kernels that are latency- or MFMA-bound.  width = 9: the bal9 rows of the same
    section (B_lin = 232, B_pcg(1) = 192, explicit-S blocks of 648 bytes), cameras with 3 intrinsics more."""
    if width == 9:
        w, ws = 9, 45                                  # camera block width, entries of a symmetric diagonal block
        table9 = {
            "k_linearize": n_obs * (24 + 16 + 2 * 8 * (w + 3)) + n_pts * 24 + n_cams * 80,
            "k_schur_pairs": n_obs * (16 + 2 * 8 * (w + 3)) + n_pts * 72 + n_cams * 8 * (ws + w) + nnzb * 8 * w * w,
            "k_backsub": n_obs * (16 + 2 * 8 * (w + 3)) + n_pts * 144 + n_cams * 8 * w,
            "k_cost": n_obs * 24 + n_pts * 24 + n_cams * 80,
        }
        return table9.get(kernel)
    table = {
        # read uv + 2 idx, write r[2] + Jc[2x6] + Jp[2x3]; read points and cameras            (B_lin)
        "k_linearize": n_obs * (24 + 160) + n_pts * 24 + n_cams * 56,
        # one implicit Schur product: read Jc, Jp, Hpp^-1; x in, y out                        (B_pcg(1))
        "k_schur_matvec": n_obs * 144 + n_pts * 48 + n_cams * 96,
        # read r, J; read Hpp^-1 (6) + g_p (3); write diagonal blocks of S (21) + rhs (6)      (B_prep)
        "k_schur_prep": n_obs * 160 + n_pts * 72 + n_cams * 216,
        # fused S assembly of the Cholesky path = B_prep + the nnzb off-diagonal 6x6 blocks of S written once
        # (explicit-S accounting of SURVEY 8d; r and J are read once for both)
        "k_schur_pairs": n_obs * 160 + n_pts * 72 + n_cams * 216 + nnzb * 288,
        # back-substitute: read J, r; Hpp^-1, g_p, points in, points out                       (B_back)
        "k_backsub": n_obs * (144 + 16) + n_pts * 144 + n_cams * 48,
        "k_cost": n_obs * 24 + n_pts * 24 + n_cams * 56,
    }
    return table.get(kernel)


def count_offdiag_blocks(arr: dict) -> int:
    """nnzb of the strictly lower triangle of the reduced camera matrix: distinct camera pairs sharing a track."""
    order = np.lexsort((arr["obs_cam"], arr["obs_pt"]))
    cam = arr["obs_cam"][order].astype(np.int64); pt = arr["obs_pt"][order]
    keys = []
    for d in range(1, 65):
        same = pt[d:] == pt[:-d]
        if not same.any():
            break
        keys.append(cam[d:][same] * (1 << 32) + cam[:-d][same])
    return int(np.unique(np.concatenate(keys)).shape[0]) if keys else 0


def gauge_aligned_centre_diff(q1, t1, q2, t2):
    """max |c1 - (s R c2 + T)| over the camera centres after the best similarity alignment (Umeyama).  Fixing the translations
    of two frames (the reference's gauge, ba_solver.cc:611-614) leaves a similarity direction almost free, along which two exact
    FP64 solvers with different summation orders drift apart without any effect on the cost; this removes that component."""
    def centres(q, t):
        x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                      2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                      2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
        return -np.einsum("nji,nj->ni", R, t)
    c1, c2 = centres(q1, t1), centres(q2, t2)
    m1, m2 = c1.mean(0), c2.mean(0)
    a, b = c1 - m1, c2 - m2
    U, S, Vt = np.linalg.svd(a.T @ b / len(a))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    sc = np.trace(np.diag(S) @ D) / (b ** 2).sum(1).mean()
    return float(np.abs(c1 - (sc * (R @ c2.T).T + m1 - sc * R @ m2)).max())


def parity_block(arr: dict, q, t, P, cpu_prob: dict):
    """The criterion of tests/test_gpu_parity.py::test_headline_config_camera_parity between the HIP result and the CPU port's:
    Gauss-Newton energy of the whole difference relative to the cost, and the largest difference of the relative pose of
    covisible camera pairs (gauge invariant, free of the drift that the reference's two-translation gauge leaves open on
    long trajectories).  Raw parameter differences are reported next to it."""
    from oracle import ba_oracle as bo
    from xrsfm_amd import capi, parity
    pairs = parity.covisible_pairs(arr["obs_cam"], arr["obs_pt"])
    dv = parity.tangent_difference(q, t, cpu_prob["cam_q"], cpu_prob["cam_t"]).reshape(-1, 6)
    ang, dtr = parity.relative_pose_difference(q, t, cpu_prob["cam_q"], cpu_prob["cam_t"], pairs)
    pr = bo.Problem(**{k: np.array(v, copy=True) for k, v in dict(arr, cam_q=q, cam_t=t, points=P).items() if k in capi.ProblemArrays.FIELDS})
    cost, rt, Fc, Ep = bo.evaluate(pr, pr.cam_q, pr.cam_t, pr.points)
    Jd = np.einsum("nij,nj->ni", np.asarray(Fc).reshape(-1, 2, 6), dv[pr.obs_cam]) + \
        np.einsum("nij,nj->ni", np.asarray(Ep).reshape(-1, 2, 3), (P - cpu_prob["points"])[pr.obs_pt])
    return {"criterion": "1/2|J dx|^2 <= 1e-10 cost; relative pose of covisible cameras within 1e-6 rad / 1e-4 units "
                         "(tests/test_gpu_parity.py::test_headline_config_camera_parity)",
            "gn_energy_over_cost": 0.5 * float((Jd ** 2).sum()) / cost, "covisible_pairs": int(pairs.shape[0]),
            "rel_pose_rot_rad": ang, "rel_pose_trans": dtr,
            "raw_max_rotation_tangent": float(np.abs(dv[:, :3]).max()), "raw_max_translation": float(np.abs(dv[:, 3:]).max())}


def cpu_baseline(arr: dict, n_cams: int, n_points: int, max_iterations: int):
    """oracle/ C restatement (kind "port") timed on the host cores; None if it is not built."""
    try:
        from oracle import ba_cpu
    except Exception:
        return None
    if not ba_cpu.available():
        return None
    threads = min(8, os.cpu_count() or 1)       # the reference asks Ceres for 8 threads (ba_solver.cc:72)
    prob = {k: np.array(v, copy=True) for k, v in arr.items()}
    t0 = time.perf_counter()
    summ = ba_cpu.solve(prob, max_iterations=max_iterations, threads=threads)
    dt = time.perf_counter() - t0
    iters = summ["n_successful"] + summ["n_unsuccessful"]
    out = {
        "value": iters * (n_cams + n_points) / dt, "unit": "cam-pts*iter/s", "cores": threads, "kind": "port",
        "sample": f"same problem, full solve ({iters} LM iterations, {dt:.2f} s), analytic Jacobians, exact Schur + "
                  f"envelope Cholesky, OpenMP; host has {os.cpu_count()} cores",
        "final_rmse_px": math.sqrt(summ["final_cost"] / (2 * arr["obs_cam"].shape[0])),
        "iterations": iters, "seconds": dt,
    }
    # beyond the reference's 8 threads: as many as the host really grants — the GPU boxes show 256 CPUs under a cgroup quota of
    # 16 (cpu.max "1600000 100000"): 64 OpenMP threads there are throttled to 16 cores' worth of time and run no faster than 8
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except Exception:
        pass
    all_cores = min(os.cpu_count() or 1, 64, quota or 64)
    out["sample"] += f" (cgroup CPU quota: {quota if quota else 'none'})"
    if all_cores > threads:                      # also timed beyond the reference's 8 threads
        prob2 = {k: np.array(v, copy=True) for k, v in arr.items()}
        t0 = time.perf_counter()
        s2 = ba_cpu.solve(prob2, max_iterations=max_iterations, threads=all_cores)
        dt2 = time.perf_counter() - t0
        out["more_threads"] = {"cores": all_cores, "value": (s2["n_successful"] + s2["n_unsuccessful"]) * (n_cams + n_points) / dt2,
                               "seconds": dt2}
    return out, prob


def parity_workload(opt):
    """Config LP (xrsfm_amd/synth.py): BASELINE.json config 4's sizes and code path plus 1200 distant-landmark tracks over 24 hub
    frames, which make the absolute camera parameters well determined (the plain L problem leaves 1e-3 of gauge drift along its
    1000-frame loop, on which no two solvers — not even the C restatement with its points relabelled — agree to 1e-5).  The HIP
    solve against the C restatement on the host cores, the LITERAL north-star bounds: |d RMSE| <= 1e-6 px, cameras <= 1e-5."""
    from oracle import ba_cpu
    from xrsfm_amd import capi, synth
    if not ba_cpu.available():
        return None
    d = synth.make_problem(**synth.CONFIGS["LP"])
    arr = {k: d[k] for k in capi.ProblemArrays.FIELDS}
    prod = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
    t0 = time.perf_counter()
    s = capi.solve(prod, opt)
    t_gpu = time.perf_counter() - t0
    cpu = {k: np.array(v, copy=True) for k, v in arr.items()}
    threads = min(8, os.cpu_count() or 1)
    sc = ba_cpu.solve(cpu, max_iterations=opt.max_iterations, threads=threads)
    n_res = 2 * arr["obs_cam"].shape[0]
    return {"workload": "LP: config 4 + 24 hub frames x 50 distant landmarks (1000 cams / 500000 points / 2000000 obs)",
            "lm_steps_hip": [s.n_successful, s.n_unsuccessful], "lm_steps_cpu": [sc["n_successful"], sc["n_unsuccessful"]],
            "final_rmse_px": math.sqrt(s.final_cost / n_res),
            "rmse_diff_px": abs(math.sqrt(s.final_cost / n_res) - math.sqrt(sc["final_cost"] / n_res)),
            "max_cam_param_diff": float(max(np.abs(cpu["cam_q"] - prod.cam_q).max(), np.abs(cpu["cam_t"] - prod.cam_t).max())),
            "max_point_diff": float(np.abs(cpu["points"] - prod.points).max()),
            "bounds": "north star: RMSE within 1e-6 px, camera parameters within 1e-5", "cpu_threads": threads,
            "one_shot_hip_s": t_gpu, "cpu_s": sc["total_s"]}


def self_profile(config: str, solver: str):
    """rocprofv3 passes launched BY THIS RUN: the per-kernel duration table (--kernel-trace --stats) and
    the HBM traffic per launch (FETCH_SIZE and WRITE_SIZE in passes of their own — counters only + kernel trace — corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes and tools/pmc_summary.py documents: FETCH_SIZE x 2 for these 8-byte-per-lane
    streams, WRITE_SIZE x 1; both counters are KiB per dispatch) of a short child run of the same configuration.  None when
    rocprofv3 is not on PATH, in a child, or when a pass fails (the caller then falls back to the committed profiles/ JSON)."""
    import re
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if os.environ.get("XRSFM_BENCH_CHILD") == "1" or os.environ.get("XRSFM_BENCH_SELFPROF") == "0":
        return None
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None
    env = dict(os.environ, XRSFM_BENCH_CHILD="1", TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--solver", solver, "--no-cpu", "--no-extras", "--steps", "1", "--warmup", "1"]

    def one_pass(extra):
        d = tempfile.mkdtemp(prefix="xba_prof_", dir="/tmp")
        try:
            r = subprocess.run([rp, *extra, "-d", d, "-o", "p", "--", *child], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            sym_cols = [c[1] for c in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
            name_col = "display_name" if "display_name" in sym_cols else ("kernel_name" if "kernel_name" in sym_cols else sym_cols[1])
            clean = lambda n: re.sub(r"\(.*", "", n).replace("void ", "").replace("xba::", "")     # noqa: E731
            if "--pmc" in extra:
                q = (f"select s.{name_col}, count(*), avg(p.value) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id "
                     "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1")
                return {clean(n): (c, a) for n, c, a in cur.execute(q)}
            q = (f"select s.{name_col}, count(*), avg(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                 "on d.kernel_id = s.id group by 1")
            return {clean(n): (c, a) for n, c, a in cur.execute(q)}
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)

    stats = one_pass(["--kernel-trace", "--stats"])
    if stats is None:
        return None
    out = {"source": "rocprofv3 passes launched by this bench.py run (child: --steps 1 --warmup 1 --no-cpu --no-extras)",
           "kernel_avg_us": {k: round(a / 1e3, 3) for k, (c, a) in stats.items() if not k.startswith("__amd")},
           "kernel_calls": {k: c for k, (c, a) in stats.items() if not k.startswith("__amd")}}
    fetch = one_pass(["--pmc", "FETCH_SIZE", "--kernel-trace"])
    write = one_pass(["--pmc", "WRITE_SIZE", "--kernel-trace"])
    if fetch is not None and write is not None:
        traffic = {}
        for k in set(fetch) | set(write):
            b = 2.0 * fetch.get(k, (0, 0.0))[1] * 1024 + write.get(k, (0, 0.0))[1] * 1024
            if b >= 5e4:
                traffic[k] = int(round(b))
        out["traffic_bytes_per_launch"] = traffic
        out["traffic_correction"] = "FETCH_SIZE x 2 (gfx950: 128-byte requests of 8-byte-per-lane streams tallied at 64), WRITE_SIZE x 1; KiB per dispatch"
    return out


FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (datasheet; v_mfma_f64_16x16x4_f64 issues at the FP64 vector rate)


def host_inclusive(arr: dict, opt, n_cams: int, n_points: int):
    """The call the BASolver adapter makes (compat/optimization/ba_solver.cc: xrsfm_ba_solve = create + run + download + destroy)
    on host buffers: Map -> SoA packing, uploads, Cholesky set-up, the solve, the read-back.  Timed twice; the second call (device
    allocation cache warm, as in a mapper that calls BA repeatedly) is reported, the first one as `first_call_ms`."""
    from xrsfm_amd import capi
    out = None
    first = None
    for rep in range(2):
        prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in arr.items()})
        dst = (prob.cam_q, prob.cam_t, prob.points)      # results go back into the caller's own storage, as in the adapter
        t0 = time.perf_counter()
        ctx = capi.Context(prob)
        t1 = time.perf_counter()
        s = ctx.run(opt)
        t2 = time.perf_counter()
        ctx.download(out=dst)
        t3 = time.perf_counter()
        ctx.close()
        t4 = time.perf_counter()
        iters = s.n_successful + s.n_unsuccessful
        out = {"create_ms": (t1 - t0) * 1e3, "run_ms": (t2 - t1) * 1e3, "download_ms": (t3 - t2) * 1e3, "destroy_ms": (t4 - t3) * 1e3,
               "total_ms": (t4 - t0) * 1e3, "lm_iterations": iters, "value": iters * (n_cams + n_points) / (t4 - t0),
               "unit": "cam-pts*iter/s",
               "what": "one-shot solve on host buffers (upload + packing, on the device from 32k observations | Cholesky set-up + solve | download), "
                       "second call of the process"}
        if rep == 0:
            first = out["total_ms"]
    out["first_call_ms"] = first
    return out


def mfma_utilisation():
    """FP64 matrix-core rate of the reduced-camera solve where it is dense: config D (2000 cameras with random visibility = 12 000
    unknowns, full S, right-looking tile Cholesky).  flop = LM steps x n^3/3; time = HIP-event totals of the profiled solve."""
    from xrsfm_amd import capi, synth
    cfg = dict(synth.CONFIGS["D"])
    d = synth.make_problem(**cfg)
    prob = capi.ProblemArrays(**{k: np.array(d[k], copy=True) for k in capi.ProblemArrays.FIELDS})
    ctx = capi.Context(prob)
    opt = capi.default_options(max_iterations=4)
    ctx.run(opt)
    ctx.reset()
    s = ctx.run(capi.default_options(max_iterations=4, profile=1))
    prof = ctx.profile()
    ctx.close()
    n = 6 * prob.n_cams
    steps = s.lm_steps_attempted
    flop = steps * n ** 3 / 3.0
    t_upd = prof.get("k_update", (0.0, 0))[0] * 1e-3
    t_chain = t_upd + prof.get("k_potrf", (0.0, 0))[0] * 1e-3 + prof.get("k_trsm", (0.0, 0))[0] * 1e-3
    if s.linear_solver_used != capi.SOLVER_CHOLESKY or t_upd <= 0.0:
        return None
    return {"config": f"D: {prob.n_cams} cams random visibility, {n} camera unknowns, dense reduced matrix", "factorisations": steps,
            "flop": flop, "trailing_update_s": t_upd, "factorisation_chain_s": t_chain,
            "achieved_tflops": flop / t_upd / 1e12, "achieved_tflops_chain": flop / t_chain / 1e12, "peak_tflops": FP64_MFMA_PEAK_TFLOPS,
            "frac": flop / t_upd / 1e12 / FP64_MFMA_PEAK_TFLOPS, "frac_chain": flop / t_chain / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "dtype": "f64", "instruction": "v_mfma_f64_16x16x4_f64"}


def mapper_bench(args):
    """--config M: the mapper-shaped call sequence (IncrementalMapper::Reconstruct, incremental_mapper.cc:33-88: GBA once, LBA +
    filters per frame, KGBA + FilterPoints3d on the geometric schedule) of a 300-frame sequential reconstruction through the
    source-compatible BASolver adapter on the test shim of base/map.h.  One replay warms the process up (allocation caches,
    code objects), the second is reported: wall time of everything the replay does (host-side frame selection, packing,
    upload, solve, download included — the adapter's real cost) and per-call percentiles per call class.  Not the headline
    metric: a separate line for BASELINE config 1's call pattern."""
    import torch  # noqa: F401  (one HIP runtime for the child's library as well)
    from xrsfm_amd import mapper_replay
    arr = mapper_replay.sequence_problem()
    r = mapper_replay.run(arr, repeats=2)
    if r["status"] != 0:
        raise SystemExit(f"mapper replay failed: status {r['status']}: {r['stderr']}")
    rep = r["replays"][1]
    ba_ms = sum(rep["classes"][k]["total_ms"] for k in ("GBA", "LBA", "KGBA"))
    out = {"metric": "mapper-shaped BA replay: wall time of the BA calls of one incremental reconstruction", "value": ba_ms, "unit": "ms",
           "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": rep["wall_ms"], "higher_is_better": False, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"{arr['cam_q'].shape[0]} frames arriving one by one / {arr['points'].shape[0]} tracks / {arr['obs_cam'].shape[0]} observations; "
                                  "GBA once, LBA + FilterPointsFrame per frame, KGBA + FilterPoints3d when registered > 1.2 x last (incremental_mapper.cc:77)",
                      "parallelism": "single GPU, one-shot xrsfm_ba_solve per call through the BASolver adapter (tests/shim)"},
           "calls": rep["classes"], "replay_wall_ms": rep["wall_ms"], "first_replay_wall_ms": r["replays"][0]["wall_ms"],
           "same_end_state_in_both_replays": r["same_end_state"],
           "free_device_bytes_after_replay": [x["free_bytes"] for x in r["replays"]], "tracks_filtered": r["n_outlier_tracks"], "tracks_never_triangulated": r["n_never_triangulated"]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="L", choices=sorted(["S", "L", "LP", "K", "U", "X", "R", "V", "D", "L0", "T", "M", "Lb9"]))
    ap.add_argument("--pcg-tol", type=float, default=None)
    ap.add_argument("--solver", default="auto", choices=["auto", "pcg", "cholesky"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong = the config split N ways (BASELINE.json config 4); weak = N x the points of the config over the same cameras")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-inclusive one-shot solve and the config-D MFMA measurement")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    if args.config == "M":
        return mapper_bench(args)

    import torch
    import torch.distributed as dist
    from xrsfm_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available() or capi.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device; the BA path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    cfg = dict(synth.CONFIGS[args.config])
    if args.scaling == "weak":
        full = weak_scaled_shard(cfg, rank, world)
        arr = {k: full[k] for k in capi.ProblemArrays.FIELDS}
        local = arr
        n_cams = arr["cam_q"].shape[0]
        sizes = torch.tensor([arr["points"].shape[0], arr["obs_cam"].shape[0]], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(sizes)
        n_points, n_obs = int(sizes[0].item()), int(sizes[1].item())      # of the whole job
    else:
        full = make_config(args.config)
        arr = {k: full[k] for k in capi.ProblemArrays.FIELDS}
        n_cams, n_points, n_obs = arr["cam_q"].shape[0], arr["points"].shape[0], arr["obs_cam"].shape[0]
        local = shard_problem(arr, rank, world)
    prob = capi.ProblemArrays(**{k: np.array(v, copy=True) for k, v in local.items()})
    ctx = capi.Context(prob, device=local_rank)
    if world > 1 or os.environ.get("XRSFM_BA_FORCE_COMM") == "1":     # the env var exercises the RCCL path on one GPU
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid = torch.tensor(list(capi.comm_unique_id()), dtype=torch.uint8, device="cuda")
        if world > 1:
            dist.broadcast(uid, 0)
        ctx.comm_init(world, rank, bytes(uid.cpu().tolist()))

    opt = capi.default_options(verbose=1 if (args.verbose and rank == 0) else 0)
    if args.pcg_tol is not None:
        opt.pcg_tolerance = args.pcg_tol
    opt.linear_solver = {"auto": capi.SOLVER_AUTO, "pcg": capi.SOLVER_PCG, "cholesky": capi.SOLVER_CHOLESKY}[args.solver]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.reset()
        ctx.run(opt)
    barrier()
    t0 = time.perf_counter()
    iters = 0
    pcg = 0
    last = None
    for _ in range(args.steps):
        ctx.reset()
        last = ctx.run(opt)
        iters += last.n_successful + last.n_unsuccessful
        pcg += last.pcg_iterations
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # one extra, untimed, profiled solve: HIP-event durations per kernel class on the library's stream
    popt = capi.default_options(profile=1)
    popt.pcg_tolerance = opt.pcg_tolerance
    popt.linear_solver = opt.linear_solver
    ctx.reset()
    ctx.run(popt)
    q, t, P = ctx.download()
    kernels = ctx.profile()
    kernel_table = {k: {"ms": round(v[0], 4), "launches": v[1]} for k, v in kernels.items() if v[1] > 0}
    roofline = None
    # dominant kernel = largest HIP-event total among the HBM-streaming kernels
    cands = [(v[0], k) for k, v in kernels.items()
             if v[1] > 0 and algorithmic_bytes(k, 1, 1, 1, 0, 9 if args.config == "Lb9" else 6) is not None]
    if cands:
        # ... and among kernels within 10 % of that total the one FURTHEST below its roofline (round 6: at config L k_schur_pairs and
        # k_linearize are now 1.29 and 1.27 ms per solve — which of them leads changes from run to run; the line quotes the lower
        # fraction either way, `streaming_kernels` has all of them)
        width0 = 9 if args.config == "Lb9" else 6
        top = max(cands)[0]
        nnzb0 = count_offdiag_blocks(local)

        def frac_of(name):
            ms_k, n_k = kernels[name]
            return algorithmic_bytes(name, prob.n_obs, prob.n_points, n_cams, nnzb0, width0) / (ms_k * 1e-3 / n_k)
        dom = min((k for ms_k, k in cands if ms_k >= 0.9 * top), key=frac_of)
        ms, launches = kernels[dom]
        avg_s = ms * 1e-3 / launches
        width = 9 if args.config == "Lb9" else 6
        alg = algorithmic_bytes(dom, prob.n_obs, prob.n_points, n_cams, count_offdiag_blocks(local), width)
        ach = alg / avg_s / 1e9
        traffic = None
        traffic_source = None
        pmc = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.config}.json")
        prof = None
        if world == 1 and rank == 0 and not args.no_extras:
            try:
                prof = self_profile(args.config, args.solver)
            except Exception:
                prof = None
        if prof is not None and prof.get("traffic_bytes_per_launch"):
            table = prof["traffic_bytes_per_launch"]
            hits = [v for k, v in table.items() if k == dom or k.startswith(dom + "<")]
            traffic = sum(hits) if hits else None
            traffic_source = prof["source"] + "; " + prof["traffic_correction"]
        elif world == 1 and os.path.exists(pmc):      # separate rocprofv3 --pmc passes (tools/pmc_summary.py), bytes per launch
            table = json.load(open(pmc))          # template instantiations of one kernel (k_schur_pairs<true|false>) make up one pass
            hits = [v for k, v in table.items() if k == dom or k.startswith(dom + "<")]
            traffic = sum(hits) if hits else None
            # NOT measured in this run: PMC counters need their own rocprofv3 passes (tools/make_profiles.sh)
            traffic_source = f"profiles/pmc_traffic_{args.config}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run, see profiles/*_pmc_traffic.md)"
        roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_source, "kernel": dom, "avg_launch_us": avg_s * 1e6, "launches": launches,
                    "algorithmic_bytes_per_launch": alg}
        # every HBM-streaming kernel of the run by the same rule (the dominant one can change from round to round: in round 1
        # it was k_schur_pairs at 0.39; its fraction stays visible here whichever kernel leads)
        per_kernel = {}
        for ms_k, k in sorted(cands, reverse=True):
            n_k = kernels[k][1]
            b_k = algorithmic_bytes(k, prob.n_obs, prob.n_points, n_cams, count_offdiag_blocks(local), width)
            per_kernel[k] = {"avg_launch_us": ms_k * 1e3 / n_k, "launches": n_k, "algorithmic_bytes_per_launch": b_k,
                             "frac": b_k / (ms_k * 1e-3 / n_k) / 1e9 / HBM_PEAK_GBS}
        if prof is not None:
            # what rocprofv3 saw in the child run of this job: average duration per kernel (must agree with the HIP-event figures
            # above) and, when the counter passes ran, HBM bytes per launch of every kernel
            roofline["rocprofv3"] = prof
            hits = [v for k, v in prof["kernel_avg_us"].items() if k == dom or k.startswith(dom + "<")]
            calls = [prof["kernel_calls"][k] for k in prof["kernel_avg_us"] if k == dom or k.startswith(dom + "<")]
            if hits:       # per pass over the items = sum over the launches of the pass (one per Gram bucket)
                n_pass = max(1, min(calls))
                roofline["rocprofv3_avg_launch_us"] = sum(h * c for h, c in zip(hits, calls)) / n_pass
            if prof.get("traffic_bytes_per_launch"):
                for k, rec in per_kernel.items():
                    th = [v for kk, v in prof["traffic_bytes_per_launch"].items() if kk == k or kk.startswith(k + "<")]
                    rec["traffic"] = sum(th) if th else None
        elif world == 1 and os.path.exists(pmc):
            # the whole per-kernel PMC table (bytes per launch, FETCH_SIZE + WRITE_SIZE passes of an earlier run), and the measured
            # traffic next to the algorithmic bytes of every streaming kernel: `traffic` above is the dominant kernel's entry only
            table = json.load(open(pmc))
            roofline["traffic_table"] = {"source": traffic_source, "bytes_per_launch": table}
            for k, rec in per_kernel.items():
                hits = [v for kk, v in table.items() if kk == k or kk.startswith(k + "<")]
                rec["traffic"] = sum(hits) if hits else None
        roofline["streaming_kernels"] = per_kernel
        # the whole LM iteration against BASELINE.md section 4: B_iter(0) = B_lin + B_prep + B_back, plus the explicit-S terms
        # (the nnzb off-diagonal 6x6 blocks written once; SURVEY 8d's B_S would also count a second read of J, N_obs*144,
        # which the fused S assembly does not do: reported separately)
        if last.linear_solver_used == 1 and world == 1 and width == 6:
            nnzb = count_offdiag_blocks(local)
            b_iter = (prob.n_obs * (184 + 160 + 168) + prob.n_points * (24 + 72 + 144) + n_cams * (56 + 216 + 160) + nnzb * 288)
            t_iter = dt / max(iters, 1)
            roofline["iteration"] = {
                "bytes": b_iter, "bytes_with_B_S_second_read_of_J": b_iter + prob.n_obs * 144, "t_iter_us": t_iter * 1e6,
                "achieved": b_iter / t_iter / 1e9, "frac": b_iter / t_iter / 1e9 / HBM_PEAK_GBS,
                "frac_with_B_S": (b_iter + prob.n_obs * 144) / t_iter / 1e9 / HBM_PEAK_GBS,
                "accounting": "BASELINE.md section 4 B_iter(0) + nnzb*288 (explicit block-sparse S, exact Cholesky: K = 0); "
                              "t_iter = timed region / LM iterations (includes the iteration-0 linearisations)"}

    per_rank = None
    if world > 1:
        # the dominant streaming kernel on EVERY rank (each streams its own shard): rank 0 reports the list next to its own block
        mine = None if roofline is None else {"rank": rank, "kernel": roofline["kernel"], "avg_launch_us": roofline["avg_launch_us"],
                                              "algorithmic_bytes_per_launch": roofline["algorithmic_bytes_per_launch"], "frac": roofline["frac"],
                                              "points": int(prob.n_points), "obs": int(prob.n_obs)}
        try:
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            per_rank = gathered
        except Exception:
            per_rank = None
    if rank == 0:
        n_res = 2 * n_obs
        out = {
            "metric": "BA iterations/sec x (cams+points)", "value": iters * (n_cams + n_points) / dt,
            "unit": "cam-pts*iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("synthetic clustered photo collection (shape and size of BASELINE config 5) " if args.config == "T" else "synthetic BAL-style ")
                                   + f"{args.config}: {n_cams} cams / {n_points} points / {n_obs} obs, "
                                   f"GBA accurate (ba_solver.cc:626-629), seed {cfg['seed']}"
                                   + (f"; {args.scaling} scaling: " + ("this problem split over the ranks" if args.scaling == "strong"
                                                                      else f"{world} config-sized point shards over the same cameras") if world > 1 else ""),
                       "parallelism": (f"points sharded x{world} ({prob.n_points} points / {prob.n_obs} obs on rank 0), cameras replicated, "
                                       f"2 RCCL all-reduces per accepted LM step") if world > 1 else "single GPU",
                       "linear_solver": "cholesky (explicit reduced camera matrix, exact)" if last.linear_solver_used == 1
                       else f"implicit-Schur PCG tol {opt.pcg_tolerance:g}"},
            "lm_iterations_per_step": iters / args.steps, "pcg_iterations_per_step": pcg / args.steps,
            "final_rmse_px": math.sqrt(last.final_cost / n_res),
            "initial_rmse_px": math.sqrt(last.initial_cost / n_res),
            "termination_reason": last.termination_reason,
            "roofline": roofline, "cpu_baseline": None, "kernels": kernel_table,
        }
        if per_rank is not None and roofline is not None:
            roofline["per_rank"] = per_rank
        if args.config == "Lb9":
            out["config"]["workload"] += "; bal9 mode: every camera's {f, k1, k2} variable, 9-wide camera blocks"
        if world == 1 and not args.no_cpu:
            res = cpu_baseline(arr, n_cams, n_points, opt.max_iterations)
            if res is not None:
                base, cpu_prob = res
                base["gpu_vs_cpu"] = out["value"] / base["value"]
                if "more_threads" in base:
                    base["more_threads"]["gpu_vs_cpu"] = out["value"] / base["more_threads"]["value"]
                base["rmse_diff_px"] = abs(base["final_rmse_px"] - out["final_rmse_px"])
                base["max_cam_param_diff"] = float(max(np.abs(cpu_prob["cam_q"] - q).max(), np.abs(cpu_prob["cam_t"] - t).max()))
                base["max_centre_diff_gauge_aligned"] = gauge_aligned_centre_diff(cpu_prob["cam_q"], cpu_prob["cam_t"], q, t)
                if args.config != "Lb9":
                    base["parity"] = parity_block(arr, q, t, P, cpu_prob)
                if args.config == "Lb9":          # the variable intrinsics {f, k1, k2} of both results
                    gi = ctx.download_intrinsics()
                    var = (np.asarray(arr["cam_const"]) & 4) != 0
                    base["max_rel_focal_diff"] = float(np.abs(gi[var, 0] / cpu_prob["intr_params"][var, 0] - 1).max())
                    base["max_distortion_diff"] = float(np.abs(gi[var, 1:3] - cpu_prob["intr_params"][var, 1:3]).max())
                if args.config == "L":
                    try:
                        base["parity_workload"] = parity_workload(opt)
                    except Exception as exc:       # a side measurement: never at the cost of the headline line
                        base["parity_workload"] = {"error": str(exc)}
                out["cpu_baseline"] = base
    ctx.close()
    if rank == 0:
        if world == 1 and not args.no_extras:
            out["host_inclusive"] = host_inclusive(arr, opt, n_cams, n_points)
            try:
                out["mfma_utilisation"] = mfma_utilisation() if args.config in ("L", "S") else None
            except Exception as exc:       # the side measurement must never cost the headline line
                out["mfma_utilisation"] = {"error": str(exc)}
        try:        # RCCL prints its version banner through C stdio: flush it first, so that the JSON line is the last line on stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
