#!/usr/bin/env python3
"""Strong-scaling projection of one LM iteration from a ONE-GPU profile (rocprofv3 kernel table written by
tools/rocprof_summary.py, e.g. profiles/r02_L_kernel_stats.md) — what `bench.py --gpus N --scaling strong` should show.

Model:
  t_iter(N) = sum over kernels that stream over the rank's OWN tracks / N       (points are sharded: k_schur_pairs, k_linearize,
              k_backsub, k_point_prep, k_cost, and the per-camera sums over the rank's scatter entries)
            + sum over kernels that every rank repeats                          (exact factorisation of the reduced camera system,
              tile fill, backward substitution, the linearisation tail)
            + launch gaps / host hand-off (measured: t_iter(1) - sum of kernel times)
            + collectives per accepted LM step: 2 all-reduces (96 KB + 1.1 MB at config L), alpha + 2 (N-1)/N bytes / beta
usage: python tools/scaling_projection.py profiles/r02_L_kernel_stats.md [--iters-per-solve 13 --solves 5] [--alpha-us 25 --beta-gbs 100]"""
import argparse
import json
import re

SHARDED = ("k_schur_pairs", "k_linearize", "k_backsub", "k_point_prep", "k_cost", "k_chol_segsum", "k_schur_prep", "k_schur_matvec", "k_cam_segsum")
IGNORED = ("__amd_rocclr", "k_fill", "k_scale_from_norms", "k_cam_lin", "devpack::", "rocprim::", "k_pair_sources")       # set-up of a solve / of the problem, not part of the iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("--lm-iterations", type=int, default=None, help="LM iterations covered by the table (default: calls of k_schur_pairs)")
    ap.add_argument("--alpha-us", type=float, default=25.0, help="latency of one small RCCL all-reduce over xGMI")
    ap.add_argument("--beta-gbs", type=float, default=100.0, help="effective all-reduce bandwidth per GPU")
    ap.add_argument("--bytes", type=float, default=96e3 + 224e3 + 0.86e6, help="all-reduced bytes per LM iteration (config L)")
    ap.add_argument("--collectives", type=int, default=2)
    ap.add_argument("--t-iter-us", type=float, default=None, help="measured one-GPU time per LM iteration (default: from the bench line in the file)")
    args = ap.parse_args()
    text = open(args.table).read()
    rows = re.findall(r"^\| `([^`]+)` \| (\d+) \| ([\d.]+) \|", text, re.M)
    kern = {n: (int(c), float(ms)) for n, c, ms in rows}
    iters = args.lm_iterations or next((c for n, (c, _) in kern.items() if "k_schur_pairs" in n or "k_schur_prep" in n), 1)
    sharded = sum(ms for n, (c, ms) in kern.items() if any(s in n for s in SHARDED)) * 1e3 / iters
    ignored = sum(ms for n, (c, ms) in kern.items() if any(s in n for s in IGNORED) or not n.strip()) * 1e3 / iters
    repl = sum(ms for n, (c, ms) in kern.items()) * 1e3 / iters - sharded - ignored
    t1 = args.t_iter_us
    if t1 is None:
        m = re.search(r'"ms_per_step": ([\d.]+).*?"lm_iterations_per_step": ([\d.]+)', text)
        t1 = float(m.group(1)) * 1e3 / float(m.group(2)) if m else sharded + repl
    gaps = max(0.0, t1 - sharded - repl)
    print(f"per LM iteration on one GPU: {t1:.0f} us = sharded kernels {sharded:.0f} + replicated kernels {repl:.0f} + gaps/host {gaps:.0f}")
    print("| GPUs | sharded | replicated | gaps | all-reduce | t_iter us | speed-up |")
    print("|---:|---:|---:|---:|---:|---:|---:|")
    out = {}
    for n in (1, 2, 4, 8):
        comm = 0.0 if n == 1 else args.collectives * args.alpha_us + 2.0 * (n - 1) / n * args.bytes / (args.beta_gbs * 1e3)
        t = sharded / n + repl + gaps + comm
        out[n] = t
        print(f"| {n} | {sharded / n:.0f} | {repl:.0f} | {gaps:.0f} | {comm:.0f} | {t:.0f} | {out[1] / t:.2f} |")
    # weak scaling (bench.py --scaling weak: N config-sized point shards over the same cameras): the sharded kernels keep their
    # one-GPU time, the job does N times the work
    print()
    print("weak scaling (N x the points over the same cameras):")
    print("| GPUs | t_iter us | throughput vs 1 GPU |")
    print("|---:|---:|---:|")
    weak = {}
    for n in (1, 2, 4, 8):
        comm = 0.0 if n == 1 else args.collectives * args.alpha_us + 2.0 * (n - 1) / n * args.bytes / (args.beta_gbs * 1e3)
        t = sharded + repl + gaps + comm
        weak[n] = n * out[1] / t
        print(f"| {n} | {t:.0f} | {weak[n]:.2f} |")
    print(json.dumps({"t_iter_us": out, "speedup_8": out[1] / out[8], "weak_throughput_8": weak[8]}))


if __name__ == "__main__":
    main()
