#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/make_profiles.sh on the GPU box) into the committed profiles/ files.

usage: python tools/finish_profiles.py r01
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    rd = lambda n: open(os.path.join(src, n)).read()        # noqa: E731
    line = rd("stats_bench_line.json").strip()
    if not line.startswith("{"):
        line = [l for l in rd("stats_bench.log").splitlines() if l.startswith('{"metric"')][-1]
    nproc = rd("nproc.txt").strip()
    with open(os.path.join(dst, f"{tag}_L_kernel_stats.md"), "w") as f:
        f.write(f"# Round {tag[1:]} — rocprofv3 kernel-trace summary, bench.py --config L --no-cpu --steps 2 (MI355X, 1 GPU)\n\n"
                "Command (tools/make_profiles.sh): `rocprofv3 --kernel-trace --stats -d DIR -o stats -- python bench.py --config L --no-cpu --steps 2`\n"
                "(4 solves: 1 warm-up, 2 timed, 1 profiled with HIP events; 13 LM iterations each).  Per-kernel table from the rocpd\n"
                "database with `tools/rocprof_summary.py`.\n\nbench.py line of the same (profiled) run:\n\n```\n" + line + "\n```\n\n" + rd("kernel_stats_table.md"))
    with open(os.path.join(dst, f"{tag}_L_pmc_traffic.md"), "w") as f:
        f.write(f"# Round {tag[1:]} — HBM traffic per kernel from rocprofv3 PMC counters, bench.py --config L (MI355X, 1 GPU)\n\n"
                "Two separate passes (tools/make_profiles.sh; TCC slots do not fit both counters):\n"
                "`rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --config L --no-cpu --steps 1 --warmup 0` and the same with `--pmc WRITE_SIZE`.\n"
                "Corrections and their calibration: see the docstring of `tools/pmc_summary.py`.  Averages per launch.\n\n" + rd("pmc_table.md"))
    shutil.copy(os.path.join(src, "pmc_traffic_L.json"), os.path.join(dst, "pmc_traffic_L.json"))
    with open(os.path.join(dst, f"{tag}_bench_lines.md"), "w") as f:
        f.write(f"# Round {tag[1:]} — bench.py lines (MI355X, 1 GPU, host with {nproc} cores), `python bench.py --config C --steps 5 --warmup 2`\n")
        for cfg, note in (("L", "default workload: `python bench.py --steps 20 --warmup 5`, incl. the rocprofv3 passes the run launches itself"), ("S", ""),
                          ("LP", "config 4 as a parity workload: + 24 hub frames x 50 distant landmarks; literal north-star bounds in cpu_baseline"), ("K", "KITTI-00-sized sequential problem"),
                          ("X", "5000 cameras: 30 000 camera unknowns on the exact Cholesky path"),
                          ("R", "ragged tracks: windows of 8 frames, 35 % missed detections"),
                          ("U", "random visibility, dense reduced camera matrix; no CPU leg"),
                          ("D", "2000 cameras, random visibility: the dense limit of the exact path (panel schedule); no CPU leg"),
                          ("V", "3000 cameras, random visibility: implicit-Schur PCG; no CPU leg"),
                          ("L0", "config 4 with SURVEY Appendix D read literally: radius-40 ring, no triangulation-angle filter"),
                          ("T", "BASELINE config 5 at its size: 7500 photos in viewpoint clusters / 1.8M points / 8.1M observations, shuffled ids; exact path, nested dissection of the camera graph (round 3: reverse Cuthill-McKee chain); no CPU leg"),
                          ("T_pcg", "the same through the implicit-Schur PCG (the only path at this size until round 2)"),
                          ("Lb9", "config 4 in bal9 mode: every camera's {f, k1, k2} variable, 9-wide camera blocks (round 4: Gram tiles, k9_pairs_gram); CPU leg = the C restatement with CW = 9; parity incl. the refined intrinsics in cpu_baseline"),
                          ("M", "mapper-shaped replay through the BASolver adapter: a different metric (wall time of the BA calls of a 300-frame incremental reconstruction)")):
            if not os.path.exists(os.path.join(src, f"bench_{cfg}.json")):
                continue
            f.write(f"\n## config {cfg}" + (f" ({note})" if note else "") + "\n```\n" + rd(f"bench_{cfg}.json").strip() + "\n```\n")
            d = json.loads(rd(f"bench_{cfg}.json"))
            b = d.get("cpu_baseline") or {}
            if cfg == "M":
                f.write(f"\n{d['value']:.1f} ms of BA calls, {d['ms_per_step']:.1f} ms for the whole replay.\n")
                continue
            f.write(f"\n{d['ms_per_step']:.2f} ms per solve ({d['lm_iterations_per_step']:.0f} LM iterations), {d['value']:.3e} {d['unit']}")
            if not b:
                f.write(".\n")
                continue
            f.write(f"; CPU port {b.get('cores')} threads: {b.get('value', 0):.3e} ({b.get('gpu_vs_cpu', 0):.0f}x)")
            if "more_threads" in b:
                m = b["more_threads"]
                f.write(f", {m['cores']} threads: {m['value']:.3e} ({m['gpu_vs_cpu']:.0f}x)")
            f.write(f"; final RMSE difference to the CPU port {b.get('rmse_diff_px', float('nan')):.1e} px.\n")
    if os.path.exists(os.path.join(src, "pmc_mix.md")):
        with open(os.path.join(dst, f"{tag}_L_instruction_mix.md"), "w") as f:
            f.write(f"# Round {tag[1:]} — instruction mix and pipe occupancy per kernel, rocprofv3 PMC (SQ counters), bench.py --config L\n\n"
                    "Three passes (tools/pmc_mix.sh; counters only + kernel trace).  One row per kernel = average over its dispatches of the per-shader-engine "
                    "counter rows (32 rows per dispatch: multiply by 32 for a whole launch; SQ_WAVES x 32 = waves of a launch).\n\n" + rd("pmc_mix.md"))
    if os.path.exists(os.path.join(src, "kernel_stats_table_D.md")):
        with open(os.path.join(dst, f"{tag}_D_kernel_stats.md"), "w") as f:
            f.write(f"# Round {tag[1:]} — rocprofv3 kernel-trace summary, bench.py --config D --no-cpu --steps 1 --warmup 1 (dense reduced solve: 12 000 camera unknowns, 188 tile columns)\n\n"
                    "k_panel2_part = 128x128 macro-tile left-looking update of two tile columns, k_ll_update_reduce = fixed-order sum of its partial tiles, "
                    "k_lv_factor = pivot factorisation + triangular solve of one tile column, k_bwd = backward substitution.\n\n" + rd("kernel_stats_table_D.md"))
            if os.path.exists(os.path.join(src, "mfma_rate.txt")):
                f.write("\n## Sustained rate of v_mfma_f64_16x16x4_f64 with nothing else in the loop (tools/bench_mfma.hip)\n\n```\n" + rd("mfma_rate.txt") + "```\n")
    for cfgk, title in (("R", "config R (ragged tracks: windows of 8 frames, 35 % missed detections)"), ("Lb9", "config Lb9 (config 4 in bal9 mode)"),
                        ("T", "config T (BASELINE config 5's shape: 7500 photos in viewpoint clusters, exact path on the dissected camera graph)")):
        if os.path.exists(os.path.join(src, f"kernel_stats_table_{cfgk}.md")):
            with open(os.path.join(dst, f"{tag}_{cfgk}_kernel_stats.md"), "w") as f:
                f.write(f"# Round {tag[1:]} — rocprofv3 kernel-trace summary, bench.py --config {cfgk} --no-cpu --no-extras --steps 2: {title}\n\n" + rd(f"kernel_stats_table_{cfgk}.md"))
    for extra in ("lba_phases.txt", "lba_timing.txt", "probe.txt", "mapper_trace.txt", "potrf.txt", "lat.txt", "pack_crossover.txt", "pack_phases.txt", "adapter_timing.txt", "pipes.txt"):
        if os.path.exists(os.path.join(src, extra)):
            shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{extra}"))
    # strong / weak scaling MODEL from the one-GPU kernel table (tools/scaling_projection.py)
    import subprocess
    table = os.path.join(dst, f"{tag}_L_kernel_stats.md")
    if os.path.exists(table):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "scaling_projection.py"), table],
                             capture_output=True, text=True).stdout
        with open(os.path.join(dst, f"{tag}_scaling_projection.md"), "w") as f:
            f.write(f"# Round {tag[1:]} — strong- and weak-scaling projection of config L from the one-GPU profile (profiles/{tag}_L_kernel_stats.md)\n\n"
                    f"`python tools/scaling_projection.py profiles/{tag}_L_kernel_stats.md` — a MODEL (DESIGN.md section 7), not a measurement: no multi-GPU box was available "
                    "to this round; the driver's SCALE run is the measurement.  alpha = 25 us per small all-reduce, beta = 100 GB/s per GPU.  Kernels of the "
                    "problem set-up (device packing, sorts) are left out of the iteration.\n\n```\n" + out + "```\n\n"
                    "Why strong scaling stops at ~1.5x: the streamed kernels (S assembly, linearisation, back-substitution, per-camera sums) divide by N, "
                    "the exact factorisation of the reduced camera system (replicated on every rank: its inputs are all-reduced), the tails and two collectives "
                    "per iteration do not.  The north star's >= 6x at 8 GPUs on 500 000 points would need a per-iteration critical path of ~80 us.\n")
    print("profiles written for", tag)


if __name__ == "__main__":
    main()
