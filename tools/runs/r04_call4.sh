#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_bal9.py -x -q 2>&1 | tail -15 | tee $OUT/pytest_bal9.txt
timeout 300 python bench.py --config Lb9 --steps 3 --warmup 1 --no-extras 2> $OUT/bench_Lb9.err | tail -1 > $OUT/bench_Lb9.json
python - $OUT/bench_Lb9.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d.get("kernels", {})
print("Lb9 ms/solve", d["ms_per_step"], "it", d["lm_iterations_per_step"], "rmse", d["final_rmse_px"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
print(" ".join(f"{n}={v['ms'] * 1e3 / max(v['launches'], 1):.1f}us x{v['launches']}" for n, v in k.items()))
cb = d.get("cpu_baseline") or {}
print({kk: cb.get(kk) for kk in ("gpu_vs_cpu", "rmse_diff_px", "max_cam_param_diff", "max_rel_focal_diff", "max_distortion_diff", "iterations")})
PY
timeout 300 bash tools/quick_prof.sh Lb9 r04d_Lb9 > /dev/null 2>&1; head -24 gpurun_out/prof_r04d_Lb9/kernel_stats_table.md
