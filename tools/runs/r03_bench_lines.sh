#!/bin/bash
# Round-3 developer aid (GPU box): one-line summaries of bench.py runs.   bash tools/runs/r03_bench_lines.sh tag "L S R" [steps]
set -u
TAG=${1:-b}; CFGS=${2:-"L S R"}; STEPS=${3:-20}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cfg in $CFGS; do
  python bench.py --config $cfg --no-cpu --no-extras --steps $STEPS --warmup 2 2> $OUT/bench_$cfg.err | tail -1 > $OUT/bench_$cfg.json
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("kernels", {})
    print(f"{sys.argv[2]:3s} ms/solve {d['ms_per_step']:.3f}  it {d['lm_iterations_per_step']}  solver {d['config']['linear_solver'][:8]}  " + " ".join(f"{n}={v['ms'] * 1e3 / max(v['launches'], 1):.1f}us" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
