#!/bin/bash
# Round-3 A/B of the fused schedules (run through gpurun from the repo root): bench lines per switch and configuration.
#   gpurun -- 'bash tools/runs/r03_ab.sh tag "L S R"'
set -u
TAG=${1:-ab}; CFGS=${2:-"L S R"}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cfg in $CFGS; do
  i=0
  for v in "A=1" "XRSFM_BA_PREP_FUSED=0" "XRSFM_BA_FILL_FUSED=0" "XRSFM_BA_PREP_FUSED=0 XRSFM_BA_FILL_FUSED=0"; do
    env $v python bench.py --config $cfg --no-cpu --no-extras --steps 20 --warmup 3 2> $OUT/bench_${cfg}_$i.err | tail -1 > $OUT/bench_${cfg}_$i.json
    python - "$OUT/bench_${cfg}_$i.json" "$cfg" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("kernels", {})
    print(f"{sys.argv[2]:3s} {sys.argv[3]:48s} ms/solve {d['ms_per_step']:.3f}  it {d['lm_iterations_per_step']}  " + " ".join(f"{n}={v['ms'] * 1e3 / max(v['launches'], 1):.1f}us" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
    i=$((i+1))
  done
done
