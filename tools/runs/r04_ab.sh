#!/bin/bash
# Round-4 A/B of library builds (run through gpurun from the repo root): one bench line per (build variant, configuration).
#   gpurun -- 'bash tools/runs/r04_ab.sh tag "L S" "default potrf00 potrf01"'
set -u
TAG=${1:-ab}; CFGS=${2:-"L S"}; LIBS=${3:-"default"}; STEPS=${4:-20}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for cfg in $CFGS; do
  for v in $LIBS; do
    if [ "$v" = "default" ]; then L=""; else L="XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba_$v.so"; fi
    env $L python bench.py --config $cfg --no-cpu --no-extras --steps $STEPS --warmup 3 2> $OUT/bench_${cfg}_$v.err | tail -1 > $OUT/bench_${cfg}_$v.json
    python - "$OUT/bench_${cfg}_$v.json" "$cfg" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("kernels", {})
    print(f"{sys.argv[2]:3s} {sys.argv[3]:10s} ms/solve {d['ms_per_step']:.3f}  it {d['lm_iterations_per_step']}  rmse {d['final_rmse_px']:.9f}  " + " ".join(f"{n}={v['ms'] * 1e3 / max(v['launches'], 1):.1f}us" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e)
PY
  done
done
