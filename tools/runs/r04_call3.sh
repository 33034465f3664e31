#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04c
mkdir -p $OUT
cd $ROOT
timeout 60 tools/bench_lat > $OUT/lat.txt 2>&1; cat $OUT/lat.txt
timeout 600 python -m pytest tests/test_gpu_hardening.py -k "one_launch_backward" -x -q 2>&1 | tail -5 | tee $OUT/pytest.txt
for cfg in L R LP S; do
  for v in 1 0; do
    XRSFM_BA_BWD_ALL=$v timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 20 --warmup 3 2> $OUT/bench_${cfg}_$v.err | tail -1 > $OUT/bench_${cfg}_$v.json
    python - "$OUT/bench_${cfg}_$v.json" "$cfg" "BWD_ALL=$v" <<'PY'
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
