#!/bin/bash
# round 5, call 3: two-level PCG preconditioner (gauge coarse space) — tests, then config V / T_pcg / U with and without; first-call timing with the warm-up
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c3
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_adapter.py -m gpu -q -x -k "pcg or PCG or long_items or adapter or variant or poison" 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
for co in 1 0; do
  XRSFM_BA_PCG_COARSE=$co XRSFM_BENCH_SELFPROF=0 timeout 600 python bench.py --config V --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_V_co$co.err | tail -1 > $OUT/bench_V_co$co.json
  XRSFM_BA_PCG_COARSE=$co XRSFM_BENCH_SELFPROF=0 timeout 900 python bench.py --config T --steps 1 --warmup 0 --no-cpu --no-extras --solver pcg 2> $OUT/bench_Tpcg_co$co.err | tail -1 > $OUT/bench_Tpcg_co$co.json
  for f in V Tpcg; do python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${f}_co$co.json").read())
    print("$f coarse=$co ms_per_step", round(d["ms_per_step"], 2), "lm_it", d.get("lm_iterations_per_step"), "pcg_it", d.get("pcg_iterations_per_step"), "rmse", d.get("final_rmse_px"), {k: (round(v["ms"], 2), v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("$f $co failed", e)
PY
  done
done
python tools/adapter_timing.py L > $OUT/adapter_timing_L.txt 2>&1; cat $OUT/adapter_timing_L.txt
python tools/adapter_timing.py K > $OUT/adapter_timing_K.txt 2>&1; cat $OUT/adapter_timing_K.txt
