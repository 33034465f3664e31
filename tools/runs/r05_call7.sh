#!/bin/bash
# round 5, call 7: PV 4 as the default pivot-block factorisation (one call site) — microbench, stamps, a slice of the suite, L / R / LP / K / S / T
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c7
mkdir -p $OUT
cd $ROOT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1; cat $OUT/potrf.txt
XBA_TL_TAG=pv4 timeout 300 python tools/timeline.py L > $OUT/timeline_pv4.txt 2>&1; grep -A17 "k_lv_factor" $OUT/timeline_pv4.txt
timeout 900 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_parity.py -m gpu -q -x -k "backward_substitution or headline or config4_parity or shape or poison or bal9 or golden" 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt
for cfg in L R LP K S; do
  XRSFM_BENCH_SELFPROF=0 timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 10 --warmup 3 2> $OUT/bench_${cfg}.err | tail -1 > $OUT/bench_${cfg}.json
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${cfg}.json").read())
    print("$cfg ms_per_step", round(d["ms_per_step"], 3), "lm_it", d.get("lm_iterations_per_step"), "rmse", d.get("final_rmse_px"), {k: (round(v["ms"], 3), v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("$cfg failed", e)
PY
done
XRSFM_BENCH_SELFPROF=0 timeout 600 python bench.py --config T --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_T.err | tail -1 > $OUT/bench_T.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_T.json").read())
print("T ms_per_step", round(d["ms_per_step"], 2), "lm_it", d.get("lm_iterations_per_step"), {k: (round(v["ms"], 2), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
