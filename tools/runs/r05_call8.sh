#!/bin/bash
# round 5, call 8: backward-substitution lists walked from the root side — bit-identity of the one-launch form, then config T with the
# one-launch backward substitution (XRSFM_BA_BWD_ALL=1) against the per-level chunk launches (default for deep trees), and R / LP
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_parity.py -m gpu -q -x -k "backward_substitution or unordered_collection or dissect" 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt
for ba in default 1; do
  if [ $ba = default ]; then unset XRSFM_BA_BWD_ALL; else export XRSFM_BA_BWD_ALL=$ba; fi
  XRSFM_BENCH_SELFPROF=0 timeout 600 python bench.py --config T --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_T_$ba.err | tail -1 > $OUT/bench_T_$ba.json
  python - <<PY
import json
d = json.loads(open("$OUT/bench_T_$ba.json").read())
print("T bwd_all=$ba ms_per_step", round(d["ms_per_step"], 2), "lm_it", d.get("lm_iterations_per_step"), "rmse", d.get("final_rmse_px"), {k: (round(v["ms"], 2), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
done
unset XRSFM_BA_BWD_ALL
for cfg in R LP X; do
  XRSFM_BENCH_SELFPROF=0 timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 5 --warmup 2 2> $OUT/bench_${cfg}.err | tail -1 > $OUT/bench_${cfg}.json
  python - <<PY
import json
d = json.loads(open("$OUT/bench_${cfg}.json").read())
print("$cfg ms_per_step", round(d["ms_per_step"], 3), "lm_it", d.get("lm_iterations_per_step"), {k: (round(v["ms"], 3), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
done
