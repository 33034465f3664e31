#!/bin/bash
# round 5, call 2: split levels whose partial tiles the factor kernel adds itself (no k_ll_update_reduce launch) — bit-identity, then L / R / LP / K / S with and without
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c2
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -k "split_level_sums or backward_substitution" 2>&1 | tail -5 > $OUT/pytest.txt
cat $OUT/pytest.txt
for cfg in L R LP K S; do
  for sf in 0 1; do
    XRSFM_BA_SPLIT_SUM=$sf XRSFM_BENCH_SELFPROF=0 timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 10 --warmup 3 2> $OUT/bench_${cfg}_sf$sf.err | tail -1 > $OUT/bench_${cfg}_sf$sf.json
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench_${cfg}_sf$sf.json").read())
    print("$cfg split_sum=$sf ms_per_step", round(d["ms_per_step"], 3), "lm_it", d.get("lm_iterations_per_step"), "rmse", d.get("final_rmse_px"), {k: (round(v["ms"], 3), v["launches"]) for k, v in d.get("kernels", {}).items()})
except Exception as e:
    print("$cfg $sf failed", e)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 2 > $OUT/stats_bench.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats_table.md > /dev/null
rm -rf $OUT/stats
head -30 $OUT/kernel_stats_table.md
# RESULT (sums of a split level inside k_lv_factor, removed again): bit-identical, but slower — L 6.42 -> 6.54 ms, R 17.2 -> 18.0, LP 14.9 -> 16.1,
# K 15.8 -> 16.3, S 2.17 -> 2.23: the factor workgroup of a thin level pays 8-11 us for 2-5 dependent round trips of 64 loads (k_lv_factor 16.5 ->
# 25.6 us) where the sum launch spreads the same loads over 16 workgroups per target (5.7 us + a kernel boundary).
