#!/bin/bash
# round 5, call 9: slot-indexed loads (Jp, rt, u, v) requested with the slot record, before the `valid` test — per-kernel times at L (rocprofv3) and bench lines L / R / K / V
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c9
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -m gpu -q -x -k "linearize or golden or poison or shape or long_items" 2>&1 | tail -3
for cfg in L R K; do
  XRSFM_BENCH_SELFPROF=0 timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 10 --warmup 3 2> $OUT/bench_${cfg}.err | tail -1 > $OUT/bench_${cfg}.json
  python - <<PY
import json
d = json.loads(open("$OUT/bench_${cfg}.json").read())
print("$cfg ms_per_step", round(d["ms_per_step"], 3), "lm_it", d.get("lm_iterations_per_step"), {k: (round(v["ms"], 3), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 2 > $OUT/stats_bench.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats_table.md > /dev/null
rm -rf $OUT/stats
head -14 $OUT/kernel_stats_table.md
# RESULT (not adopted): requesting the slot-indexed streams ahead of the `valid` test made all three streaming kernels slightly slower at L —
# k_schur_pairs 99.6 -> 101.8 us, k_backsub 66.4 -> 69.2, k_linearize 79.0 -> 80.3 (the loads are issued one round trip early but hold their registers longer)
