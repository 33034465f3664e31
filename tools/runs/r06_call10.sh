#!/bin/bash
# round 6, call 10: per-context camera-run tables (k_gram_runs) instead of ballot loops in k_linearize / k_schur_pairs: tests on ragged
# shapes, bench R / L / K; pinned staging in the track filter: does the slow pose refinement after a whole-map filter go away?
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c10
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x -k "shape_tiles or poison or ragged or one_gram or random_problem or golden or filter" 2>&1 | tail -6
for cfg in R L K; do
  timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | grep '^{"metric"' > $OUT/bench_${cfg}.json
  echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${cfg}.json)"
done
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; grep -E "mapper_main|slow call" $OUT/slow_calls.txt | tail -14
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config R --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/iteration_gaps.py $DB 20 > $OUT/gaps_R_20.txt; head -5 $OUT/gaps_R_20.txt; tail -5 $OUT/gaps_R_20.txt
rm -rf $OUT/tr
