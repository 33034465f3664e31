#!/bin/bash
# round 6, call 6: main Gram launch issued before the side-stream launches; k_lin_tail stage 2 with loads in flight; which pose
# refinements of the mapper replay are slow (LM step counts)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c6
mkdir -p $OUT
cd $ROOT
for cfg in R L K S; do
  timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 8 --warmup 2 2>/dev/null | grep '^{"metric"' > $OUT/bench_${cfg}.json
  echo "$cfg $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${cfg}.json)"
done
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; cat $OUT/slow_calls.txt | head -40
cd /tmp && export TMPDIR=/tmp
for cfg in R L; do
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/iteration_gaps.py $DB 20 > $OUT/gaps_${cfg}_20.txt; head -7 $OUT/gaps_${cfg}_20.txt; tail -6 $OUT/gaps_${cfg}_20.txt
rm -rf $OUT/tr
done
