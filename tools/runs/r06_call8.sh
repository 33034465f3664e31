#!/bin/bash
# round 6, call 8: does the second stream's chunk launch overlap the main stream at config T? (kernel trace of a short run: per-queue
# timelines of a few levels) + where a slow pose refinement of the mapper replay spends its time
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c8
mkdir -p $OUT
cd $ROOT
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; grep -E "slow call|RefineFramePose" $OUT/slow_calls.txt | head -12
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/trace_window.py $DB "k_lv_factor<false>" 3000 40 > $OUT/window_T.txt; cat $OUT/window_T.txt
rm -rf $OUT/tr
