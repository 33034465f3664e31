#!/bin/bash
# round 5, call 15: launch sequence and idle gaps of one LM iteration at config L (rocprofv3 kernel trace, tools/iteration_gaps.py)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c15
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/bench.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/iteration_gaps.py $DB 20 > $OUT/gaps_20.txt; cat $OUT/gaps_20.txt
python $ROOT/tools/iteration_gaps.py $DB 22 | tail -1
rm -rf $OUT/tr
