#!/bin/bash
# round 4: the complete GPU suite on the final library; config T line and kernel table
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_r04
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_final.txt
python bench.py --config T --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_T.err | tail -1 > $OUT/bench_T.json
python -c "
import json; d = json.load(open('$OUT/bench_T.json')); print('T', d['ms_per_step'], d['lm_iterations_per_step'], d['value'])"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsT -o stats -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 1 > $OUT/statsT_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsT -name "*.db" | head -1) $OUT/kernel_stats_table_T.md > /dev/null; rm -rf $OUT/statsT )
head -8 $OUT/kernel_stats_table_T.md | cut -c1-120
