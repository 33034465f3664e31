#!/bin/bash
# Developer aid (run through gpurun from the repo root): rocprofv3 kernel-trace summary of one bench configuration.
#   gpurun -- 'bash tools/quick_prof.sh L tag [extra bench args]'
set -u
CFG=${1:-L}; TAG=${2:-q}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --config $CFG --no-cpu --no-extras --steps 2 "$@" > $OUT/stats_bench.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats_table.md
grep "^{\"metric\"" $OUT/stats_bench.log | tail -1 > $OUT/stats_bench_line.json
rm -rf $OUT/stats
