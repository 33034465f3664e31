#!/bin/bash
# Developer aid: rocprofv3 kernel-trace summary of an arbitrary command.   gpurun -- 'bash tools/quick_prof_cmd.sh tag python tools/lba_phases.py'
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- "$@" > $OUT/cmd.log 2>&1 )
python $ROOT/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats_table.md
rm -rf $OUT/stats
