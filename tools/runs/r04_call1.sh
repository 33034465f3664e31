#!/bin/bash
# GPU call 1 of round 4: pivot-tile microbench, A/B of the potrf builds, kernel table of L, then the whole GPU suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1
cat $OUT/potrf.txt
timeout 600 bash tools/runs/r04_ab.sh r04a "L S" "default potrf00 potrf01 potrf11 potrf20" 20 2>&1 | tee $OUT/ab.txt
timeout 300 bash tools/quick_prof.sh L r04a_L > /dev/null 2>&1
cat gpurun_out/prof_r04a_L/kernel_stats_table.md | head -30
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest.txt
