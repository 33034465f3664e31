#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/pack_crossover.py 2>&1 | tee $OUT/crossover.txt
