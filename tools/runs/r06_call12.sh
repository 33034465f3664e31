#!/bin/bash
# round 6, call 12: is the device still busy when a slow pose refinement starts? (hipDeviceSynchronize in front of its upload, trace mode)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c12
mkdir -p $OUT
cd $ROOT
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; grep -E "mapper_main|slow call" $OUT/slow_calls.txt | tail -12
