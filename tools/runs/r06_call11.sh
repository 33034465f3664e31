#!/bin/bash
# round 6, call 11: are the slow pose refinements of the mapper replay (20-27 ms between launch and the end of hipStreamSynchronize, the
# kernel itself <= 0.2 ms) the runtime's interrupt-driven wait?  The same replay with HSA_ENABLE_INTERRUPT=0 (polling waits)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c11
mkdir -p $OUT
cd $ROOT
for intr in 1 0; do
  echo "== HSA_ENABLE_INTERRUPT=$intr"
  HSA_ENABLE_INTERRUPT=$intr XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls_$intr.txt 2>&1; grep -E "mapper_main|slow call" $OUT/slow_calls_$intr.txt | tail -10
done
