#!/bin/bash
# round 6, call 14: the 20-30 ms the GPU takes to complete two trivial kernels right after a KGBA + whole-map filter — runtime
# settings that touch queue / scratch management (zero-code experiments)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c14
mkdir -p $OUT
cd $ROOT
run() { echo "== $*"; env "$@" XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py 2>&1 | grep -E "slow call" | sed 's/.*launch call/launch call/' | tail -4; }
run HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
run GPU_MAX_HW_QUEUES=1
run XRSFM_BA_DEVICE_PACK=0 XRSFM_BA_DEVICE_KEYS=0
run HIP_FORCE_DEV_KERNARG=1
