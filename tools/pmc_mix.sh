#!/bin/bash
# Instruction mix / pipe occupancy of the kernels of one bench configuration (rocprofv3 PMC, counters only + kernel trace).
#   gpurun -- 'bash tools/pmc_mix.sh L tag'
set -u
CFG=${1:-L}; TAG=${2:-mix}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; rm -f $OUT/mix.md
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python $ROOT/bench.py --config $CFG --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
  python $ROOT/tools/pmc_generic.py $(find $OUT/p$i -name "*.db" | head -1) $OUT/mix.md > /dev/null
  rm -rf $OUT/p$i
done
cat $OUT/mix.md
