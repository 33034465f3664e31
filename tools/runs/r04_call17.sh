#!/bin/bash
# round 4, call 17: chunked backward (4 tiles); counters of the factorisation kernels at config T
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_c17; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "nested_dissection or stored_operands" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 600 python tools/t_sweep.py default > $OUT/t_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/t_sweep.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/p$i.log 2>&1
  python $ROOT/tools/pmc_generic.py $(find $OUT/p$i -name "*.db" | head -1) $OUT/mix.md > /dev/null 2>&1
  rm -rf $OUT/p$i
done
grep "k_ll_update\|k_lv_factor\|k_lv_bwd\|k_chol_segsum_v\|kernel \|---" $OUT/mix.md | cut -c1-220
