#!/bin/bash
# round 4, call 21: k_chol_segsum_v with one wave per block (config T)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_c21; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "stored_operands or nested_dissection" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 600 python tools/t_sweep.py default > $OUT/t_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/t_sweep.txt
