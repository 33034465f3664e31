#!/bin/bash
# round 4, call 13: nested dissection + stored-operand blocks at config T; potrf variants
OUT=gpurun_out/r04_c13; mkdir -p $OUT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "nested_dissection or stored_operands or clustered_collection or test_gpu_pack" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python tools/t_sweep.py default XRSFM_BA_ND=0 XRSFM_BA_PAIR_V=0 XRSFM_BA_ND_CHUNK=4 XRSFM_BA_ND_CHUNK=10 > $OUT/t_sweep.txt 2>&1
cat $OUT/t_sweep.txt | grep -v amdgpu.ids
cat $OUT/potrf.txt | grep -v amdgpu.ids | head -12
