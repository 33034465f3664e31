#!/bin/bash
# round 4, call 15: nested dissection + stored-operand blocks (wave-per-entry sum) at config T
OUT=gpurun_out/r04_c15; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "nested_dissection or stored_operands or clustered_collection or test_gpu_pack or one_launch_backward" > $OUT/tests.txt 2>&1
tail -5 $OUT/tests.txt
timeout 900 python tools/t_sweep.py default XRSFM_BA_ND=0 > $OUT/t_sweep.txt 2>&1
cat $OUT/t_sweep.txt | grep -v amdgpu.ids
timeout 900 python bench.py --config T --no-cpu --steps 2 --warmup 1 > $OUT/bench_T.txt 2>&1
tail -1 $OUT/bench_T.txt | cut -c1-1500
