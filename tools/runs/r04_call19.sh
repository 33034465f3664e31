#!/bin/bash
# round 4, call 19: the tiles outside the first level composed by a launch of their own (config T, 20 000 sequential cameras)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_c19; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "twenty_thousand or config5 or nested_dissection or clustered_collection or config3 or packed" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 600 python tools/t_sweep.py default > $OUT/t_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/t_sweep.txt
timeout 300 python bench.py --config X --no-cpu --no-extras --steps 3 --warmup 1 2>/dev/null | tail -1 | cut -c1-330
