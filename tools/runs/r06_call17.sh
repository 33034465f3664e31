#!/bin/bash
# round 6, call 17: new tests — the HIP result within the oracle's self-disagreement on the fuzz mismatch seeds, a dissected collection on
# 2 / 4 / 8 ranks sharing the GPU, bal9 mode with the per-context run tables (+ its bench line)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c17
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_fuzz_conditioning.py tests/test_multirank_gpu.py tests/test_gpu_bal9.py -m gpu -q -x -k "oracles_self or dissected_collection or bal9" 2>&1 | tail -6
timeout 600 python bench.py --config Lb9 --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
