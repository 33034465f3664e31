#!/bin/bash
# round 5, call 22: gram_tile4 for tiles of at most 4 cameras only (schedule entry per lane, parked in LDS); bal9 kernels at 3 waves per SIMD.
# hardening + bal9 + parity subsets, then L / Lb9 / S / K bench lines
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c22
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_bal9.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or headline or config4_parity or ragged or gram" 2>&1 | tail -3
for cfg in L Lb9 S K; do
  python bench.py --config $cfg --no-cpu --no-extras --steps 10 --warmup 3 > $OUT/bench_$cfg.log 2>&1
  echo "== $cfg"; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$cfg.log | tail -1; grep -o '"frac": [0-9.]*' $OUT/bench_$cfg.log | head -2
done
# RESULT: 75 + 24 tests green; L 6.23 ms (frac 0.459), Lb9 17.2, S 2.07, K 15.5.
