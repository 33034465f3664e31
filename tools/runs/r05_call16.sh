#!/bin/bash
# round 5, call 16: k_lin_tail with its job slices requested first — per-kernel averages at L, tests that cover the tail
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c16
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -m gpu -q -x -k "golden or headline or config4_parity or poison or shape or fused" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 3 --warmup 1 > $OUT/bench.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table.md > /dev/null; rm -rf $OUT/st
grep -E "k_lin_tail|k_linearize|k_schur_pairs|k_backsub" $OUT/table.md
grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log | tail -1
# RESULT (not adopted): 23.7 -> 23.1 us — the tail is not waiting for those loads.
