#!/bin/bash
# round 5, call 19: Gram blocks from v_mfma_f64_4x4x4_4b_f64 (gram_tile4) against the 16x16 tiles (XRSFM_BA_GRAM4=0): A/B test, then
# per-kernel times at L, R and Lb9 in both forms
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c19
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -s -k "gram_blocks_from_4x4" 2>&1 | grep -E "gram4 vs|passed|failed|Error|assert" | head -30
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bal9.py -m gpu -q -x -k "golden or headline or config4_parity or bal9" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for cfg in L R Lb9; do for g4 in 1 0; do
  export XRSFM_BA_GRAM4=$g4
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 3 --warmup 1 > $OUT/bench_${cfg}_$g4.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_${cfg}_$g4.md > /dev/null; rm -rf $OUT/st
  echo "== $cfg gram4=$g4"; grep -E "k_schur_pairs|k9_pairs_gram" $OUT/table_${cfg}_$g4.md
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${cfg}_$g4.log | tail -1
done; done
# RESULT: bit-identical; L 99.6 -> 96.2 us, R and Lb9 unchanged (the schedule was looked up in global memory where the products start).
