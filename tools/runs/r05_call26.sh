#!/bin/bash
# round 5, call 26: staged operand rows padded by ONE column (default now) against two (library pad2): GPU tests that touch the S assembly,
# then R / K / Lb9 / L per-kernel times in both forms
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c26
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_bal9.py tests/test_gpu_pack.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or headline or config4_parity or ragged or gram or schur" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for cfg in R K Lb9 L; do for v in "" _pad2; do
  export XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba$v.so
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 4 --warmup 1 > $OUT/bench_$cfg$v.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_$cfg$v.md > /dev/null; rm -rf $OUT/st
  echo "== $cfg lib '$v'"; grep -E "k_schur_pairs|k9_pairs_gram" $OUT/table_$cfg$v.md | head -3; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$cfg$v.log | tail -1
done; done
# RESULT: tests green; pad 1 vs 2: L 89.9 / 91.1, K 170.1 / 174.4, R <.,3> 89.4 / 91.5 and <.,4> 47.7 / 57.6, Lb9 197.8 / 198.0 us.
