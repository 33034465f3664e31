#!/bin/bash
# round 5, call 21: gram_tile4 with the tile's schedule requested at kernel start and parked in LDS (call 19: looked up in global
# memory where the products start = a memory round trip in the middle of every tile)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c21
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -s -k "gram_blocks_from_4x4" 2>&1 | grep -E "gram4 vs|passed|failed|Error|assert" | head -30
for g in 1 0; do XRSFM_BA_GRAM4=$g timeout 120 python tools/timeline.py L 2>&1 | sed -n "/k_schur_pairs/,/k_lv_factor/p" > $OUT/tl_gram4_$g.txt; done; paste $OUT/tl_gram4_1.txt $OUT/tl_gram4_0.txt | head -10
cd /tmp && export TMPDIR=/tmp
for cfg in L R Lb9; do for g4 in 1 0; do
  export XRSFM_BA_GRAM4=$g4
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 3 --warmup 1 > $OUT/bench_${cfg}_$g4.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_${cfg}_$g4.md > /dev/null; rm -rf $OUT/st
  echo "== $cfg gram4=$g4"; grep -E "k_schur_pairs|k9_pairs_gram" $OUT/table_${cfg}_$g4.md
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${cfg}_$g4.log | tail -1
done; done
# RESULT: L 101.2 -> 96.3 us on this box; R <.,3> 91.6 -> 101.4 (worse), Lb9 202.7 -> 205.5: the 4x4 form is kept for tiles of <= 4 cameras only.
