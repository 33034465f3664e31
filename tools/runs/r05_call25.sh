#!/bin/bash
# round 5, call 25: padding of a staged operand row 2 -> 1 column (odd row stride: the 4-row groups of one matrix instruction fall on
# different LDS banks); k_schur_pairs time and bank-conflict cycles at L
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in "" _pad1; do
  export XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba$v.so
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 5 --warmup 2 > $OUT/bench$v.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table$v.md > /dev/null; rm -rf $OUT/st
  echo "== lib '$v'"; grep -E "k_schur_pairs" $OUT/table$v.md; grep -o '"ms_per_step": [0-9.]*' $OUT/bench$v.log | tail -1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d $OUT/p -o p -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/p$v.log 2>&1
  python $ROOT/tools/pmc_generic.py $(find $OUT/p -name "*.db" | head -1) $OUT/mix$v.md > /dev/null; rm -rf $OUT/p
  grep -E "kernel|k_schur_pairs" $OUT/mix$v.md | head -3
done
# RESULT: bank-conflict cycles 2.18e5 -> 1.49e5, 92.6 -> 90.4 us on this box: adopted (kGramPad = 1).
