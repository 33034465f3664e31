#!/bin/bash
# round 5, call 27: k_linearize — the zeros of lanes without an observation set in the else branch (a full tile skips 21 v_mov)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c27
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -k "poison or shape or strict" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or headline or linearize or ragged" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for cfg in L R; do
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 5 --warmup 2 > $OUT/bench_$cfg.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_$cfg.md > /dev/null; rm -rf $OUT/st
  echo "== $cfg"; grep -E "k_schur_pairs|k_backsub|k_linearize|k_cost" $OUT/table_$cfg.md | head -5
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$cfg.log | tail -1
done
# RESULT (not adopted): k_linearize 77.3 us either way — it is not bound by its instruction count; the change was reverted.
