#!/bin/bash
# round 5, call 23: per-lane temporaries that only lanes with an observation read are DEFINED without an instruction (no v_mov per
# register pair: 46 in k_schur_pairs, 28 in k_backsub); the poison build still fills them with NaN.  Hardening + parity subset, then per-kernel times.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c23
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or headline or config4_parity or ragged or pcg or schur_product" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for cfg in L R; do
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config $cfg --no-cpu --no-extras --steps 5 --warmup 2 > $OUT/bench_$cfg.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_$cfg.md > /dev/null; rm -rf $OUT/st
  echo "== $cfg"; grep -E "k_schur_pairs|k_backsub|k_linearize" $OUT/table_$cfg.md
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$cfg.log | tail -1
done
# RESULT: hardening (poison / strict) green; k_schur_pairs 96.2 -> 92.1 us at L, k_backsub unchanged (66.5: latency-bound).
