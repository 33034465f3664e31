#!/bin/bash
# round 6, call 16: row-paired early chunks (k_ll_update_part2: two targets of one column share the operand L_kj): A/B test, bench T
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c16
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -m gpu -q -x -k "look_ahead or ring_of_small or dissect or collection or config5" 2>&1 | tail -5
for pr in 1 0; do
  XRSFM_BA_LA_PAIRS=$pr timeout 900 python bench.py --config T --no-cpu --no-extras --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' > $OUT/bench_T_$pr.json
  echo "T pairs=$pr $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_T_$pr.json)"
done
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/trace_window.py $DB "k_lv_factor<false>" 3000 24 > $OUT/window_T.txt; cat $OUT/window_T.txt
rm -rf $OUT/tr
