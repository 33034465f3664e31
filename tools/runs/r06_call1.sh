#!/bin/bash
# round 6, call 1: where config R's S assembly spends its time — kernel trace window of one pass (what overlaps what on the two
# streams), cycle stamps inside k_schur_pairs on R — and baseline bench lines of this box (L, R, S)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c1
mkdir -p $OUT
cd $ROOT
for cfg in L R S; do timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' > $OUT/bench_$cfg.json; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$cfg.json; done
timeout 300 python tools/timeline.py R > $OUT/tl_R.txt 2>&1; sed -n "/k_schur_pairs/,/k_lv_factor/p" $OUT/tl_R.txt
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config R --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/iteration_gaps.py $DB 20 > $OUT/gaps_R_20.txt; cat $OUT/gaps_R_20.txt
python $ROOT/tools/trace_window.py $DB "k_schur_pairs<true, true, 3>" 40 12 > $OUT/window_R.txt; cat $OUT/window_R.txt
rm -rf $OUT/tr
