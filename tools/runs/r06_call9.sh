#!/bin/bash
# round 6, call 9: level look-ahead with the late contributions inside the factor kernel (early chunks only on the second stream):
# A/B test, bench T at depth 0 / 1 / 2, trace window at depth 1; GPU wake-up latency after host-only phases
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c9
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/idle_wake_probe.py 2>&1 | tee $OUT/idle_wake.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "look_ahead or ring_of_small or dissect or collection" 2>&1 | tail -5
for dep in 0 1 2; do
  XRSFM_BA_LA_DEPTH=$dep timeout 900 python bench.py --config T --no-cpu --no-extras --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' > $OUT/bench_T_$dep.json
  echo "T depth=$dep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_T_$dep.json)"
done
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/trace_window.py $DB "k_lv_factor<false>" 3000 24 > $OUT/window_T.txt; cat $OUT/window_T.txt
rm -rf $OUT/tr
