#!/bin/bash
# round 6, call 18: the level look-ahead's two cross-stream dependencies through stream memory operations (hipStreamWriteValue64 /
# hipStreamWaitValue64 on two signal words) instead of events: config T, and the A/B test under that setting
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c18
mkdir -p $OUT
cd $ROOT
XRSFM_BA_LA_SYNC=value timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "look_ahead" 2>&1 | tail -3
for sync in value event; do
  XRSFM_BA_LA_SYNC=$sync timeout 900 python bench.py --config T --no-cpu --no-extras --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' > $OUT/bench_T_$sync.json
  echo "T sync=$sync $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_T_$sync.json)"
done
cd /tmp && export TMPDIR=/tmp
XRSFM_BA_LA_SYNC=value XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/trace_window.py $DB "k_lv_factor<false>" 3000 14 > $OUT/window_T.txt; cat $OUT/window_T.txt
rm -rf $OUT/tr
