#!/bin/bash
# round 6, call 7: level look-ahead on the dissected-collection schedule (config T): A/B test, bench T with depth 0 / 1 / 2 / 3
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c7
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "look_ahead or ring_of_small or dissect or collection" 2>&1 | tail -8
for dep in 0 2 1 3; do
  XRSFM_BA_LA_DEPTH=$dep timeout 900 python bench.py --config T --no-cpu --no-extras --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' > $OUT/bench_T_$dep.json
  echo "T depth=$dep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_T_$dep.json)"
done
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; tail -12 $OUT/slow_calls.txt
