#!/bin/bash
# round 4, call 23: stored-operand blocks at config U / D (all per-pair, 1 M entries)?
for cfg in U D; do
for v in 0 1; do
  echo "config $cfg XRSFM_BA_PAIR_V=$v"
  XRSFM_BA_PAIR_V=$v XRSFM_BENCH_SELFPROF=0 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['lm_iterations_per_step'], {k: v for k, v in d['kernels'].items() if k in ('k_schur_pairs', 'k_block_segsum')})"
done
done
