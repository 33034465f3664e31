#!/bin/bash
# round 4, call 24: chunk kernel of the level schedules with one LDS image, four workgroups per CU
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_c24; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "nested_dissection or one_launch_backward or twenty_thousand or config3" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 600 python tools/t_sweep.py default > $OUT/t_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/t_sweep.txt
for cfg in L R X; do
  XRSFM_BENCH_SELFPROF=0 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['lm_iterations_per_step'], d['kernels'].get('k_update'))"
done
