#!/bin/bash
# round 4, call 22: config R / LP lines and R's kernel table after the stored-operand threshold fix and the forked Gram buckets
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_r04; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "ragged or long_items or stored_operands or closures or fuzz" > $OUT/tests_c22.txt 2>&1
tail -3 $OUT/tests_c22.txt
for cfg in R LP; do
  XRSFM_BENCH_SELFPROF=0 python bench.py --config $cfg --steps 5 --warmup 2 2> $OUT/bench_$cfg.err | tail -1 > $OUT/bench_$cfg.json
done
XRSFM_BA_GRAM_FORK=0 XRSFM_BENCH_SELFPROF=0 python bench.py --config R --steps 5 --warmup 2 --no-cpu --no-extras 2>/dev/null | tail -1 | cut -c1-300
python - $OUT/bench_R.json $OUT/bench_LP.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); print(f.split('/')[-1], d["ms_per_step"], d["lm_iterations_per_step"], d["value"])
PY
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsR -o stats -- python $ROOT/bench.py --config R --no-cpu --no-extras --steps 2 > $OUT/statsR_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsR -name "*.db" | head -1) $OUT/kernel_stats_table_R.md > /dev/null; rm -rf $OUT/statsR )
head -12 $OUT/kernel_stats_table_R.md | cut -c1-120
