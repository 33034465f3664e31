#!/bin/bash
# round 4: the complete GPU suite on the final library + the selection of the stored-operand path per configuration
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04full4
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for cfg in U R; do
  XRSFM_BENCH_SELFPROF=0 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['lm_iterations_per_step'], {k: v for k, v in d['kernels'].items() if k in ('k_schur_pairs', 'k_block_segsum')})"
done
