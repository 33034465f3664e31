#!/bin/bash
# round 5, call 24: gram4 stores — destinations of a batch requested together, unsigned address arithmetic; per-kernel time at L and the
# instruction mix of the run (three counter passes)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c24
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -k "gram_blocks or poison" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 5 --warmup 2 > $OUT/bench_L.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table_L.md > /dev/null; rm -rf $OUT/st
grep -E "k_schur_pairs|k_backsub|k_linearize" $OUT/table_L.md
grep -o '"ms_per_step": [0-9.]*' $OUT/bench_L.log | tail -1
bash $ROOT/tools/pmc_mix.sh L r05c24 > $OUT/mix.md 2>&1; grep -E "kernel|k_schur_pairs|k_linearize|k_backsub" $OUT/mix.md | head -20
# RESULT: 92.1 -> 90.7 us; 774 vector instructions per tile incl. 59 matrix (round 4: 789 + 35.5), bank-conflict cycles 2.18e5 of 2.66e5 LDS-active.
