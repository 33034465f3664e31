#!/bin/bash
# round 6, call 3: one Gram launch for heights 1..3 (multi-pass staging, one LDS class) — A/B bit-identity, robustness tests
# (watchdog, backward-substitution time-out, fuzz conditioning), bench lines R / L / K / T-less
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c3
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_gpu_hardening.py tests/test_lifetime_gpu.py tests/test_fuzz_conditioning.py -m gpu -q -x -k "one_gram_launch or watchdog or timeout_stops or within_the_oracles or gram_blocks_from_4x4" 2>&1 | tail -15
for cfg in R L K; do for m in 1 0; do
  XRSFM_BA_GRAM_MERGE=$m timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 5 --warmup 2 2>/dev/null | grep '^{"metric"' > $OUT/bench_${cfg}_$m.json
  echo "$cfg merge=$m $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_${cfg}_$m.json)"
done; done
cd /tmp && export TMPDIR=/tmp
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config R --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/bench_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python $ROOT/tools/iteration_gaps.py $DB 20 > $OUT/gaps_R_20.txt; head -8 $OUT/gaps_R_20.txt; tail -1 $OUT/gaps_R_20.txt
rm -rf $OUT/tr
