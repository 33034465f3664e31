#!/bin/bash
# round 5, call 29: with stored operands the block scatter buffer holds the Gram cells only (compact numbering): A/B tests, config T at size
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c29
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py -m gpu -q -x -k "stored_operands" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config5 or unordered or random_visibility" 2>&1 | tail -2
XRSFM_BENCH_SELFPROF=0 python bench.py --config T --steps 1 --warmup 1 --no-cpu --no-extras 2> $OUT/T.err | tail -1 > $OUT/T.json
python -c "
import json; d=json.loads(open('$OUT/T.json').read()); print('T', d['ms_per_step'], d['lm_iterations_per_step'], d['final_rmse_px'])"
XRSFM_BENCH_SELFPROF=0 python bench.py --config D --steps 1 --warmup 1 --no-cpu --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('D', d['ms_per_step'], d['lm_iterations_per_step'], d['final_rmse_px'])"
# RESULT: A/B tests green (collection: 2 Gram tiles; ragged map: 2526 Gram tiles + 13 per-pair items with PAIR_V forced); T 960 ms, 35 LM iterations,
# final RMSE 1.0285542455561951 px = the value of profiles/r05_bench_lines.md to the last digit; D 223.7 ms.
