#!/bin/bash
# round 5, call 4: the MFMA-panel form of the 16x16 pivot-block factorisation (PV 4) in the microbench; PCG vector kernels after the
# one-pass reductions (config V); the config-4-sized multi-rank test
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c4
mkdir -p $OUT
cd $ROOT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1; cat $OUT/potrf.txt
timeout 900 python -m pytest tests/test_multirank_gpu.py tests/test_gpu_parity.py -m gpu -q -x -k "config4_sized or pcg_gauge" 2>&1 | tail -5 > $OUT/pytest.txt; cat $OUT/pytest.txt
XRSFM_BENCH_SELFPROF=0 timeout 600 python bench.py --config V --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_V.err | tail -1 > $OUT/bench_V.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_V.json").read())
print("V ms_per_step", round(d["ms_per_step"], 2), "lm_it", d.get("lm_iterations_per_step"), "pcg_it", d.get("pcg_iterations_per_step"), {k: (round(v["ms"], 2), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
