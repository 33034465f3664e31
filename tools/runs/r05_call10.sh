#!/bin/bash
# round 5, call 10: the single-workgroup PCG vector update — tests, config V; where the first create of a process spends its time
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c10
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_multirank_gpu.py -m gpu -q -x -k "pcg or long_items or variant or (ranks_equal and pcg)" 2>&1 | tail -4
XRSFM_BENCH_SELFPROF=0 timeout 600 python bench.py --config V --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_V.err | tail -1 > $OUT/bench_V.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_V.json").read())
print("V ms_per_step", round(d["ms_per_step"], 2), "lm_it", d.get("lm_iterations_per_step"), "pcg_it", d.get("pcg_iterations_per_step"), {k: (round(v["ms"], 2), v["launches"]) for k, v in d.get("kernels", {}).items()})
PY
XRSFM_BA_PACK_TIMING=1 python tools/adapter_timing.py L 2>&1 | tail -70 > $OUT/adapter_timing_L.txt; cat $OUT/adapter_timing_L.txt
# RESULT: (1) the single-workgroup PCG vector update (k_pcg_vec1: x, r, z, the nine dot products and p for all cameras in one workgroup, instead of
# k_pcg_xr + k_pcg_p) is 4x SLOWER — 121 us per PCG iteration against 31 at config V (3000 cameras: three dependent passes over 1.2 MB on ONE
# CU, eleven workgroup-wide reductions); V 269 -> 433 ms.  Removed.  (2) first GBA of a process after BASolver()'s warm-up: create 47.5 ms
# (113 in round 4), of which ~40 ms are hipMalloc (2 GB of device memory mapped for the first time) and 18 ms the first use of the rocPRIM
# sorts; the first run 22.9 ms against 7.2 (first launch of every kernel).
