#!/bin/bash
# round 5, call 12: why is the SECOND one-shot call of bench.py's host_inclusive slower than the first at K (create 23.7 ms against ~10)?
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c12
mkdir -p $OUT
cd $ROOT
XRSFM_BA_PACK_TIMING=1 XRSFM_BENCH_SELFPROF=0 python bench.py --config K --no-cpu --steps 1 --warmup 0 2> $OUT/K.err | tail -1 > $OUT/K.json
grep -E "^\[create\]|^\[devpack\]|^\[chol setup\]|^== " $OUT/K.err | tail -60
python - <<PY
import json
d = json.loads(open("$OUT/K.json").read()); print(d["host_inclusive"])
PY
