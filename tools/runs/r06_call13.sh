#!/bin/bash
# round 6, call 13: k_refine_pose without a scratch segment — do the 20-28 ms pose refinements after a large KGBA go away?
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c13
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "refine" 2>&1 | tail -4
XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; grep -E "mapper_main|slow call" $OUT/slow_calls.txt | tail -12
timeout 900 python bench.py --config M 2>/dev/null | grep '^{"metric"' > $OUT/bench_M.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c13/bench_M.json"))
print("BA ms", d["value"], "wall", d["replay_wall_ms"])
for k, v in d["calls"].items(): print(k, {a: round(b, 3) for a, b in v.items()})
PY
