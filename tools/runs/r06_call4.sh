#!/bin/bash
# round 6, call 4: watchdog test (stall in front of the first scalar hand-over), adapter observation cache (test + timing at L),
# mapper replay with the "filters+refine" class split three ways
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c4
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_lifetime_gpu.py tests/test_adapter.py -m gpu -q -x -k "watchdog or timeout_stops or observation_cache" 2>&1 | tail -15
timeout 600 python tools/adapter_timing.py L > $OUT/adapter_timing_L.txt 2>&1; cat $OUT/adapter_timing_L.txt
timeout 900 python bench.py --config M 2>/dev/null | grep '^{"metric"' > $OUT/bench_M.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c4/bench_M.json"))
print("BA ms", d["value"], "wall", d["replay_wall_ms"])
for k, v in d["calls"].items(): print(k, v)
PY
