#!/bin/bash
# round 6, call 5: track filter as one arena / one upload / one download; slow calls of the mapper replay; watchdog test
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_c5
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_lifetime_gpu.py tests/test_gpu_parity.py tests/test_adapter.py tests/test_mapper_replay.py tests/test_replay.py -m gpu -q -x -k "watchdog or filter or replay or observation_cache" 2>&1 | tail -8
timeout 900 python tools/mapper_slow_calls.py > $OUT/slow_calls.txt 2>&1; cat $OUT/slow_calls.txt | head -60
timeout 900 python bench.py --config M 2>/dev/null | grep '^{"metric"' > $OUT/bench_M.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_c5/bench_M.json"))
print("BA ms", d["value"], "wall", d["replay_wall_ms"])
for k, v in d["calls"].items(): print(k, {a: round(b, 3) for a, b in v.items()})
PY
