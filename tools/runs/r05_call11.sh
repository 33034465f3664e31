#!/bin/bash
# round 5, call 11: mapper replay after the adapter's table / sort rule (LBA p50 1.04 ms in the profile run against 0.84 in round 4)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c11
mkdir -p $OUT
cd $ROOT
python bench.py --config M 2> $OUT/bench_M.err | tail -1 > $OUT/bench_M.json
python - <<PY
import json
d = json.loads(open("$OUT/bench_M.json").read())
print({k: {kk: round(vv, 2) for kk, vv in v.items() if kk in ("count", "total_ms", "p50", "p90")} for k, v in d["calls"].items()}, round(d["value"], 1), round(d["replay_wall_ms"]))
PY
timeout 600 python -m pytest tests/test_adapter.py tests/test_mapper_replay.py -m gpu -q -x 2>&1 | tail -3
