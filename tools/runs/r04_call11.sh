#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04k
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_adapter.py tests/test_mapper_replay.py tests/test_gpu_pack.py tests/test_replay.py -x -q 2>&1 | tail -6 | tee $OUT/pytest.txt
timeout 300 python tools/adapter_timing.py L 2>&1 | tee $OUT/adapter_timing.txt
XRSFM_BA_PACK_TIMING=1 timeout 300 python tools/pack_phases.py L 2>&1 | tail -22 > $OUT/pack_phases.txt; tail -22 $OUT/pack_phases.txt | grep -v "^\[plan\]"
for i in 1 2; do
timeout 600 python bench.py --config M 2> $OUT/bench_M.err | tail -1 > $OUT/bench_M_$i.json
python - $OUT/bench_M_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("M BA ms", d["value"], "replay wall", d["replay_wall_ms"], {k: {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in d["calls"].items()})
PY
done
