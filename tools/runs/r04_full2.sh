#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04full2
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $OUT/pytest.txt
timeout 300 python tools/adapter_timing.py L 2>&1 | tee $OUT/adapter_timing.txt
timeout 600 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("L ms/solve", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "iter frac", d["roofline"]["iteration"]["frac"])
print(d.get("host_inclusive"))
PY
timeout 600 python bench.py --config M 2> $OUT/bench_M.err | tail -1 > $OUT/bench_M.json
python - $OUT/bench_M.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("M BA ms", d["value"], "replay wall", d["replay_wall_ms"], {k: {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in d["calls"].items()})
PY
