#!/bin/bash
# round 4: the complete GPU suite + the driver's default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04full3
mkdir -p $OUT
cd $ROOT
timeout 1700 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -40 | tee $OUT/pytest.txt
timeout 600 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("L ms/solve", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "iter frac", d["roofline"]["iteration"]["frac"])
print(d.get("host_inclusive"))
PY
