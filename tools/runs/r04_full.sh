#!/bin/bash
# whole GPU suite + the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04full
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest.txt
timeout 600 python bench.py 2> $OUT/bench.err | tail -1 > $OUT/bench.json
python - $OUT/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("L ms/solve", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "iter frac", d["roofline"]["iteration"]["frac"])
print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "sample"}) for k, v in (d.get("cpu_baseline") or {}).items() if k in ("value", "cores", "gpu_vs_cpu", "max_cam_param_diff", "parity_workload")})
print(d.get("host_inclusive"))
PY
