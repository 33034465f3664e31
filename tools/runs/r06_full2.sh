#!/bin/bash
# round 6: full GPU suite + smoke + the driver command at the current HEAD (pinned upload / download staging included)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_full2
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $OUT/pytest.txt; cat $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
( time python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/bench.json ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read())
r = d["roofline"]; c = d["cpu_baseline"]
print("ms_per_step", d["ms_per_step"], "value", d["value"], "frac", r["frac"], "iter", r["iteration"]["frac"], "traffic", r.get("traffic"), "rocprof_us", r.get("rocprofv3_avg_launch_us"), "cpu x", c["gpu_vs_cpu"], "LP cam", c["parity_workload"]["max_cam_param_diff"], "host", d["host_inclusive"]["total_ms"], d["host_inclusive"]["first_call_ms"], "mfma", d["mfma_utilisation"]["frac"])
PY
timeout 600 python tools/lba_timing.py 2>&1 | tail -4
timeout 900 python bench.py --config M 2>/dev/null | grep '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('M BA', round(d['value'],1), 'wall', round(d['replay_wall_ms'],1), {k:(round(v['total_ms'],1), round(v['p50'],3)) for k,v in d['calls'].items()})"
