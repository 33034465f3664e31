#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_bal9.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -6 | tee $OUT/pytest.txt
for cfg in R K Lb9; do
  timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 10 --warmup 2 2> $OUT/bench_$cfg.err | tail -1 > $OUT/bench_$cfg.json
  python - "$OUT/bench_$cfg.json" "$cfg" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    k = d.get("kernels", {})
    print(f"{sys.argv[2]:3s} ms/solve {d['ms_per_step']:.3f}  it {d['lm_iterations_per_step']}  rmse {d['final_rmse_px']:.9f}  " + " ".join(f"{n}={v['ms'] * 1e3 / max(v['launches'], 1):.1f}us" for n, v in k.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
timeout 600 python bench.py --no-cpu > $OUT/bench_L_selfprof.json 2> $OUT/bench_L_selfprof.err
python - $OUT/bench_L_selfprof.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("L ms/solve", d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "|", r.get("traffic_source"))
print("rocprofv3_avg_launch_us", r.get("rocprofv3_avg_launch_us"), "hip events avg", r["avg_launch_us"])
print(json.dumps(r.get("rocprofv3", {}))[:1500])
print(d.get("host_inclusive"))
PY
XRSFM_BA_PACK_TIMING=1 timeout 300 python tools/pack_phases.py L 2>&1 | tail -45 > $OUT/pack_phases.txt; tail -45 $OUT/pack_phases.txt
timeout 300 python tools/timeline.py R 2>&1 | head -12 > $OUT/timeline_R.txt; cat $OUT/timeline_R.txt
