#!/bin/bash
# round 5, call 14: k_backsub as a persistent kernel (a wave walks items, the next item's slot record requested one item ahead) at L / K / R
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c14
mkdir -p $OUT
cd $ROOT
for cfg in L K R; do
for pw in 0 512 1024 2048; do
  XRSFM_BA_PERSIST=$pw XRSFM_BENCH_SELFPROF=0 timeout 300 python bench.py --config $cfg --no-cpu --no-extras --steps 6 --warmup 2 2> $OUT/bench_${cfg}_$pw.err | tail -1 > $OUT/bench_${cfg}_$pw.json
  python - <<PY
import json
d = json.loads(open("$OUT/bench_${cfg}_$pw.json").read())
k = d["kernels"]["k_backsub"]
print("$cfg persist=$pw ms_per_step", round(d["ms_per_step"], 3), "k_backsub us per launch", round(1e3 * k["ms"] / k["launches"], 2), "rmse", d.get("final_rmse_px"))
PY
done; done
# RESULT (not adopted): the persistent form is slower — k_backsub 76 us (one wave per item, in this refactored build; 68 in the shipped one) against
# 85 us with 1024-2048 persistent workgroups and 101 with 512 at L; K 143 -> 162-193, R 83 -> 91-111.  The dispatcher's wave-per-item
# balancing is worth more than the slot record requested an item ahead; the refactoring the experiment needed (item body as a function, forced
# 4 waves per SIMD) alone cost the shipped kernel 8 us, so it was reverted with it.
