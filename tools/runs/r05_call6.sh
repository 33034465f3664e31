#!/bin/bash
# round 5, call 6: PV 4 with the square roots off the dependent chain — microbench + cycle stamps (incl. wave 1 around the last row of the inverse)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c6
mkdir -p $OUT
cd $ROOT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1; cat $OUT/potrf.txt
for tag in pv1 pv4; do
  XBA_TL_TAG=$tag timeout 300 python tools/timeline.py L > $OUT/timeline_$tag.txt 2>&1
  echo "== $tag"; grep -A17 "k_lv_factor" $OUT/timeline_$tag.txt
done
