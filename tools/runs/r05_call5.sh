#!/bin/bash
# round 5, call 5: cycle stamps inside k_lv_factor at config L (where do the 10.6 us of the pivot-tile factorisation go?), PV 1 and PV 4
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c5
mkdir -p $OUT
cd $ROOT
for tag in pv1 pv4; do
  XBA_TL_TAG=$tag timeout 300 python tools/timeline.py L > $OUT/timeline_$tag.txt 2>&1
  echo "== $tag"; grep -A17 "k_lv_factor" $OUT/timeline_$tag.txt
done
