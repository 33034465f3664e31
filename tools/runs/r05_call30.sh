#!/bin/bash
# round 5, call 30: k_schur_pairs with the wave's priority raised (s_setprio 2) from the Gram stage on (prio1) / from the end of the loads on (prio2):
# does "finish the tile you started" shorten the kernel?
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c30
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in _prio1 _prio2 ""; do
  export XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba$v.so
  [ -z "$v" ] && unset XRSFM_BA_LIB
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 5 --warmup 2 > $OUT/bench$v.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table$v.md > /dev/null; rm -rf $OUT/st
  echo "== lib '$v'"; grep -E "k_schur_pairs" $OUT/table$v.md; grep -o '"ms_per_step": [0-9.]*' $OUT/bench$v.log | tail -1
done
# RESULT (not adopted): 89.2 / 88.5 us against 88.6 without: the order in which the SIMD picks its waves is not what holds the kernel.
