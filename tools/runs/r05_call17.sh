#!/bin/bash
# round 5, call 17: bal9 kernels at 3 waves per SIMD (k9_linearize with the long-track sums out of the common path; k9_pairs_gram
# register-allocated for 3 waves) against the round-4 occupancy (library k9b = 2 / 2) and k9_linearize at 4 (k9a: 2 spilled values);
# and tools/bench_pipes: do FP64 matrix and vector instructions of different waves of a SIMD overlap?
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c17
mkdir -p $OUT
cd $ROOT
timeout 60 tools/bench_pipes > $OUT/pipes.txt 2>&1; cat $OUT/pipes.txt
timeout 600 python -m pytest tests/test_gpu_bal9.py -m gpu -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in "" _k9a _k9b; do
  export XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba$v.so
  XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace --stats -d $OUT/st$v -o st -- python $ROOT/bench.py --config Lb9 --no-cpu --no-extras --steps 3 --warmup 1 > $OUT/bench$v.log 2>&1
  python $ROOT/tools/rocprof_summary.py $(find $OUT/st$v -name "*.db" | head -1) $OUT/table$v.md > /dev/null; rm -rf $OUT/st$v
  echo "== lib '$v'"; grep -E "k9_" $OUT/table$v.md
  grep -o '"ms_per_step": [0-9.]*' $OUT/bench$v.log | tail -1
done
# RESULT: pipes — 2 vector + 2 matrix waves per SIMD take exactly t(vector alone) + t(matrix alone) (18.8 = 1.56 + 17.3 ms): FP64 matrix
# instructions run on the vector data path; 16x16x4 = 64 cycles, 4x4x4 (4 blocks) = 16.5, a vector FMA 4.5.  bal9: Lb9 18.6 (2/2 waves) -> 17.2 ms
# (3/3: k9_pairs_gram 235 -> 199 us, k9_linearize 172 -> 148); k9_linearize at 4 waves (2 spills) 142 us: not taken.
