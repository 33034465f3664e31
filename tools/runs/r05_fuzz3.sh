#!/bin/bash
# round 5: extended fuzz with fresh seeds on the final library (Gram blocks from 4x4x4 instructions, dead temporaries, one padding column)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_fuzz3
mkdir -p $OUT
cd $ROOT
( timeout 230 python tools/fuzz_extended.py 3300 60 big > $OUT/big.txt 2>&1 ) &
( timeout 230 python tools/fuzz_extended.py 5600 220 tiny > $OUT/tiny.txt 2>&1 ) &
( timeout 230 python tools/fuzz_extended.py 6400 120 > $OUT/std.txt 2>&1 ) &
wait
for f in big tiny std; do echo "== $f"; grep -c MISMATCH $OUT/$f.txt; tail -2 $OUT/$f.txt; done
# RESULT: big 56 problems / 0 mismatches, tiny 220 / 0, standard 120 / 1 (seed 6449, the known ill-conditioned kind: 72 cameras + a 66-observation
# track with random image points, cost 3.2e7, relative cost difference 8.7e-9).
