#!/bin/bash
# round 5: a longer fuzz with fresh seeds on the final library, all four problem classes
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_fuzz4
mkdir -p $OUT
cd $ROOT
( timeout 280 python tools/fuzz_extended.py 3500 120 big > $OUT/big.txt 2>&1 ) &
( timeout 280 python tools/fuzz_extended.py 4400 120 pcg > $OUT/pcg.txt 2>&1 ) &
( timeout 280 python tools/fuzz_extended.py 5900 400 tiny > $OUT/tiny.txt 2>&1 ) &
( timeout 280 python tools/fuzz_extended.py 6600 240 > $OUT/std.txt 2>&1 ) &
wait
for f in big pcg tiny std; do echo "== $f"; grep MISMATCH $OUT/$f.txt | cut -c1-160; tail -1 $OUT/$f.txt; done
# RESULT: big 112 problems / 0 mismatches, tiny 400 / 0, pcg 119 / 1, standard 237 / 3 — all four of the known ill-conditioned kind (72 cameras +
# one 66-observation track with random image points; costs 4e5-2e12, LM decisions equal).
