#!/bin/bash
# round 6: fresh-seed fuzz of the final library (one Gram launch, multi-pass staging, run tables, look-ahead), all four classes
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_fuzz1
mkdir -p $OUT
cd $ROOT
timeout 1500 python tools/fuzz_extended.py 8000 160 big > $OUT/big.txt 2>&1; tail -3 $OUT/big.txt
timeout 900 python tools/fuzz_extended.py 8200 120 pcg > $OUT/pcg.txt 2>&1; tail -3 $OUT/pcg.txt
timeout 900 python tools/fuzz_extended.py 8400 500 tiny > $OUT/tiny.txt 2>&1; tail -3 $OUT/tiny.txt
timeout 1500 python tools/fuzz_extended.py 9000 300 > $OUT/std.txt 2>&1; tail -5 $OUT/std.txt
