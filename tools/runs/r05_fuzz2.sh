#!/bin/bash
# round 5: the fuzz seeds that mismatch (all of the "72 cameras + one 66-observation track with random image points" kind) on this round's library and on round 4's
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
echo "== round 5, exact"; python tools/fuzz_seeds.py 1 6084 6114 6119 6139 2>&1 | grep seed
echo "== round 4 library, exact"; XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba_r04.so python tools/fuzz_seeds.py 1 6084 6114 6119 6139 2>&1 | grep seed
echo "== round 5, pcg"; python tools/fuzz_seeds.py 0 4054 4099 4119 4149 4154 4159 2>&1 | grep seed
echo "== round 5, pcg, block-Jacobi alone"; XRSFM_BA_PCG_COARSE=0 python tools/fuzz_seeds.py 0 4054 4099 4119 4149 4154 4159 2>&1 | grep seed
echo "== round 4 library, pcg"; XRSFM_BA_LIB=$ROOT/xrsfm_amd/lib/libxrsfm_ba_r04.so python tools/fuzz_seeds.py 0 4054 4099 4119 4149 4154 4159 2>&1 | grep seed
