#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_pack.py -x -q 2>&1 | tail -12 | tee $OUT/pytest_pack.txt
XRSFM_BA_DEVICE_PACK=1 timeout 1200 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_forced.txt
XRSFM_BA_PACK_TIMING=1 timeout 300 python tools/pack_phases.py L 2>&1 | tail -22 > $OUT/pack_phases.txt; cat $OUT/pack_phases.txt
timeout 300 python tools/create_timing.py L R K 2>&1 | tail -4 | tee $OUT/create_timing.txt
