#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_pack.py -x -q 2>&1 | tail -12 | tee $OUT/pytest.txt
XRSFM_BA_PACK_TIMING=1 timeout 300 python tools/pack_phases.py L 2>&1 | tail -34 > $OUT/pack_phases.txt; cat $OUT/pack_phases.txt
timeout 300 python tools/create_timing.py 2>&1 | tail -12 | tee $OUT/create_timing.txt
