#!/bin/bash
# round 4, call 18: XCD-aware chunk order of the partial-product launches at config T
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04_c18; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q -k "nested_dissection or stored_operands or clustered_collection" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
timeout 600 python tools/t_sweep.py default XRSFM_BA_CHUNK_ORDER=0 XRSFM_BA_ND_CHUNK=4 XRSFM_BA_ND_CHUNK=12 > $OUT/t_sweep.txt 2>&1
grep -v amdgpu.ids $OUT/t_sweep.txt
