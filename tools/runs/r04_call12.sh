#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/r04l
timeout 300 python tools/adapter_timing.py L 2>&1 | tee gpurun_out/r04l/adapter_timing.txt
