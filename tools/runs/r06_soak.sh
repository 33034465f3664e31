#!/bin/bash
# round 6: determinism soak — repeated solves of one context must be bit-identical (two-stream S assembly, level look-ahead on two streams)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout 900 python tools/soak_determinism.py C 60 6 2>&1 | tail -2
timeout 900 python tools/soak_determinism.py R 100 2>&1 | tail -2
timeout 900 python tools/soak_determinism.py L 100 2>&1 | tail -2
timeout 1200 python tools/soak_determinism.py T 6 8 2>&1 | tail -2
