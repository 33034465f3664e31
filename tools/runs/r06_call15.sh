#!/bin/bash
# round 6, call 15: with pinned staging for batched uploads / downloads: are the slow pose refinements of the mapper replay gone? (3 replays)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
for i in 1 2 3; do XRSFM_BA_TRACE_CALLS=1 timeout 900 python tools/mapper_slow_calls.py 2>&1 | grep -E "mapper_main.*class 3|slow call" | sed 's/.*launch call/launch call/' | tail -4; echo "--"; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py -m gpu -q -x -k "config5 or collection or look_ahead or pair" 2>&1 | tail -3
