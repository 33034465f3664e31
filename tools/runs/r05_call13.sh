#!/bin/bash
# round 5, call 13: back-to-back one-shot calls at K, with and without a pause between them
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
echo "--- no pause"; XRSFM_BA_PACK_TIMING=1 python tools/oneshot_repeat.py K 4 2>&1 | grep -E "^call|upload \+ observation"
echo "--- 100 ms pause"; XRSFM_BA_PACK_TIMING=1 python tools/oneshot_repeat.py K 4 100 2>&1 | grep -E "^call|upload \+ observation"
# RESULT: the slow call is the SECOND one-shot call of a process only — its H2D upload of the pageable observation arrays takes ~20 ms instead of
# 2.6 (inside hipMemcpy), with or without a pause, with or without deferring the release of the previous context's host memory (tried, removed):
# no pause 19.6 / 23.9 / 2.7 / 2.7 ms, 100 ms pause 22.9 / 6.8 / 2.6 / 3.6, and with the deferral 18.9 / 5.7 / 2.6 / 2.6 and 21.4 / 25.6 / 2.6 / 3.3.
# A one-off of the runtime, not of K / X: steady-state create at K is 9.5-10 ms.
