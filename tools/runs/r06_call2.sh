#!/bin/bash
# round 6, call 2: instruction mix / pipe occupancy of config R's kernels (is k_schur_pairs<.,3> instruction-bound there?)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
bash tools/pmc_mix.sh R r06_R > /dev/null 2>&1
grep -E "^\| kernel|k_schur_pairs|k_linearize|k_backsub|k_chol_segsum" gpurun_out/pmc_r06_R/mix.md
