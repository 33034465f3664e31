#!/bin/bash
# GPU call 2 of round 4: pivot-tile microbench (PV3), where the ragged configuration R spends its time, the tests that changed.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
timeout 120 tools/bench_potrf > $OUT/potrf.txt 2>&1; cat $OUT/potrf.txt
timeout 300 python tools/timeline.py R > $OUT/timeline_R.txt 2>&1; cat $OUT/timeline_R.txt | head -60
timeout 300 bash tools/quick_prof.sh R r04b_R > /dev/null 2>&1; head -30 gpurun_out/prof_r04b_R/kernel_stats_table.md
timeout 300 bash tools/quick_prof.sh LP r04b_LP > /dev/null 2>&1; head -30 gpurun_out/prof_r04b_LP/kernel_stats_table.md
timeout 900 python -m pytest tests/test_multirank_gpu.py "tests/test_gpu_parity.py::test_config4_parity_workload_meets_the_literal_north_star_bounds" "tests/test_gpu_parity.py::test_debug_backsub_needs_a_solved_step" -x -q -s 2>&1 | grep -v "^\[Gloo\]\|amdgpu.ids" | tail -25 | tee $OUT/pytest.txt
