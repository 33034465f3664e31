#!/bin/bash
# Collect the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/make_profiles.sh r01'
# Outputs land in gpurun_out/prof_<tag>/; tools/finish_profiles.sh <tag> turns them into profiles/*.md afterwards.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (no CPU leg: it would only add host time)
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 2 > $OUT/stats_bench.log 2>&1
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (counters only, with --kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/fetch_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- python $ROOT/bench.py --config L --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/write_bench.log 2>&1
# 2b. instruction mix / pipe occupancy of the kernels (three SQ counter passes, tools/pmc_mix.sh)
bash $ROOT/tools/pmc_mix.sh L $TAG > /dev/null 2>&1
cp $ROOT/gpurun_out/pmc_$TAG/mix.md $OUT/pmc_mix.md 2>/dev/null
cd /tmp
# 3. summaries while the databases are at hand (the .db files are too big to bring back)
python $ROOT/tools/rocprof_summary.py $(find $OUT/stats -name "*.db" | head -1) $OUT/kernel_stats_table.md > /dev/null
python $ROOT/tools/pmc_summary.py $(find $OUT/fetch -name "*.db" | head -1) $(find $OUT/write -name "*.db" | head -1) $OUT/pmc_table.md $OUT/pmc_traffic_L.json > /dev/null
grep "^{\"metric\"" $OUT/stats_bench.log | tail -1 > $OUT/stats_bench_line.json
rm -rf $OUT/stats $OUT/fetch $OUT/write
# 4. bench lines with the CPU leg
cd $ROOT
python bench.py --config L --steps 20 --warmup 5 2> $OUT/bench_L.err | tail -1 > $OUT/bench_L.json      # (the driver's command: incl. its own rocprofv3 passes)
for cfg in S K X R LP; do
  XRSFM_BENCH_SELFPROF=0 python bench.py --config $cfg --steps 5 --warmup 2 2> $OUT/bench_$cfg.err | tail -1 > $OUT/bench_$cfg.json
done
python bench.py --config U --steps 3 --warmup 1 --no-cpu 2> $OUT/bench_U.err | tail -1 > $OUT/bench_U.json
python bench.py --config V --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_V.err | tail -1 > $OUT/bench_V.json
python bench.py --config L0 --steps 3 --warmup 1 --no-extras 2> $OUT/bench_L0.err | tail -1 > $OUT/bench_L0.json
python bench.py --config D --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_D.err | tail -1 > $OUT/bench_D.json
python bench.py --config T --steps 2 --warmup 1 --no-cpu --no-extras 2> $OUT/bench_T.err | tail -1 > $OUT/bench_T.json
python bench.py --config T --steps 1 --warmup 0 --no-cpu --no-extras --solver pcg 2> $OUT/bench_T_pcg.err | tail -1 > $OUT/bench_T_pcg.json
python bench.py --config M 2> $OUT/bench_M.err | tail -1 > $OUT/bench_M.json
python bench.py --config Lb9 --steps 3 --warmup 1 --no-extras 2> $OUT/bench_Lb9.err | tail -1 > $OUT/bench_Lb9.json
python tools/mapper_trace.py $OUT/mapper_trace.txt > /dev/null 2>&1
# 5. the dense reduced solve (config D): per-kernel table, and the sustained FP64 matrix-core rate of the instruction it uses
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsD -o stats -- python $ROOT/bench.py --config D --no-cpu --no-extras --steps 1 --warmup 1 > $OUT/statsD_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsD -name "*.db" | head -1) $OUT/kernel_stats_table_D.md > /dev/null; rm -rf $OUT/statsD )
[ -x tools/bench_mfma ] && tools/bench_mfma > $OUT/mfma_rate.txt 2>&1
# 6. round 4: the ragged configuration's kernel table, the pivot-tile microbench, packing on host vs device, the adapter's share
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsR -o stats -- python $ROOT/bench.py --config R --no-cpu --no-extras --steps 2 > $OUT/statsR_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsR -name "*.db" | head -1) $OUT/kernel_stats_table_R.md > /dev/null; rm -rf $OUT/statsR )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsB -o stats -- python $ROOT/bench.py --config Lb9 --no-cpu --no-extras --steps 2 > $OUT/statsB_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsB -name "*.db" | head -1) $OUT/kernel_stats_table_Lb9.md > /dev/null; rm -rf $OUT/statsB )
( cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/statsT -o stats -- python $ROOT/bench.py --config T --no-cpu --no-extras --steps 1 --warmup 1 > $OUT/statsT_bench.log 2>&1; \
  python $ROOT/tools/rocprof_summary.py $(find $OUT/statsT -name "*.db" | head -1) $OUT/kernel_stats_table_T.md > /dev/null; rm -rf $OUT/statsT )
[ -x tools/bench_potrf ] && tools/bench_potrf > $OUT/potrf.txt 2>&1
[ -x tools/bench_lat ] && tools/bench_lat > $OUT/lat.txt 2>&1
[ -x tools/bench_pipes ] && tools/bench_pipes > $OUT/pipes.txt 2>&1
python tools/pack_crossover.py > $OUT/pack_crossover.txt 2>&1
XRSFM_BA_PACK_TIMING=1 python tools/pack_phases.py L 2>&1 | tail -22 > $OUT/pack_phases.txt
python tools/adapter_timing.py L > $OUT/adapter_timing.txt 2>&1
python tools/lba_phases.py > $OUT/lba_phases.txt 2>&1
python tools/lba_timing.py > $OUT/lba_timing.txt 2>&1
python __graft_entry__.py probe 2>&1 | grep "\[probe\]" > $OUT/probe.txt
nproc > $OUT/nproc.txt
ls -la $OUT
