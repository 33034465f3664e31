#!/bin/bash
# round 6: the mapper-side lines at the final HEAD (pinned staging everywhere): bench M, adapter timing, LBA phases / timing
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_final_m
mkdir -p $OUT
cd $ROOT
python bench.py --config M 2> /dev/null | tail -1 > $OUT/bench_M.json
python tools/mapper_trace.py $OUT/mapper_trace.txt > /dev/null 2>&1
python tools/adapter_timing.py L > $OUT/adapter_timing.txt 2>&1
python tools/lba_phases.py > $OUT/lba_phases.txt 2>&1
python tools/lba_timing.py > $OUT/lba_timing.txt 2>&1
python bench.py --config L --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('host_inclusive', d['host_inclusive'])"
tail -3 $OUT/adapter_timing.txt; cat $OUT/lba_phases.txt $OUT/lba_timing.txt | grep -v amdgpu
