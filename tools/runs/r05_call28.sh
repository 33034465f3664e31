#!/bin/bash
# round 5, call 28: where an LBA-sized call's create / run time goes (phase timers of the host-side set-up; per-kernel trace of 20 calls)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_c28
mkdir -p $OUT
cd $ROOT
XRSFM_BA_PACK_TIMING=1 python tools/lba_phases.py > $OUT/phases.txt 2>&1
grep -E "^\[" $OUT/phases.txt | sort | uniq -c | sort -rn | head -5
python - <<'PY' > $OUT/avg.txt
import re,collections
acc=collections.OrderedDict(); n=collections.Counter()
for l in open("gpurun_out/r05_c28/phases.txt"):
    m=re.match(r"\[(.*?)\]\s+(.*?)\s+([0-9.]+) ms", l)
    if m:
        k=m.group(1)+" | "+m.group(2).strip(); acc[k]=acc.get(k,0)+float(m.group(3)); n[k]+=1
for k,v in acc.items(): print("%-70s %.4f ms x %d" % (k, v/n[k], n[k]))
PY
cat $OUT/avg.txt; tail -1 $OUT/phases.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o st -- python $ROOT/tools/lba_phases.py > $OUT/trace.log 2>&1
python $ROOT/tools/rocprof_summary.py $(find $OUT/st -name "*.db" | head -1) $OUT/table.md > /dev/null; rm -rf $OUT/st
head -30 $OUT/table.md
# RESULT: steady state (7 cameras / 6000 observations, 5 LM iterations): create 0.24 ms = host packing 0.14 + uploads 0.02 + buffers 0.02 + Cholesky set-up 0.05;
# run 0.39 ms = 57 us of kernels per iteration (k_lv_factor<true> 16.3, k_lin_tail 10.0 x 1.4, k_schur_pairs 7.9, k_linearize 6.4 x 1.4, k_backsub 5.5,
# k_chol_segsum 4.6: each one tile's latency) x 5 + launches; download 0.06.
