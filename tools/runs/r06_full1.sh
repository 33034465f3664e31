#!/bin/bash
# round 6: full GPU test suite + smoke + default bench line (mid-round check)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06_full1
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $OUT/pytest_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
for dep in 0 2; do
  XRSFM_BA_LA_DEPTH=$dep timeout 900 python bench.py --config T --no-cpu --no-extras --steps 2 --warmup 1 2>/dev/null | grep '^{"metric"' > $OUT/bench_T_$dep.json
  echo "T depth=$dep $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_T_$dep.json)"
done
cd /tmp && export TMPDIR=/tmp
# what the GPU runs around a slow pose refinement of the mapper replay
XRSFM_BENCH_SELFPROF=0 rocprofv3 --kernel-trace -d $OUT/tr -o tr -- python $ROOT/bench.py --config M > $OUT/bench_M_trace.log 2>&1
DB=$(find $OUT/tr -name "*.db" | head -1)
python - "$DB" <<'PY' | tee $OUT/refine_kernels.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
ref = [(i, r) for i, r in enumerate(rows) if "k_refine_pose" in r[0]]
durs = sorted((r[2] - r[1]) / 1e3 for _, r in ref)
print("k_refine_pose launches", len(ref), "median us", durs[len(durs) // 2], "max us", durs[-1])
for i, r in ref:
    d = (r[2] - r[1]) / 1e3
    if d > 2000:
        prev = rows[i - 1]
        print(f"slow refine kernel {d:.0f} us; previous kernel {prev[0][:50]} ended {(r[1] - prev[2]) / 1e3:.0f} us before its start")
PY
rm -rf $OUT/tr
