#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) of a rocprofv3 rocpd SQLite database.

usage: rocprof_summary.py <results.db> [out.md]
The database comes from  `rocprofv3 --kernel-trace --stats -d DIR -o NAME -- <cmd>`  (ROCm 7.2 writes rocpd by default).
"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "display_name" if "display_name" in sym_cols else ("kernel_name" if "kernel_name" in sym_cols else sym_cols[1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1 order by 3 desc")
    rows = cur.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, n, tot, avg, mn, mx in rows:
        short = re.sub(r"\(.*", "", name)
        short = re.sub(r"^void ", "", short)
        lines.append(f"| `{short[:70]}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} |")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
