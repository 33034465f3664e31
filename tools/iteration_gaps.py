#!/usr/bin/env python3
"""Developer aid: the launch sequence of one LM iteration from a rocprofv3 rocpd database: start offset, duration and the idle
gap in front of every kernel (the gap in front of the first kernel after k_lin_tail is the host hand-off).

usage: iteration_gaps.py <results.db> [iteration index, default 8]
"""
import re
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = cur.execute("select s.display_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start").fetchall()
    rows = [(re.sub(r"\(.*", "", n).replace("void ", "").replace("xba::", ""), a, b) for n, a, b in rows]
    tails = [i for i, r in enumerate(rows) if r[0].startswith("k_lin_tail")]
    lo, hi = tails[which] + 1, tails[which + 1] + 1
    t0 = rows[lo - 1][2]
    busy = 0
    prev_end = t0
    for n, a, b in rows[lo:hi]:
        print(f"{(a - t0) / 1e3:9.2f} us  +{(b - a) / 1e3:7.2f}  gap {(a - prev_end) / 1e3:6.2f}  {n}")
        busy += b - a
        prev_end = b
    print(f"iteration: {(rows[hi - 1][2] - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle {(rows[hi - 1][2] - t0 - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
