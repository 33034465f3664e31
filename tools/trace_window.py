#!/usr/bin/env python3
"""Developer aid: a window of a rocprofv3 kernel trace as a timeline — start (us, relative), duration, queue, grid, kernel —
to see what overlaps what.  usage: trace_window.py <results.db> <kernel substring to anchor on> [occurrence] [count]"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    occ = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    count = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    grid = "d.grid_size_x" if "grid_size_x" in cols else "0"
    wg = "d.workgroup_size_x" if "workgroup_size_x" in cols else "1"
    queue = "d.queue_id" if "queue_id" in cols else ("d.stream_id" if "stream_id" in cols else "0")
    rows = cur.execute(f"select d.start, d.end, {queue}, {grid}, {wg}, s.display_name from rocpd_kernel_dispatch d "
                       "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[5]]
    if not idx:
        print("no such kernel; columns:", cols)
        return
    a = idx[min(occ, len(idx) - 1)]
    t0 = rows[a][0]
    for st, en, q, g, w, name in rows[a:a + count]:
        print(f"{(st - t0) / 1e3:9.2f} +{(en - st) / 1e3:7.2f} us  q{q}  wg {g // max(w, 1):5d}  {name[:60]}")


if __name__ == "__main__":
    main()
