#!/usr/bin/env python3
"""Developer aid: durations (us) and grid sizes of the first N launches of one kernel in a rocprofv3 rocpd database.

usage: launch_series.py <results.db> <kernel substring> [N]
"""
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    grid = "d.grid_size_x" if "grid_size_x" in cols else ("d.grid_x" if "grid_x" in cols else "0")
    wg = "d.workgroup_size_x" if "workgroup_size_x" in cols else "1"
    q = (f"select d.start, d.end - d.start, {grid}, {wg} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "where s.display_name like ? order by d.start limit ?")
    rows = cur.execute(q, (f"%{sys.argv[2]}%", n)).fetchall()
    for i, (st, dur, g, w) in enumerate(rows):
        print(f"{i:4d} {dur / 1e3:9.2f} us   workgroups {g // max(w, 1) if w else g}")


if __name__ == "__main__":
    main()
