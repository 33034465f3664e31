#!/usr/bin/env python3
"""Per-kernel averages of every counter of a rocprofv3 --pmc rocpd database (one table per pass).

usage: pmc_generic.py <results.db> [out.md]
"""
import collections
import re
import sqlite3
import sys


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_pmc_event)")]
    info_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_pmc)")]
    name_col = "name" if "name" in info_cols else info_cols[1]
    q = (f"select s.display_name, i.{name_col}, count(*), avg(p.value) from rocpd_pmc_event p "
         "join rocpd_kernel_dispatch d on p.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
         "join rocpd_info_pmc i on p.pmc_id = i.id group by 1, 2")
    table = collections.defaultdict(dict)
    counters = []
    launches = {}
    for kern, ctr, n, avg in cur.execute(q):
        k = re.sub(r"\(.*", "", kern).replace("void ", "")
        table[k][ctr] = avg
        launches[k] = n
        if ctr not in counters:
            counters.append(ctr)
    counters.sort()
    lines = ["| kernel | launches | " + " | ".join(counters) + " |", "|---|---:|" + "---:|" * len(counters)]
    key = counters[0] if counters else None
    for k in sorted(table, key=lambda x: -table[x].get(key, 0.0)):
        lines.append(f"| `{k[:48]}` | {launches[k]} | " + " | ".join(f"{table[k].get(c, 0.0):.4g}" for c in counters) + " |")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out + "\n\n")
    print(out)


if __name__ == "__main__":
    main()
