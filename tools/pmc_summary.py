#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately).

usage: pmc_summary.py <fetch_results.db> <write_results.db> [out.md [out.json]]

Counters are in KiB per dispatch.  Correction applied (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, i.e. it reads 1/2 of the bytes of a coalesced streaming read — doubled here.  The guide
calibrates that for 16 B/lane; this code base streams 8 B/lane (FP64 SoA), so the factor was re-calibrated on kernels with a
known byte count: k_schur_prep (Fs+Es+Hinv+gp+idx = 330 MB expected, 166.5 MiB raw), k_backsub (380 MB expected, 190 raw),
k_cost (61 MB expected, 29.6 raw): ratio 1.98-2.05.  WRITE_SIZE needs no correction in this access pattern (k_linearize:
368 MB expected, 357 MiB counted).
"""
import re
import sqlite3
import sys


def per_kernel(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select s.display_name, count(*), avg(p.value) from rocpd_pmc_event p join rocpd_kernel_dispatch d on p.event_id = d.event_id "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1")
    return {re.sub(r"\(.*", "", n).replace("void ", ""): (c, a) for n, c, a in cur.execute(q)}


def main():
    f, w = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    rows = []
    for k in sorted(set(f) | set(w)):
        fr = f.get(k, (0, 0.0)); wr = w.get(k, (0, 0.0))
        rd = 2.0 * fr[1] * 1024 / 1e6; wt = wr[1] * 1024 / 1e6
        rows.append((rd + wt, k, fr[0] or wr[0], rd, wt))
    lines = ["| kernel | launches | read MB/launch (FETCH_SIZE x2) | write MB/launch (WRITE_SIZE) | traffic MB/launch |", "|---|---:|---:|---:|---:|"]
    for tot, k, n, rd, wt in sorted(rows, reverse=True):
        if tot >= 0.05:
            lines.append(f"| `{k[:60]}` | {n} | {rd:.1f} | {wt:.1f} | {tot:.1f} |")
    out = "\n".join(lines)
    if len(sys.argv) > 4:      # machine-readable copy for bench.py: kernel -> HBM bytes per launch
        import json
        with open(sys.argv[4], "w") as fh:
            json.dump({k.replace("xba::", ""): round(tot * 1e6) for tot, k, n, rd, wt in rows if tot >= 0.05}, fh, indent=1, sort_keys=True)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "a") as fh:
            fh.write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
