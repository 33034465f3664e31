"""Where a mapper-shaped run spends its BA time: replays tests/shim/mapper_main.cc once with XRSFM_BA_TRACE_CALLS=1 and
summarises the per-call lines (sizes, create / run / download, LM steps) by call size.  Usage: python tools/mapper_trace.py [out.txt]"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xrsfm_amd import mapper_replay  # noqa: E402


def main():
    exe = mapper_replay.build()
    arr = mapper_replay.sequence_problem(300)
    env = dict(os.environ, XRSFM_BA_TRACE_CALLS="1")
    if len(sys.argv) > 2:
        env["XRSFM_BA_PACK_TIMING"] = "1"
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        mapper_replay.dump(arr, inp)
        p = subprocess.run([exe, inp, out, "2"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env, timeout=1800)
    rows = []
    for line in p.stderr.splitlines():
        m = re.match(r"\[xrsfm_ba_solve\] cams (\d+) points (\d+) obs (\d+) \| create ([\d.]+) run ([\d.]+) download ([\d.]+) destroy ([\d.]+) ms \| LM (\d+)\+(\d+) solver (\d+)", line)
        if m:
            rows.append([float(x) for x in m.groups()])
    rows = np.array(rows)
    half = rows[len(rows) // 2:]                         # second replay: warm caches
    lines = [f"{len(rows)} xrsfm_ba_solve calls traced (two replays); second replay below", ""]
    lines.append("| calls | cams | points | obs | create ms | run ms | download ms | destroy ms | LM steps |")
    lines.append("|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for lo, hi in ((0, 12), (12, 24), (24, 40), (40, 80), (80, 10 ** 9)):
        sel = half[(half[:, 0] >= lo) & (half[:, 0] < hi)]
        if len(sel):
            med = np.median(sel, axis=0)
            lines.append(f"| {len(sel)} | {med[0]:.0f} | {med[1]:.0f} | {med[2]:.0f} | {med[3]:.3f} | {med[4]:.3f} | {med[5]:.3f} | {med[6]:.3f} | {med[7] + med[8]:.0f} |")
    lines.append("")
    lines.append(f"sum over the second replay: create {half[:, 3].sum():.1f} ms, run {half[:, 4].sum():.1f} ms, download {half[:, 5].sum():.1f} ms, destroy {half[:, 6].sum():.1f} ms")
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n\n" + "\n".join(p.stderr.splitlines()[-400:]) + "\n")


if __name__ == "__main__":
    main()
