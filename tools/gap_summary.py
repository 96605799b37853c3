#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run (rocpd .db): for the LAST `n` kernels of the
trace (graph replays of the denoising step), sum of kernel durations vs wall span.  usage: gap_summary.py <dir> [n]"""
import glob, os, sqlite3, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
rows = sorted(c.execute(f"select start, end from {kd}").fetchall())[-n:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e in rows)
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print(f"{len(rows)} kernels: span {span / 1e6:.3f} ms, sum of durations {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %), "
      f"positive gaps {sum(pos) / 1e6:.3f} ms (mean {sum(pos) / max(len(pos), 1) / 1e3:.2f} us, n={len(pos)}), "
      f"overlaps {-sum(g for g in gaps if g < 0) / 1e6:.3f} ms")
