#!/usr/bin/env python
"""Idle time inside and between the replays of the step graph, from a rocprofv3 kernel trace of bench.py:
per step (advance_step ends it) the span, the sum of kernel durations, the idle time between kernels of the step and the
gap to the first kernel of the next replay.  usage: step_gaps.py <trace dir>"""
import glob
import os
import sqlite3
import sys

db = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
steps, cur = [], []
for name, st, en in rows:
    cur.append((name, st, en))
    if "advance_step" in name:
        steps.append(cur)
        cur = []
steps = [s for s in steps if len(s) > 300]          # replays of the full step graph only
for i, s in enumerate(steps[-6:]):
    span = s[-1][2] - s[0][1]
    busy = sum(en - st for _, st, en in s)
    idle = sum(max(0, s[k + 1][1] - s[k][2]) for k in range(len(s) - 1))
    print(f"step {i}: {len(s)} kernels, span {span / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle between kernels {idle / 1e6:.3f} ms")
for a, b in zip(steps[-6:], steps[-5:]):
    print(f"gap to next replay: {(b[0][1] - a[-1][2]) / 1e3:.1f} us")
