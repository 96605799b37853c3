#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd .db.  usage: pmc_summary.py <dir> [name-filter]"""
import glob, os, sqlite3, sys, collections
d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except Exception as e:
        print("no counters in", db, e); continue
    q = "select kernel_name, counter_name, value, grid_size, workgroup_size from counters_collection" if "kernel_name" in cols else None
    if q is None:
        print("columns:", cols); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for kn, cn, v, gs, ws in c.execute(q):
        if filt and filt not in kn: continue
        key = (kn.replace("(anonymous namespace)::", "").split("(")[0][:80], gs // max(ws, 1), cn)
        agg[key][0] += 1; agg[key][1] += v
    for (kn, blocks, cn), (n, tot) in sorted(agg.items()):
        print(f"{kn:82s} blocks={blocks:6d} {cn:28s} n={n:4d} avg={tot / n:16.1f}")
