#!/usr/bin/env python
"""Per-(kernel, grid) totals of a rocprofv3 kernel trace (rocpd .db): the in-graph time of every SHAPE of a kernel.
usage: prof_shapes.py <dir> [n_steps] [name-filter]"""
import glob, os, sqlite3, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
filt = sys.argv[3] if len(sys.argv) > 3 else ""
db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
gcols = [x for x in cols if "grid" in x.lower() or "workgroup" in x.lower()]
if "--cols" in sys.argv:
    print(cols); sys.exit(0)
q = "select name, end-start, %s from kernels" % ", ".join(gcols)
agg = {}
for row in c.execute(q):
    name, dur, geo = row[0], row[1], tuple(row[2:])
    if filt and filt not in name:
        continue
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    a = agg.setdefault((name, geo), [0, 0])
    a[0] += 1; a[1] += dur
print("geometry columns:", gcols)
for (name, geo), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{name:46s} {str(geo):34s} n/step={n / steps:6.1f} avg={tot / n / 1e3:8.1f} us  ms/step={tot / 1e6 / steps:7.3f}")
