#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace (rocpd .db or *_kernel_trace.csv) into per-kernel totals.
usage: prof_summary.py <dir-or-file> [n_steps]   (n_steps: UNet steps in the run, to print ms/step)"""
import csv
import glob
import os
import sqlite3
import sys


def rows_from(path):
    if os.path.isdir(path):
        dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
        csvs = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        path = (csvs or dbs or [None])[0]
    if path is None:
        raise SystemExit("no trace found")
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        return c.execute("select name, end-start from kernels").fetchall()
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for a, b in (("_ZN12_GLOBAL__N_1", ""),):
        n = n.replace(a, b)
    return n.split("(")[0][:60]


def main():
    rows = rows_from(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
    # one-off kernels of model construction (weight init, packing, host-tensor copies) are not part of a denoising step
    setup = ("pack_", "distribution_elementwise", "vectorized_elementwise", "CatArrayBatchedCopy", "copyBuffer", "fillBuffer",
             "elementwise_kernel", "xattn_pack", "reduce_kernel<", "matmul_f32_kernel")
    n_setup = sum(1 for name, _ in rows if any(k in name for k in setup))
    t_setup = sum(d for name, d in rows if any(k in name for k in setup))
    rows = [(n, d) for n, d in rows if not any(k in n for k in setup)]
    print(f"(left out: {n_setup} set-up dispatches — weight init / packing / copies — {t_setup / 1e6:.2f} ms in total)")
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(short(name), [0, 0, 1e18, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"total kernel time {tot / 1e6:.2f} ms over {sum(a[0] for a in agg.values())} dispatches" +
          (f"  ({tot / 1e6 / steps:.3f} ms per UNet step over {steps:g} steps)" if steps else ""))
    print(f"{'kernel':60s} {'calls':>7s} {'total_ms':>10s} {'%':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{k:60s} {a[0]:7d} {a[1] / 1e6:10.2f} {100 * a[1] / tot:6.1f} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f}")


if __name__ == "__main__":
    main()
