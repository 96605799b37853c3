#!/usr/bin/env python
"""Per resolution level (64² / 32² / 16² / 8² rows of the b = 2, f = 5 step) and op kind: ms per step and launches, from a
tools/opprof.py table (run it with --top 1000 so that every op is listed).   usage: level_summary.py <opprof.txt>"""
import re
import sys

KINDS = ("gemm_ln", "gemm", "conv3x3_wino", "conv3x3", "upsample_gather", "rowchain", "flash_attn", "xattn", "temporal_attn", "groupnorm_stats", "groupnorm",
         "layernorm", "ff_fused")


def level(tag):
    kv = dict(re.findall(r"(\w+)=(-?\d+)", tag))
    kind = tag.split()[0]
    if kind in ("gemm", "gemm_ln", "rowchain", "layernorm", "ff_fused"):
        M = int(kv["M"])
    elif kind == "upsample_gather":
        n, H, W = map(int, re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        M = n * H * W * 4
    elif kind == "conv3x3_wino":
        n, H, W = map(int, re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        M = n * H * W
    elif kind == "conv3x3":
        n, H, W = map(int, re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        M = n * H * W * (4 if int(kv.get("up", 0)) else 1) // int(kv.get("s", 1)) ** 2
    elif kind in ("flash_attn", "xattn"):
        M = int(kv["B"]) * int(kv["Lq"])
    elif kind == "temporal_attn":
        M = int(kv["S"]) * int(kv["F"]) * int(kv["P"])
    elif kind in ("groupnorm", "groupnorm_stats"):
        M = int(kv["S"]) * int(kv["R"])
    else:
        return "other"
    for name, rows in (("64^2", 40960), ("32^2", 10240), ("16^2", 2560), ("8^2", 640)):
        if M >= rows * 0.45:   # the shared CFG prefix runs some ops on half the rows
            return name
    return "tiny"


agg, lv = {}, {}
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)\s+[\d.]+%\s+(\d+)\s", line)
    k = re.search(r"\b(%s)\b .*$" % "|".join(KINDS), line)
    if not m:
        continue
    tag = k.group(0) if k else "other"
    key = (level(tag), tag.split()[0])
    a = agg.setdefault(key, [0.0, 0])
    a[0] += float(m.group(1))
    a[1] += int(m.group(2))
for (l, k), (ms, n) in sorted(agg.items()):
    print(f"{l:5s} {k:16s} {ms:7.3f} ms {n:4d} launches")
    t = lv.setdefault(l, [0.0, 0])
    t[0] += ms
    t[1] += n
print()
for l, (ms, n) in sorted(lv.items()):
    print(f"{l:5s} total            {ms:7.3f} ms {n:4d} launches")
