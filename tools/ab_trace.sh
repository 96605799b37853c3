#!/bin/bash
# Kernel-level A/B of an environment switch on one box: a rocprofv3 kernel trace of a short bench.py run per value, then the
# per-kernel difference (calls, total ms) — what an end-to-end A/B cannot tell: WHICH kernels paid for a change.
#   usage (through gpurun): bash tools/ab_trace.sh VAR A B      -> gpurun_out/abt_<VAR>_<value>/kernel_stats.txt + the diff on stdout
# NB: never use this (or any timing) on a build / switch that computes garbage: the step runs power-capped, and
# Inf / NaN operands toggle fewer wires — the whole step clocks higher and an "upper bound" comes out several times too big.
VAR=$1; A=$2; B=$3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for v in "$A" "$B"; do
  out=gpurun_out/abt_${VAR}_$v; mkdir -p "$out"
  env $VAR=$v timeout 400 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --no-extra-configs > "$out/trace.log" 2>&1
  python tools/prof_summary.py "$out/trace" 5 < /dev/null > "$out/kernel_stats.txt"
  rm -rf "$out/trace"
done
python - "gpurun_out/abt_${VAR}_$A/kernel_stats.txt" "gpurun_out/abt_${VAR}_$B/kernel_stats.txt" <<'PY'
import re, sys
def load(p):
    d = {}
    for l in open(p):
        m = re.match(r"(.+?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", l)
        if m:
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
for p in sys.argv[1:3]:
    print(p, open(p).readlines()[1].strip())
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    if abs(tb - ta) > 0.02 or ca != cb:
        rows.append((tb - ta, k, ca, ta, cb, tb))
for r in sorted(rows):
    print(f"{r[0] / 5:+7.3f} ms/step  {r[1][:72]:72s} {r[2]:4d} {r[3]:7.2f} -> {r[4]:4d} {r[5]:7.2f}")
PY
