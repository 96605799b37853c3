#!/bin/bash
# Socket power / shader clock / energy per step of two values of an environment switch on one box (VERDICT r5 #1d): bench.py
# runs under a ~10 Hz rocm-smi sampler per value; energy per step = mean power over the loop samples x ms per step.
#   usage (through gpurun): bash tools/energy_ab.sh VAR A B   -> gpurun_out/energy_<VAR>.txt
VAR=$1; A=$2; B=$3
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
out=gpurun_out/energy_${VAR}.txt; : > "$out"
for v in "$A" "$B" "$A" "$B"; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-extra-configs --steps 6 > gpurun_out/e_bench.json 2>/dev/null &
  bp=$!; : > gpurun_out/e_samples.txt
  while kill -0 $bp 2>/dev/null; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed -e 's/^GPU\[0\]\s*:\s*//' | tr '\n' '|' >> gpurun_out/e_samples.txt; echo >> gpurun_out/e_samples.txt
  done
  wait $bp
  python - "$VAR=$v" <<'PY' >> "$out"
import json, re, sys
d = json.loads(open("gpurun_out/e_bench.json").read().strip().splitlines()[-1])
pw, ck = [], []
for l in open("gpurun_out/e_samples.txt"):
    p = re.search(r"Package Power \(W\): ([\d.]+)", l); c = re.search(r"sclk clock level: \d+:? \((\d+)Mhz\)", l)
    if p and c and float(p.group(1)) > 700:     # samples inside the denoising loops (idle / set-up draw far less)
        pw.append(float(p.group(1))); ck.append(float(c.group(1)))
ms = d["roofline"]["avg_launch_ms"]
if pw:
    mp = sum(pw) / len(pw)
    print(f"{sys.argv[1]:28s} {ms:7.3f} ms per step  {mp:7.1f} W  {sum(ck) / len(ck):6.0f} MHz  {mp * ms / 1e3:6.2f} J per step  ({len(pw)} samples)")
else:
    print(f"{sys.argv[1]:28s} {ms:7.3f} ms per step  (no power samples)")
PY
done
cat "$out"
