#!/bin/bash
# Samples the GPU's shader clock, power and temperature (rocm-smi, ~10 Hz) while bench.py runs, to tell a power / thermal
# clock cap from a code difference when two boxes time the same binary differently (two populations of boxes).
# usage: tools/clock_probe.sh [bench.py args]   -> gpurun_out/clock_probe.txt
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/clock_probe.txt
: > "$out"
{
  echo "== static"
  rocm-smi --showperflevel --showpowercap --showmaxpower --showclocks --showtemp 2>&1 | grep -v "^$" | head -60
} >> "$out"
python bench.py --no-cpu-baseline --no-extra-configs "$@" > gpurun_out/clock_probe_bench.json 2> gpurun_out/clock_probe_bench.err &
bp=$!
: > gpurun_out/clock_probe_samples.txt
while kill -0 $bp 2>/dev/null; do
  s=$(rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (junction|memory)|GPU use" | sed -e 's/^GPU\[0\]\s*:\s*//' | tr '\n' '|')
  echo "$(date +%s.%N | cut -c1-14) $s" >> gpurun_out/clock_probe_samples.txt
done
wait $bp
{
  echo "== bench line"
  cat gpurun_out/clock_probe_bench.json
  echo "== samples (every 10th)"
  awk 'NR % 10 == 1' gpurun_out/clock_probe_samples.txt | head -150
  echo "== sclk histogram over all samples"
  grep -o "sclk clock level: [0-9]* (\([0-9]*\)Mhz)" gpurun_out/clock_probe_samples.txt | grep -o "([0-9]*Mhz)" | sort | uniq -c | sort -k2 -t'(' -n
} >> "$out"
