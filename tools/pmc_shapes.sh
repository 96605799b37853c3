#!/bin/bash
# HBM bytes per launch of single kbench shapes (run through gpurun): bash tools/pmc_shapes.sh <tag> "<what>:<only-substring>" ...
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace domains); summary: gpurun_out/<tag>/shapes.txt
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for spec in "$@"; do
  i=$((i+1))
  what=${spec%%:*}; only=${spec#*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c -d "$out/s${i}_$c" -o p -- python tools/kbench.py $what --only "$only" --variants=-1 --rounds 1 > "$out/s${i}_$c.log" 2>&1 < /dev/null
  done
  echo "== $spec" >> "$out/shapes.txt"
  grep -h "TF\|TB/s" "$out/s${i}_FETCH_SIZE.log" >> "$out/shapes.txt"
  python tools/pmc_summary.py "$out/s${i}_FETCH_SIZE" igemm < /dev/null >> "$out/shapes.txt"
  python tools/pmc_summary.py "$out/s${i}_WRITE_SIZE" igemm < /dev/null >> "$out/shapes.txt"
done
cat "$out/shapes.txt"
