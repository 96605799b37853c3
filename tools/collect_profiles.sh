#!/bin/bash
# Round profile on the GPU box (run through gpurun):  bash tools/collect_profiles.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of a short bench.py run  -> gpurun_out/<tag>/kernel_stats.txt
#   2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (no trace domains) -> gpurun_out/<tag>/hbm_traffic.json
# Copy the two summaries into profiles/ afterwards (gpurun_out/ is scratch).
set -u
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
timeout 400 rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --no-extra-configs > "$out/trace.log" 2>&1
python tools/prof_summary.py "$out/trace" 5 < /dev/null > "$out/kernel_stats.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c -d "$out/pmc_$c" -o p -- python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-extra-configs --no-graph > "$out/pmc_$c.log" 2>&1
done
python tools/hbm_traffic.py "$out" 2 < /dev/null > "$out/hbm_traffic.json"
rm -rf "$out/trace" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE   # raw rocpd databases: tens of MB, summaries are kept
head -30 "$out/kernel_stats.txt"; cat "$out/hbm_traffic.json"
