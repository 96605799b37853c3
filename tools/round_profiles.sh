#!/bin/bash
# Everything profiles/ holds for a round, on ONE box (run through gpurun):  bash tools/round_profiles.sh <tag>
#   bench line, kernel stats + HBM traffic (collect_profiles.sh), SQ counters over the real step (pmc_step.sh), per-op and
#   per-level tables (opprof.py, level_summary.py), determinism soak, the other BASELINE configurations.
set -u
tag=${1:-r5}
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
python bench.py > "$out/bench_line.json" 2> "$out/bench.err"
bash tools/collect_profiles.sh "$tag" > "$out/collect.log" 2>&1
bash tools/pmc_step.sh "${tag}_pmc" > "$out/pmc.log" 2>&1
python tools/opprof.py --top 1000 > "$out/opprof.txt" 2>&1
python tools/level_summary.py "$out/opprof.txt" > "$out/levels.txt" 2>&1
python tools/soak.py 8 > "$out/soak.txt" 2>&1
{   # (configs 1 / 3 / 5 and the reference-context story are in the bench line's extra_configs since round 6)
  python tools/bench_vae.py
} > "$out/other_configs.jsonl" 2> "$out/other.err"
tail -1 "$out/bench_line.json" | cut -c1-600; tail -3 "$out/soak.txt"; tail -6 "$out/levels.txt"; cat "$out/hbm_traffic.json" | head -8
