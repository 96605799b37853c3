#!/bin/bash
# same-box A/B of per-shape tile rules in the replayed step graph (RCDM_SHAPE_RULES, igemm.hip find_shape_rule)
#   usage: bash tools/ab_rules.sh "<rules A>" ["<rules B>" ...]   (each against the library's table alone, two rounds)
run() { RCDM_SHAPE_RULES="$2" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  run base ""
  i=0
  for r in "$@"; do i=$((i+1)); run "set$i" "$r"; done
done
