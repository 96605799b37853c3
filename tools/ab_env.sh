#!/bin/bash
# A/B of an environment switch on one box, interleaved bench.py runs: tools/ab_env.sh VAR A B [rounds]
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq $R); do for v in $A $B; do
  env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
