#!/bin/bash
# A/B of library builds on one box, interleaved bench.py runs: tools/ab_lib.sh <rounds> <name> [<name> ...]
# (names of rcdms_amd/lib/librcdm_<name>.so: "hip" is the product build, others come from rcdms_amd.build.build_variant)
cd "$(dirname "$0")/.."
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$v.so timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done
