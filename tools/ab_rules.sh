#!/bin/bash
# same-box A/B of per-shape tile rules in the replayed step graph (RCDM_SHAPE_RULES, igemm.hip find_shape_rule)
R2="1,640,1280,2560,3,3;1,2560,1280,2560,5,0"
R3="9,40960,320,320,9,1;9,40960,320,640,9,1;9,40960,320,960,9,1;9,20480,320,320,9,0"
R5="9,10240,320,320,9,0;9,10240,640,320,1,1;9,2560,1280,640,9,0;9,640,1280,2560,9,0"
run() { RCDM_SHAPE_RULES="$2" python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
run base "off"; run R2+R3 "$R2;$R3"; run R5 "$R5"; run R2+R3+R5 "$R2;$R3;$R5"
done
