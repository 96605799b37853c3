#!/bin/bash
# rocprofv3 kernel-trace summaries of a short bench.py run under two settings of an environment switch:
#   bash tools/trace_ab.sh VAR A B   ->  gpurun_out/trace_ab_<VAR>_<value>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=$1
for v in $2 $3; do
  out=gpurun_out/trace_ab_$v
  rm -rf $out; mkdir -p $out
  env $VAR=$v timeout 400 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline > $out/trace.log 2>&1
  python tools/prof_summary.py $out/trace 5 < /dev/null > gpurun_out/trace_ab_${VAR}_$v.txt
  python tools/prof_shapes.py $out/trace 5 < /dev/null > gpurun_out/trace_shapes_${VAR}_$v.txt
  rm -rf $out
done
