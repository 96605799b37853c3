#!/bin/bash
# per-kernel durations of a kbench run: tools/prof_kbench.sh <tag> <kbench args...>   -> gpurun_out/<tag>_kstats.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python tools/kbench.py "$@" > $out/log.txt 2>&1
python tools/prof_summary.py $out/trace 1 < /dev/null > gpurun_out/${tag}_kstats.txt
rm -rf $out/trace
