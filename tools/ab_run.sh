#!/bin/bash
# A/B of two builds of the library on one box: rcdms_amd/lib/librcdm_old.so vs librcdm_hip.so
# (kernel micro-bench + two interleaved bench.py runs each).  Output: gpurun_out/ab.log + bench lines on stdout.
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_hip_kernels.py tests/test_hip_fuzz.py -x -q -m gpu 2>&1 | tail -3
for e in old new; do
  echo LIB=$e
  L=$PWD/rcdms_amd/lib/librcdm_hip.so; [ $e = old ] && L=$PWD/rcdms_amd/lib/librcdm_old.so
  RCDM_LIB=$L timeout 300 python tools/kbench.py gemm --variants -1 2>&1 | tail -26
  RCDM_LIB=$L timeout 300 python tools/kbench.py conv --variants -1 2>&1 | tail -14
done > gpurun_out/ab.log 2>&1
for e in old new old new; do
  L=$PWD/rcdms_amd/lib/librcdm_hip.so; [ $e = old ] && L=$PWD/rcdms_amd/lib/librcdm_old.so
  RCDM_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', d['value'], d['ms_per_step'])"
done
