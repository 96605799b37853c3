# A/B + ablation run of the ping-pong kernel (tools/kbench.py): product build and the -DRCDM_PP_ABLATE debug builds
# (rcdms_amd.build.build_variant('abl6' | 'abl1', ...)), with and without the per-block k rotation
O=gpurun_out/r2_pp5_kb.log
timeout 300 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or conv" 2>&1 | tail -4 > gpurun_out/r2_pp5_t.log
for rot in 0 1; do for lib in hip abl1 abl6; do
  echo "== rotate $rot lib $lib" >> $O
  RCDM_PP_ROTATE=$rot RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py conv --only "L0 960" --variants=-2,6 >> $O 2>&1
  RCDM_PP_ROTATE=$rot RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py conv --only "L1 1280" --variants=-2,6,8 >> $O 2>&1
  RCDM_PP_ROTATE=$rot RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py gemm --only "geglu N=5120" --variants=-2,6,7,8 >> $O 2>&1
done; done
RCDM_PP_ROTATE=0 timeout 300 python tools/kbench.py conv --variants=-2,6,8 >> $O 2>&1
RCDM_PP_ROTATE=1 timeout 300 python tools/kbench.py conv --variants=-2,6,8 >> $O 2>&1
RCDM_PP_ROTATE=1 timeout 300 python tools/kbench.py gemm --only "L" --variants=-2,6,7,8 >> $O 2>&1
