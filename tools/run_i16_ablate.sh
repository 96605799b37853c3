# Ablation of the 160x160 kernel (tools/kbench.py, variant 9): product build vs the -DRCDM_I16_ABLATE debug builds
# (rcdms_amd.build.build_variant): 1 = no epilogue, 2 = epilogue without global stores, 9 = no epilogue and no DMA
O=gpurun_out/i16_ablate.log; : > $O
for lib in ${LIBS:-hip i16a1 i16a2 i16a9}; do
  echo "== lib $lib" >> $O
  for s in "L0 qkv" "L0 geglu" "L1 geglu" "L2 geglu" "L1 qkv" "L2 qkv"; do
    RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py gemm --only "$s" --variants=9 >> $O 2>&1
  done
done
cat $O
