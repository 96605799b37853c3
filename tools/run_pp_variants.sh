# kbench of the ping-pong kernel under debug build variants (rcdms_amd.build.build_variant): tools/run_pp_variants.sh v1 v2 ...
O=gpurun_out/r2_ppv.log; rm -f $O
for rep in 1 2; do for lib in "$@"; do
  echo "== lib $lib" >> $O
  RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py conv --only "L0" --variants=6 >> $O 2>&1
  RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py conv --only "L1 1280" --variants=6,8 >> $O 2>&1
  RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so timeout 200 python tools/kbench.py gemm --only "L0 ff-out" --variants=6 >> $O 2>&1
done; done
