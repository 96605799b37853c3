for shape in "L0 GN cross C=320" "L0 GN frame" "L0 GN cross C=960" "L1 GN cross" "L1 GN frame" "L2 GN"; do
  for lib in ${LIBS:-oldepi hip}; do
    tag="gn_${lib}"
    RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$lib.so bash tools/prof_kbench.sh $tag norm --only "$shape" 
    echo "== $shape | $lib"; grep "gn_" gpurun_out/${tag}_kstats.txt | cut -c1-30,64-130
  done
done
