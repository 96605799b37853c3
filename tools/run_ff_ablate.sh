#!/bin/bash
# rowff ablation builds (rcdms_amd.build.build_variant("ffaN", ["-DRCDM_FF_ABLATE=N"])) timed on the L0 feed-forward
for v in hip ffpd2 ffa1 ffa2 ffa4 ffa6 ffa38 ffa16 ffa32; do
  [ -f rcdms_amd/lib/librcdm_$v.so ] || continue
  echo -n "$v: "; RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_$v.so timeout 200 python tools/kbench.py ff --rounds 5 --only "L0 FF C=320" 2>&1 | grep "ff " | sed "s/.*chain/chain/" | tr "\n" ";"; echo
done
