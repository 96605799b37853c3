#!/bin/bash
# SQ counter passes over the kernels of the REAL step (run through gpurun):  bash tools/pmc_step.sh <tag>
# Four separate rocprofv3 --pmc runs (no trace domains) of a two-step eager bench.py run: every kernel of the denoising step
# at its own shapes, cold operands, in the step's own order.  Summary: gpurun_out/<tag>/pmc.txt, digest: pmc_digest.txt
set -u
tag=${1:-r5pmc}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set -d "$out/p$i" -o p -- python bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-extra-configs --no-graph > "$out/p$i.log" 2>&1 < /dev/null
done
python tools/pmc_summary.py "$out" < /dev/null > "$out/pmc.txt"
rm -rf "$out"/p[0-9]   # raw rocpd databases
python tools/pmc_digest.py "$out/pmc.txt" igemm row_chain flash_attn temporal xattn splitk gn_ > "$out/pmc_digest.txt"
grep -c "" "$out/pmc.txt"; grep -- "->\|^==" "$out/pmc_digest.txt" | head -120
