#!/bin/bash
# SQ counter passes over one kbench group (run through gpurun):  bash tools/pmc_kernel.sh <tag> <kbench-args...>
# Each pass is its own rocprofv3 --pmc run (no trace domains).  Summaries: gpurun_out/<tag>/pmc.txt
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p "$out"
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d "$out/p$i" -o p -- python tools/kbench.py "$@" --rounds 1 > "$out/p$i.log" 2>&1 < /dev/null
done
python tools/pmc_summary.py "$out" < /dev/null > "$out/pmc.txt"
rm -rf "$out"/p[0-9]   # raw rocpd databases
cat "$out/pmc.txt"
