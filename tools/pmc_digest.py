#!/usr/bin/env python
"""Digest of tools/pmc_kernel.sh outputs (gpurun_out/<tag>/pmc.txt): per (kernel, grid) the raw SQ / GRBM counter averages
per launch and the derived figures DESIGN.md quotes — kernel clocks per XCD, MFMA-pipe occupancy
(SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs), share of wave cycles parked (SQ_WAIT_ANY) and
issue-stalled (SQ_WAIT_INST_ANY), LDS bank-conflict share.   usage: pmc_digest.py <pmc.txt> [name-filter ...]"""
import collections
import re
import sys

rows = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    m = re.match(r"(?:void )?(.+?)\s+blocks=\s*(\d+)\s+(\w+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
    if m and (len(sys.argv) < 3 or any(f in m.group(1) for f in sys.argv[2:])):
        rows[(m.group(1).strip(), int(m.group(2)))][m.group(3)] = float(m.group(5))
for (k, blocks), c in sorted(rows.items()):
    print(f"== {k}  ({blocks} blocks)")
    for name in sorted(c):
        print(f"   {name:28s} {c[name]:16.0f}")
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
        clk = c["GRBM_GUI_ACTIVE"] / 8
        out = [f"kernel {clk:.0f} clk per XCD", f"MFMA pipe busy {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / clk:.1f} %"]
        if "SQ_WAIT_ANY" in c:
            out.append(f"waves parked {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %")
        if "SQ_WAIT_INST_ANY" in c:
            out.append(f"issue-stalled {100 * c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.0f} %")
        if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
            out.append(f"VALU-active {100 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES']:.0f} % of wave cycles")
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_ACTIVE_INST_LDS"):
            out.append(f"LDS bank-conflict cycles {100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_ACTIVE_INST_LDS']:.0f} % of LDS-active")
        print("   -> " + "; ".join(out))
