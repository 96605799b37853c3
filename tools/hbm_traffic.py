#!/usr/bin/env python
"""HBM bytes per UNet step from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units), restricted to the
kernels of the denoising step, with the gfx950 correction of MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B
requests as 64 B for wide coalesced reads -> doubled.  usage: hbm_traffic.py <dir> <n_steps>"""
import glob, json, os, sqlite3, sys
d, steps = sys.argv[1], float(sys.argv[2])
STEP_KERNELS = ("igemm", "wino_", "upsample_gather", "row_chain", "splitk_reduce", "flash_attn", "xattn_kernel", "temporal_attn", "layernorm", "gn_", "small_linear",
                "load_table_row",
                "timestep_embed", "assemble_input", "cfg_ddim", "load_timestep", "advance_step")
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob(os.path.join(d, "pmc_%s" % c, "**", "*.db"), recursive=True)[0]
    con = sqlite3.connect(db)
    tot = 0.0
    per = {}
    for name, v in con.execute("select kernel_name, value from counters_collection where counter_name=?", (c,)):
        if any(k in name for k in STEP_KERNELS):
            tot += v
            key = next(k for k in STEP_KERNELS if k in name)
            per[key] = per.get(key, 0.0) + v
    res[c] = {"kb_per_step_raw": tot / steps, "by_kernel_kb_per_step": {k: round(v / steps) for k, v in sorted(per.items(), key=lambda kv: -kv[1])}}
fetch = res["FETCH_SIZE"]["kb_per_step_raw"] * 1024 * 2   # gfx950: double the wide-read count
write = res["WRITE_SIZE"]["kb_per_step_raw"] * 1024
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import build as rbuild  # noqa: E402
print(json.dumps({"hbm_bytes_per_unet_step": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
                  "steps_in_run": steps, "csrc_stamp": rbuild._stamp(),   # bench.py drops the record when the kernels change
                  "raw": res}, indent=1))
