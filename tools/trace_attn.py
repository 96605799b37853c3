#!/usr/bin/env python
"""Where a wave of the flash-attention kernel spends its loop (needs a -DRCDM_ATTN_TRACE build:
  python -c "from rcdms_amd import build; print(build.build_variant('attntrace', ['-DRCDM_ATTN_TRACE']))"
  RCDM_LIB=rcdms_amd/lib/librcdm_attntrace.so python tools/trace_attn.py):
per-wave s_memtime sums of the four phases of a key tile — barrier + K/V staging, QK^T issue, softmax, PV issue — for the
64x64 self-attention (B 10, H 8, L 4096, d 40), averaged over waves, in ticks per key tile."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402

B, H, L, d = (int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (10, 8, 4096, 40)))
dev = "cuda"
C = H * d
q = torch.randn(B * L, 3 * C, device=dev).half()
o = torch.empty(B * L, C, device=dev, dtype=torch.float16)
nblocks = (L + 127) // 128 * H * B
tr = torch.zeros(nblocks * 4 * 8, dtype=torch.int64, device=dev)
desc = hip.AttnDesc(B, H, L, L, d, 3 * C, 3 * C, 3 * C, C, d ** -0.5)
args = (desc, q.data_ptr(), q.data_ptr() + 2 * C, q.data_ptr() + 4 * C, o.data_ptr())
for _ in range(3):
    hip.flash_attn(*args)
hip.load().rcdm_debug_set_attn_trace(tr.data_ptr())
hip.flash_attn(*args)
torch.cuda.synchronize()
hip.load().rcdm_debug_set_attn_trace(0)
t = tr.view(-1, 8).double().cpu()
t = t[t[:, 5] > 0]
if not len(t):
    raise SystemExit("no trace records: not an RCDM_ATTN_TRACE build")
tiles = t[:, 5]
names = ["loop", "barrier + staging", "QK^T issue", "softmax", "PV issue"]
print(f"{len(t)} waves, {tiles.mean():.0f} key tiles each")
for i, n in enumerate(names):
    print(f"  {n:20s} {(t[:, i] / tiles).mean():8.1f} ticks per tile   (min {(t[:, i] / tiles).min():.1f}, max {(t[:, i] / tiles).max():.1f})")
