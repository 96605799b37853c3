#!/usr/bin/env python
"""Epilogue phase breakdown of one igemm launch.  Needs a debug build of the library (8 int64 per block: {start, end,
epilogue, k-steps, wait-for-first-barrier, staging, pre-store wait, tiles}):
    python -c "from rcdms_amd import build; print(build.build_variant('dbg', ['-DRCDM_TRACE_PHASES']))"
    RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_dbg.so python tools/trace_phases.py M N K epi [variant]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip
M, N, K, epi = [int(x) for x in sys.argv[1:5]]
v = int(sys.argv[5]) if len(sys.argv) > 5 else -1
A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda"); nout = N // 2 if epi & 8 else N
res = torch.randn(M, nout, device="cuda").half(); out = torch.empty(M, nout, device="cuda", dtype=torch.float16)
hip.set_igemm_variant(v)
d = hip.GemmDesc(M, N, K, K, nout, nout, epi, 1, 0, 1.0, 1)
tr = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
for _ in range(3):
    hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), 0, 0)
torch.cuda.synchronize()
hip.load().rcdm_debug_set_igemm_trace(tr.data_ptr())
hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), 0, 0)
torch.cuda.synchronize()
hip.load().rcdm_debug_set_igemm_trace(0)
t = tr.view(-1, 8).cpu()
t = t[t[:, 3] > 0].double()
tiles = t[:, 7]
dur = t[:, 1] - t[:, 0]   # (float64: s_memtime values are ~1e13)
print(f"M={M} N={N} K={K} epi={epi} v={v}: blocks {len(t)}, tiles/block {tiles.median():.0f}, k-steps/tile {(t[:,3]/tiles).median():.0f}")
print(f"  per tile: total {(dur/tiles).median():.0f} ticks; epilogue {(t[:,2]/tiles).median():.0f} = first barrier {(t[:,4]/tiles).median():.0f}"
      f" + staging {(t[:,5]/tiles).median():.0f} + pre-store wait {(t[:,6]/tiles).median():.0f}"
      f" + post {((t[:,2]-t[:,4]-t[:,5]-t[:,6])/tiles).median():.0f}; k-loop per step {((dur-t[:,2])/t[:,3]).median():.0f}")
