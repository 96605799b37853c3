#!/usr/bin/env python
"""Kernel-start timeline of the deferred-LayerNorm consumer (igemm_dma, rcdm_gemm_lnx) beside the plain bias GEMM of the same
shape: per block, ticks from kernel start to "statistics loads issued", "prologue DMA issued", "statistics finished", block
duration and epilogue.  Needs a -DRCDM_TRACE_LX build:
    python -c "from rcdms_amd import build; print(build.build_variant('dbg', ['-DRCDM_TRACE_LX']))"
    RCDM_LIB=$PWD/rcdms_amd/lib/librcdm_dbg.so python tools/trace_lnx.py M N K"""
import sys, os, torch
sys.path.insert(0, "/root/repo")
from rcdms_amd import hip
M, N, K = [int(x) for x in sys.argv[1:4]]
A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.float16)
S = W.float().sum(1).contiguous()
parts = 20 if K == 1280 else 5
stat = torch.rand(parts * M * 2, device="cuda") + 1.0
d = hip.GemmDesc(M, N, K, K, N, 0, 1, 1, 0, 1.0, 1)
lx = hip.Lnx(0, 0, 0, stat.data_ptr(), parts, M, S.data_ptr(), 1e-5, K)
def run(lnx):
    if lnx: hip.gemm_lnx(d, lx, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
    else: hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
for lnx in (0, 1, 0, 1):
    tr = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    for _ in range(50): run(lnx)
    torch.cuda.synchronize()
    hip.load().rcdm_debug_set_igemm_trace(tr.data_ptr())
    run(lnx)
    torch.cuda.synchronize()
    hip.load().rcdm_debug_set_igemm_trace(0)
    t = tr.view(-1, 8).cpu(); t = t[t[:, 3] > 0].double()
    if not len(t):
        print("lnx", lnx, "no trace"); continue
    dur = t[:, 1] - t[:, 0]; span = t[:, 1].max() - t[:, 0].min()
    print(f"lnx={lnx}: blocks {len(t)}, kernel span {span:.0f} ticks, block dur med {dur.median():.0f}, first start->last start {t[:,0].max()-t[:,0].min():.0f}, epilogue {t[:,2].median():.0f}; start -> stat loads issued {t[:,4].median():.0f}, -> prologue DMA issued {t[:,5].median():.0f}, -> stats finished {t[:,6].median():.0f}; (dur - epi) / steps {((dur-t[:,2])/t[:,3]).median():.0f}")
