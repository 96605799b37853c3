#!/usr/bin/env python
"""Per-block timing of one igemm launch via the s_memtime trace hook: block duration, epilogue share, per-k-step time."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip
M, N, K, epi = [int(x) for x in sys.argv[1:5]]
variants = [int(v) for v in sys.argv[5].split(",")] if len(sys.argv) > 5 else [1, 2]
A = torch.randn(M, K, device="cuda").half(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
bias = torch.randn(N, device="cuda"); nout = N // 2 if epi & 8 else N
res = torch.randn(M, nout, device="cuda").half(); out = torch.empty(M, nout, device="cuda", dtype=torch.float16)
for v in variants:
    hip.set_igemm_variant(v)
    d = hip.GemmDesc(M, N, K, K, nout, nout, epi, 1, 0, 1.0, 1)
    tr = torch.zeros(8192 * 4, dtype=torch.int64, device="cuda")
    # warm: a few hundred back-to-back launches bring the shader clock up (a launch after an idle gap runs at ~1.5 GHz
    # for its first milliseconds, tools/mfma_peak.py), then the traced launch is timed with events
    for _ in range(400):
        hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), 0, 0)
    hip.load().rcdm_debug_set_igemm_trace(tr.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), 0, 0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    hip.load().rcdm_debug_set_igemm_trace(0)
    t = tr.view(-1, 4).cpu()
    t = t[t[:, 3] > 0]
    dur = (t[:, 1] - t[:, 0]).float(); epi_t = t[:, 2].float(); steps = t[:, 3].float()
    span = int(dur.max())   # persistent blocks: the longest block spans (almost) the whole kernel
    print(f"v{v}: {us:.1f} us by events (incl. ~4 us of launch latency), longest block {span} ticks -> shader clock >= "
          f"{span / us / 1e3:.2f} GHz during this launch (s_memtime ticks are shader clocks, tools/mfma_peak.py)")
    print(f"v{v}: blocks {len(t)}, kernel span {span} ticks; per block: dur med {dur.median():.0f} (min {dur.min():.0f} max {dur.max():.0f}), "
          f"epilogue {epi_t.median():.0f} ({100 * (epi_t / dur).median():.0f}%), steps {steps.median():.0f}, "
          f"non-epilogue per k-step {((dur - epi_t) / steps).median():.0f} ticks")
hip.set_igemm_variant(-1)
