#!/usr/bin/env python
"""The vendor libraries on the UNet's hot shapes, beside tools/kbench.py's numbers for the same shapes on the same box:
torch.mm (hipBLASLt / rocBLAS) for the GEMMs — bare, and with the epilogue this library fuses done the way a framework
would do it on top of the library call (bias + residual add, or the GEGLU gating pass) — and F.conv2d (MIOpen,
channels_last f16) for the convs.  usage: python tools/libbench.py [--only substring]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kbench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
a = ap.parse_args()

for name, M, N, K, epi in kbench.GEMMS:
    if a.only and a.only not in name:
        continue
    A = torch.randn(M, K, device="cuda").half()
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda").half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    bare, _ = kbench.timeit(lambda: torch.mm(A, W.t(), out=out), 5)
    if epi & 8:        # GEGLU: hidden * gelu(gate) over the two halves of the projection (diffusers GEGLU.forward)
        def fused():
            y = torch.addmm(bias, A, W.t())
            h, g = y.chunk(2, dim=-1)
            return h * F.gelu(g)
        what = "addmm + h*gelu(g)"
    elif epi & 4:      # + bias + residual
        res = torch.randn(M, N, device="cuda").half()

        def fused():
            return torch.addmm(bias, A, W.t()).add_(res)
        what = "addmm + residual"
    elif epi & 16:
        def fused():
            return F.gelu(torch.addmm(bias, A, W.t()))
        what = "addmm + gelu"
    elif epi & 1:
        def fused():
            return torch.addmm(bias, A, W.t())
        what = "addmm"
    else:
        fused, what = None, ""
    line = f"torch.mm {name:28s} {2.0 * M * N * K / 1e9:8.1f} GF | bare {bare:7.1f}us {2.0 * M * N * K / bare / 1e6:6.0f}TF"
    if fused is not None:
        t, _ = kbench.timeit(fused, 5)
        line += f" | {what}: {t:7.1f}us"
    print(line, flush=True)

for name, n, H, Wd, cin, cout in kbench.CONVS:
    if a.only and a.only not in name:
        continue
    x = torch.randn(n, cin, H, Wd, device="cuda").half().to(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * (9 * cin) ** -0.5).half().to(memory_format=torch.channels_last)
    med, _ = kbench.timeit(lambda: F.conv2d(x, w, padding=1), 5)
    fl = 2.0 * n * H * Wd * cin * cout * 9
    print(f"torch.conv {name:26s} {fl / 1e9:8.1f} GF | {med:7.1f}us {fl / med / 1e6:6.0f}TF", flush=True)
