#!/usr/bin/env python
"""Weights-cold GEMM timing: the same GEMM launched over a rotation of many distinct weight matrices (more bytes than the L2s
and the memory-side cache hold), as in the step graph where every weight matrix is read exactly once per step; activations stay
warm.  Per tile variant: us per launch warm (one weight matrix) | cold (rotation).
usage: python tools/coldw_bench.py [--variants 1,3,4,5,9,10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402

DEV = "cuda"
SHAPES = [("L1 CxC", 10240, 640, 640, 5), ("L2 CxC", 2560, 1280, 1280, 5), ("L3 CxC", 640, 1280, 1280, 5),
          ("L2 qkv", 2560, 3840, 1280, 0), ("L2 ffz K=6400", 2560, 1280, 6400, 5), ("L3 qkv", 640, 3840, 1280, 0),
          ("L1 ffz K=3200", 10240, 640, 3200, 5)]


def timeit(fn, n, rounds=5):
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / n * 1e3)
    best.sort()
    return best[len(best) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="-1,1,3,4,5,9,10")
    a = ap.parse_args()
    vs = [int(v) for v in a.variants.split(",")]
    for name, M, N, K, epi in SHAPES:
        ncopy = max(8, int(700e6 // (N * K * 2)))          # >= 700 MB of weights in the rotation
        A = torch.randn(M, K, device=DEV).half()
        Ws = [(torch.randn(N, K, device=DEV) * K ** -0.5).half() for _ in range(ncopy)]
        bias = torch.randn(N, device=DEV)
        res = torch.randn(M, N, device=DEV).half()
        out = torch.empty(M, N, device=DEV, dtype=torch.float16)
        line = f"{name:16s} M={M:5d} N={N:5d} K={K:5d} x{ncopy:3d} |"
        for v in vs:
            hip.set_igemm_variant(v)
            d = hip.GemmDesc(M, N, K, K, N, N, epi, 1, 0, 1.0, 0)
            try:
                wsb = hip.gemm_workspace_bytes(d)
                ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
                fn = lambda i, one=False: hip.gemm(d, A.data_ptr(), Ws[0 if one else i % ncopy].data_ptr(), bias.data_ptr(), 0, res.data_ptr(),
                                                   out.data_ptr(), ws.data_ptr(), ws.numel())
                warm = timeit(lambda i: fn(i, True), 50)
                cold = timeit(fn, ncopy)
                line += f" v{v}: {warm:5.1f} | {cold:5.1f} |"
            except hip.RcdmError:
                line += f" v{v}: refused |"
        print(line, flush=True)
    hip.set_igemm_variant(-1)


if __name__ == "__main__":
    main()
