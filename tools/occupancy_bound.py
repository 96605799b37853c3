#!/usr/bin/env python
"""What a second wave per SIMD would buy the single-round GEMMs (one 4-wave block per CU): the same GEMM with M doubled
puts two blocks on every CU — if 2x the work takes much less than 2x the time, an 8-wave block that splits K between two
wave groups (one tile, two waves per SIMD) is worth building.  usage: python tools/occupancy_bound.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402

DEV = "cuda"
SHAPES = [("L1 ff-out", 10240, 640, 2560, 5), ("L2 ff-out", 2560, 1280, 5120, 5), ("L1 CxC", 10240, 640, 640, 5),
          ("L2 CxC", 2560, 1280, 1280, 5), ("L3 CxC", 640, 1280, 1280, 5), ("L3 ff-out", 640, 1280, 5120, 5), ("L1 ffz", 10240, 640, 3200, 5), ("L2 ffz", 2560, 1280, 6400, 5)]


def main():
    hip.load()
    for name, M, N, K, epi in SHAPES:
        line = f"{name:10s} N={N} K={K}:"
        for v in (-1, 9, 1):
            for mult in (1, 2):
                Mx = M * mult
                A = torch.randn(Mx, K, device=DEV).half()
                W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
                bias = torch.randn(N, device=DEV)
                res = torch.randn(Mx, N, device=DEV).half()
                out = torch.empty(Mx, N, device=DEV, dtype=torch.float16)
                hip.set_igemm_variant(v)
                d = hip.GemmDesc(Mx, N, K, K, N, N, epi, 1, 0, 1.0, 1 if v >= 0 else 0)
                wsb = hip.gemm_workspace_bytes(d)
                ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
                fn = lambda: hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(),
                                      ws.data_ptr(), ws.numel())
                med, _ = timeit(fn, 5)
                line += f"  v{v} M={Mx}: {med:6.1f}us"
        print(line, flush=True)
    hip.set_igemm_variant(-1)


if __name__ == "__main__":
    main()
