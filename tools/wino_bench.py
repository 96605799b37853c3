#!/usr/bin/env python
"""Winograd F(2x2, 3x3) (rcdm_conv3x3_wino) against the nine-tap implicit GEMM (rcdm_conv3x3 / rcdm_conv3x3_add1x1) on the
stride-1 3x3 convolutions of the UNet's 16x16 / 8x8 levels (b f = 10 images): HIP-event timed, the two forms interleaved,
a rotation of distinct operand sets so that weights are HBM-cold as in the step graph.
usage: python tools/wino_bench.py [--split N] [--sets 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402

DEV = "cuda"
SHAPES = [  # (name, n_img, H, W, cin, cin2, cout, epi)
    ("16^2 1280->1280 conv1", 10, 16, 16, 1280, 0, 1280, 3),
    ("16^2 1280->1280 conv2+res", 10, 16, 16, 1280, 0, 1280, 5),
    ("16^2 2560->1280 conv1", 10, 16, 16, 2560, 0, 1280, 3),
    ("16^2 1920->1280 conv1", 10, 16, 16, 1920, 0, 1280, 3),
    ("16^2 640->1280 conv1", 10, 16, 16, 640, 0, 1280, 3),
    ("16^2 1280->1280 +1x1(2560)", 10, 16, 16, 1280, 2560, 1280, 1),
    ("16^2 1280->1280 +1x1(640)", 10, 16, 16, 1280, 640, 1280, 1),
    ("8^2 1280->1280 conv1", 10, 8, 8, 1280, 0, 1280, 3),
    ("8^2 2560->1280 conv1", 10, 8, 8, 2560, 0, 1280, 3),
    ("8^2 1280->1280 +1x1(2560)", 10, 8, 8, 1280, 2560, 1280, 1),
    ("32^2 640->640 conv1", 10, 32, 32, 640, 0, 640, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--sets", type=int, default=6)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--shape", type=int, default=-1, help="index into SHAPES (default: all)")
    a = ap.parse_args()
    hip.load()
    for name, n_img, H, W, cin, cin2, cout, epi in (SHAPES if a.shape < 0 else [SHAPES[a.shape]]):
        M = n_img * H * W
        sets = []
        for s in range(a.sets):
            x = torch.randn(M, cin, device=DEV).half()
            x2 = torch.randn(M, cin2, device=DEV).half() if cin2 else None
            w = torch.randn(cout, cin, 3, 3, device=DEV) * (9 * cin) ** -0.5
            U = torch.empty(16, cout, cin, dtype=torch.float16, device=DEV)
            hip.pack_conv3x3_wino(w.data_ptr(), cout, cin, U.data_ptr())
            wp = torch.empty(cout, 9 * cin, dtype=torch.float16, device=DEV)
            hip.pack_conv3x3(w.data_ptr(), cout, cin, cin, wp.data_ptr())
            w2 = (torch.randn(cout, cin2, device=DEV) * cin2 ** -0.5).half() if cin2 else None
            wk = torch.cat([wp, w2], dim=1).contiguous() if cin2 else wp
            sets.append((x, x2, U, w2, wk))
        bias = torch.randn(cout, device=DEV)
        temb = torch.randn(2, cout, device=DEV)
        res = torch.randn(M, cout, device=DEV).half()
        out = torch.empty(M, cout, dtype=torch.float16, device=DEV)
        d = hip.ConvDesc(n_img, H, W, cin, cout, 1, 0, cin, cout, cout if epi & 4 else 0, epi, M // 2, cout, 1.0, a.split, 0, 0, cin2,
                         cin2)
        d0 = hip.ConvDesc(n_img, H, W, cin, cout, 1, 0, cin, cout, cout if epi & 4 else 0, epi, M // 2, cout, 1.0, 0, 0, 0, cin2, cin2)
        wsw = torch.empty(max(hip.conv3x3_wino_workspace_bytes(d), 16), dtype=torch.uint8, device=DEV)
        wsd = torch.empty(max(hip.conv3x3_workspace_bytes(d0), 16), dtype=torch.uint8, device=DEV)
        bp, tp, rp = bias.data_ptr() if epi & 1 else 0, temb.data_ptr() if epi & 2 else 0, res.data_ptr() if epi & 4 else 0

        def wino(i):
            x, x2, U, w2, wk = sets[i % a.sets]
            hip.conv3x3_wino(d, x.data_ptr(), U.data_ptr(), bp, tp, rp, out.data_ptr(), wsw.data_ptr(), wsw.numel(),
                             x2=x2.data_ptr() if cin2 else 0, W2=w2.data_ptr() if cin2 else 0)

        def direct(i):
            x, x2, U, w2, wk = sets[i % a.sets]
            if cin2:
                hip.conv3x3_add1x1(d0, x.data_ptr(), x2.data_ptr(), wk.data_ptr(), bp, tp, rp, out.data_ptr(), wsd.data_ptr(), wsd.numel())
            else:
                hip.conv3x3(d0, x.data_ptr(), wk.data_ptr(), bp, tp, rp, out.data_ptr(), wsd.data_ptr(), wsd.numel())

        for i in range(30):
            wino(i)
            direct(i)
        torch.cuda.synchronize()
        tw, td = [], []
        for r in range(a.rounds):
            for fn, acc in ((wino, tw), (direct, td)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(2 * a.sets):
                    fn(i)
                e1.record()
                torch.cuda.synchronize()
                acc.append(e0.elapsed_time(e1) / (2 * a.sets) * 1e3)
        tw.sort()
        td.sort()
        pq = hip.conv3x3_wino_plan_query(d)
        print(f"{name:30s} winograd {tw[len(tw) // 2]:7.1f} us (3 launches, split {pq[5]})   nine-tap {td[len(td) // 2]:7.1f} us"
              f"   ratio {tw[len(tw) // 2] / td[len(td) // 2]:.2f}", flush=True)


if __name__ == "__main__":
    main()
