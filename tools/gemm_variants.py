#!/usr/bin/env python
"""rcdm_gemm at one shape on every tile family (forced variant), HIP-event timed over a rotation of cold operand sets.
usage: python tools/gemm_variants.py M N K [--variants 1,2,6,7,8,9] [--split 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("M", type=int)
    ap.add_argument("N", type=int)
    ap.add_argument("K", type=int)
    ap.add_argument("--variants", default="1,2,6,7,8,9")
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--sets", type=int, default=6)
    a = ap.parse_args()
    hip.load()
    sets = [(torch.randn(a.M, a.K, device="cuda").half(), (torch.randn(a.N, a.K, device="cuda") * a.K ** -0.5).half())
            for _ in range(a.sets)]
    out = torch.empty(a.M, a.N, dtype=torch.float16, device="cuda")
    for v in [int(x) for x in a.variants.split(",")]:
        hip.set_igemm_variant(v)
        d = hip.GemmDesc(a.M, a.N, a.K, a.K, a.N, 0, 0, 1, 0, 1.0, a.split, 0)
        ws = torch.empty(max(hip.gemm_workspace_bytes(d), 16), dtype=torch.uint8, device="cuda")

        def run(i):
            A, W = sets[i % a.sets]
            hip.gemm(d, A.data_ptr(), W.data_ptr(), 0, 0, 0, out.data_ptr(), ws.data_ptr(), ws.numel())
        for i in range(40):
            run(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12 * 1e3)
        ts.sort()
        print(f"variant {v:2d}: {ts[3]:7.1f} us  ({2e-6 * a.M * a.N * a.K / ts[3]:.0f} TFLOP/s)  plan {hip.gemm_plan_query(d)}", flush=True)
    hip.set_igemm_variant(-1)


if __name__ == "__main__":
    main()
