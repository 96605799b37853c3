#!/usr/bin/env python
"""Sweep tile variant x split-K factor for every distinct GEMM / conv3x3 shape of the bench workload's launch plan and
report where the library's shape heuristic (pick_variant / plan_splits, igemm.hip) is not the fastest choice.
Every candidate is timed back to back with HIP events on the plan's own shapes (synthetic operands); the in-graph times
are 5-15 % higher (operands HBM-cold) but rank the same.   usage: python tools/autotune.py [--latent 64] [--min-gain 0.05]"""
import argparse
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402

DEV = "cuda"


def shapes_of_plan(latent):
    import bench
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    model = bench.build_model(torch.device(DEV, 0))
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    story = synth.synthetic_story(stories=1, latent_hw=(latent, latent), ctx_len=85, seed=42)
    loop = DenoiseLoop(model, 1, 5, latent, latent, 85, 2.0, sched, 4)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    count = {}
    for tag in loop.prog.plan.tags:
        if tag.startswith(("gemm ", "conv3x3 ")):
            count[tag] = count.get(tag, 0) + 1
    del loop, model
    torch.cuda.empty_cache()
    return count


def bench_gemm(tag, variants, splits):
    kv = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", tag)}
    M, N, K, epi = kv["M"], kv["N"], kv["K"], kv["epi"]
    A = torch.randn(M, K, device=DEV).half()
    W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    bias = torch.randn(N, device=DEV)
    nout = N // 2 if epi & 8 else N
    res = torch.randn(M, nout, device=DEV).half()
    rowvec = torch.randn(16, nout, device=DEV)
    out = torch.empty(M, nout, device=DEV, dtype=torch.float16)
    rows = {}
    for v in variants:
        for sp in splits:
            hip.set_igemm_variant(v)
            d = hip.GemmDesc(M, N, K, K, nout, nout, epi, max(M // 2, 1), nout, 1.0, sp)
            wsb = hip.gemm_workspace_bytes(d)
            if wsb > (1 << 30):
                continue
            ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=DEV)
            try:
                fn = lambda: hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), rowvec.data_ptr(), res.data_ptr(),
                                      out.data_ptr(), ws.data_ptr(), ws.numel())
                rows[(v, sp)] = timeit(fn, 3, 8)[0]
            except hip.RcdmError:
                pass
    hip.set_igemm_variant(-1)
    return rows


def bench_conv(tag, variants, splits):
    n, H, W_ = (int(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
    cin, cout = (int(v) for v in re.search(r"(\d+)->(\d+)", tag).groups())
    kv = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", tag)}
    s, up, epi = kv["s"], kv["up"], kv["epi"]
    rows_out = n * H * W_ * (4 if up else 1) // (s * s)
    x = torch.randn(n * H * W_, cin, device=DEV).half()
    w = (torch.randn(cout, (16 if up == 2 else 9) * cin, device=DEV) * (9 * cin) ** -0.5).half()   # (up = 2: the four-phase image, [4][cout][4 cin])
    bias = torch.randn(cout, device=DEV)
    res = torch.randn(rows_out, cout, device=DEV).half()
    rowvec = torch.randn(16, cout, device=DEV)
    out = torch.empty(rows_out, cout, device=DEV, dtype=torch.float16)
    rows = {}
    for v in variants:
        for sp in splits:
            hip.set_igemm_variant(v)
            d = hip.ConvDesc(n, H, W_, cin, cout, s, up, cin, cout, cout, epi, max(rows_out // 2, 1), cout, 1.0, sp)
            wsb = hip.conv3x3_workspace_bytes(d)
            if wsb > (1 << 30):
                continue
            ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device=DEV)
            try:
                fn = lambda: hip.conv3x3(d, x.data_ptr(), w.data_ptr(), bias.data_ptr(), rowvec.data_ptr(), res.data_ptr(),
                                         out.data_ptr(), ws.data_ptr(), ws.numel())
                rows[(v, sp)] = timeit(fn, 3, 8)[0]
            except hip.RcdmError:
                pass
    hip.set_igemm_variant(-1)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--min-gain", type=float, default=0.05)
    a = ap.parse_args()
    count = shapes_of_plan(a.latent)
    variants = [-1, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    splits = [0, 1, 2, 3, 4, 6, 8]
    total_gain = 0.0
    for tag, n in sorted(count.items()):
        rows = (bench_gemm if tag.startswith("gemm") else bench_conv)(tag, variants, splits)
        if (-1, 0) not in rows:
            continue
        # the heuristic's own choice is timed first AND last (the first candidate of a new shape runs on cold clocks /
        # a cold TLB and reads 10-40 % slow): the later, lower number is the baseline
        again = (bench_gemm if tag.startswith("gemm") else bench_conv)(tag, [-1], [0])
        base = min(rows[(-1, 0)], again.get((-1, 0), 1e30))
        rows[(-1, 0)] = base
        (bv, bs), best = min(rows.items(), key=lambda kv: kv[1])
        gain = (base - best) * n
        flag = "  <-- " if best < base * (1 - a.min_gain) else ""
        if flag:
            total_gain += gain
        print(f"{tag:58s} n={n:3d} auto {base:7.1f} us | best v{bv} split {bs}: {best:7.1f} us  ({gain:7.1f} us per step){flag}", flush=True)
    print(f"sum of the flagged gains: {total_gain / 1e3:.3f} ms per step (back-to-back timing)")


if __name__ == "__main__":
    main()
