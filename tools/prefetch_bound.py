#!/usr/bin/env python
"""Bound of fetching the next launch's weights during the current one: every GEMM / conv of the step is timed inside the
step's own launch sequence twice — as is (its weights last read a whole step ago: HBM-cold), and right after a kernel that
reads exactly its weight image (so they sit in the memory-side cache when the op starts).  The difference, summed over the
step, is what a perfect weight prefetch could buy.   usage: python tools/prefetch_bound.py [--passes 5]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=5)
    a = ap.parse_args()
    import bench
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    story = synth.synthetic_story(stories=1, latent_hw=(64, 64), ctx_len=85, seed=42)
    loop = DenoiseLoop(model, 1, 5, 64, 64, 85, 2.0, sched, 4)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    loop.run(use_graph=False)
    plan = loop.prog.plan
    sink = torch.zeros(1, device=dev)

    def run(touch):
        times = {i: [] for i in plan.op_weights}
        for p in range(a.passes + 1):
            evs = {}
            for i, op in enumerate(plan.ops):
                w = plan.op_weights.get(i)
                if w is None:
                    op()
                    continue
                if touch:
                    sink.add_(w.view(torch.int16)[::32].sum())   # one element per 64-byte line: the whole image is read
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                op()
                e1.record()
                evs[i] = (e0, e1)
            torch.cuda.synchronize()
            if p:
                for i, (e0, e1) in evs.items():
                    times[i].append(e0.elapsed_time(e1) * 1e3)
        return {i: statistics.median(t) for i, t in times.items()}

    with torch.cuda.stream(loop.prog.stream):
        cold = run(False)
        warm = run(True)
        cold2 = run(False)
    agg = {}
    for i in cold:
        r = agg.setdefault(plan.tags[i], [0, 0.0, 0.0, 0.0, 0])
        r[0] += 1
        r[1] += cold[i]
        r[2] += warm[i]
        r[3] += cold2[i]
        r[4] = plan.op_weights[i].numel() * 2
    tc, tw, tc2 = sum(cold.values()), sum(warm.values()), sum(cold2.values())
    print(f"{len(cold)} GEMM / conv ops per step: in sequence {tc / 1e3:.3f} ms (repeat {tc2 / 1e3:.3f}), with their weights read just before {tw / 1e3:.3f} ms"
          f" -> bound of a perfect weight prefetch {(min(tc, tc2) - tw) / 1e3:.3f} ms per step")
    print(f"{'op':62s} {'n':>3s} {'MB':>6s} {'cold us':>8s} {'warm us':>8s} {'gain/step':>10s}")
    for tag, (n, c, w, c2, nb) in sorted(agg.items(), key=lambda kv: -(min(kv[1][1], kv[1][3]) - kv[1][2]))[:40]:
        print(f"{tag:62s} {n:3d} {nb / 1e6:6.1f} {min(c, c2) / n:8.1f} {w / n:8.1f} {min(c, c2) - w:9.1f}us")


if __name__ == "__main__":
    main()
