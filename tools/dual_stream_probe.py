#!/usr/bin/env python
"""Probe: is ONE b=2 UNet step graph faster or slower than TWO concurrent b=1 step graphs on two HIP streams?
(The two CFG halves of a step are independent after the shared prefix; at the 16x16 / 8x8 levels single kernels do not
fill the chip and dependent launches are latency-bound, so two chains could overlap each other's gaps.)
usage: python tools/dual_stream_probe.py [--latent 64] [--iters 20]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import bench
    from rcdms_amd import engine, synth
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    hw = a.latent
    st = synth.synthetic_story(stories=1, latent_hw=(hw, hw), ctx_len=85, seed=42)
    x = torch.cat([torch.cat([st["latents"]] * 2), st["mask"], st["masked_latents"]], dim=1).to(dev)
    ctx = st["ctx"].to(dev)
    cfg, sd = model.engine_config(), model.state_dict()

    def make(b, rows):
        p = engine.UNetProgram(cfg, sd, b, 5, hw, hw, 85, dev)
        p.forward(x[rows], 981, ctx.view(2, 5, 85, 768)[rows].reshape(-1, 85, 768), use_graph=False)
        g = p.capture()
        return p, g

    with torch.no_grad():
        p2, g2 = make(2, [0, 1])
        pa, ga = make(1, [0])
        pb, gb = make(1, [1])

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def one():
        with torch.cuda.stream(p2.stream):
            g2.launch()

    def two():
        with torch.cuda.stream(pa.stream):
            ga.launch()
        with torch.cuda.stream(pb.stream):
            gb.launch()

    def two_serial():
        with torch.cuda.stream(pa.stream):
            ga.launch()
            gb.launch()

    for r in range(3):
        print(f"round {r}: b=2 graph {timed(one, a.iters):.3f} ms | two b=1 graphs, two streams {timed(two, a.iters):.3f} ms | "
              f"two b=1 graphs, one stream {timed(two_serial, a.iters):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
