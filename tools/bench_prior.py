#!/usr/bin/env python
"""BASELINE config 5: stage-1 frame-prior transformer diffusion, 50-step UnCLIP sampling of one 5-frame story with CFG
(batch 2 x 5 = 10 sequences of 97 tokens) on one MI355X.  Random-init weights of the real 2.85 G-parameter shape,
synthetic conditioning.  Prints one JSON line: stories/s, ms per denoising step, and the roofline of the step.  Bound:
MFMA — 5.5 TFLOP of GEMM per step (970 token rows x 2.88 G weights x 2) against 5.8 GB of f16 weights, i.e. 2.2 ms at the
2.5 PFLOP/s peak vs 0.7 ms of weight streaming at 8 TB/s; achieved = algorithmic GEMM + attention flops / step time.
usage: python tools/bench_prior.py [--steps 3] [--warmup 1] [--sample-steps 50] [--layers 20]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sample-steps", type=int, default=50)
    ap.add_argument("--layers", type=int, default=20)
    ap.add_argument("--guidance", type=float, default=4.0)
    a = ap.parse_args()
    import __graft_entry__
    __graft_entry__.build()
    from rcdms_amd import hip
    from rcdms_amd.sampler import PriorLoop
    from rcdms_amd.scheduler import UnCLIPScheduler
    from src.models.myprior_transformer import MyPriorTransformer
    dev = torch.device("cuda", 0)
    mk = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
              temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)
    with torch.device("meta"):
        m = MyPriorTransformer(num_attention_heads=32, attention_head_dim=64, num_layers=a.layers, embedding_dim=1280,
                               num_embeddings=91, additional_embeddings=6, unet_use_cross_frame_attention=False,
                               unet_use_temporal_attention=False, use_motion_module=True, motion_module_type="Vanilla",
                               motion_module_kwargs=mk)
    m = m.to_empty(device=dev).eval()
    import bench
    bench.init_weights_(m)
    nparam = sum(p.numel() for p in m.parameters())
    reps = 2 if a.guidance > 1 else 1
    B, T, E = 5 * reps, 91, 1280
    g = torch.Generator(device=dev).manual_seed(42)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    mask = torch.ones(B, T, device=dev)
    mask[:, 20:] = 0
    loop = PriorLoop(m, 5, T, a.guidance, UnCLIPScheduler(), a.sample_steps)
    args = (rn(B, E), rn(B, T, E), rn(B, E), rn(B, E), mask)
    lat = rn(5, E)

    def one():
        loop.load(lat, *args, generator=g)
        loop.run()
    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    ev0, ev1 = hip.Event(), hip.Event()
    gpu_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loop.load(lat, *args, generator=g)
        sp = loop.stream.cuda_stream
        ev0.record(sp)
        loop.run()
        ev1.record(sp)
        gpu_ms += ev0.elapsed_ms(ev1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_step = gpu_ms / (a.steps * a.sample_steps)
    import re
    flops = 0.0
    for tag in loop.prog.plan.tags:      # algorithmic work of one step, from the launch plan's own descriptors
        g_ = re.match(r"gemm M=(\d+) N=(\d+) K=(\d+)", tag)
        f_ = re.match(r"flash_attn_masked B=(\d+) H=(\d+) L=(\d+) d=(\d+)", tag)
        if g_:
            flops += 2.0 * int(g_.group(1)) * int(g_.group(2)) * int(g_.group(3))
        elif f_:
            bb, hh, ll, dd = (int(v) for v in f_.groups())
            flops += 4.0 * bb * hh * ll * ll * dd
    print(json.dumps({
        "metric": "stories/sec (stage-1 prior, 50-step UnCLIP, 5 frames, CFG)", "value": round(a.steps / dt, 4),
        "unit": "stories/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"stage-1 prior transformer, {a.layers} layers x (block + motion module), {nparam / 1e9:.2f} G "
                               f"parameters, batch {B} x 97 tokens, {a.sample_steps}-step UnCLIP, CFG {a.guidance}"},
        "roofline": {"bound": "mfma", "kernel": "denoise-step graph (prior forward + CFG + UnCLIP step)",
                     "achieved": round(flops / (ms_step * 1e-3) / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": round(flops / (ms_step * 1e-3) / 2.5e15, 4), "traffic": None,
                     "tflop_per_step": round(flops / 1e12, 3), "weight_gb": round(2.0 * nparam / 1e9, 2),
                     "avg_launch_ms": round(ms_step, 4), "launches": a.steps * a.sample_steps}}), flush=True)


if __name__ == "__main__":
    main()
