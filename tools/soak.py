#!/usr/bin/env python
"""Determinism soak: the full-width 50-step story N times from the same inputs; every run must reproduce the first
bit for bit (a missed wait or an LDS race in any kernel of the step graph shows up as a differing latent sooner or later).
usage: python tools/soak.py [runs]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rcdms_amd import synth  # noqa: E402
from rcdms_amd.sampler import DenoiseLoop  # noqa: E402
from rcdms_amd.scheduler import DDIMScheduler  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
story = synth.synthetic_story(stories=1, latent_hw=(64, 64), ctx_len=85, seed=42)
loop = DenoiseLoop(model, 1, 5, 64, 64, 85, 2.0, sched, 50)
digests = []
for i in range(runs):
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    lat = loop.run().float().cpu()
    assert torch.isfinite(lat).all()
    digests.append(hashlib.sha256(lat.numpy().tobytes()).hexdigest()[:16])
    print(i, digests[-1], flush=True)
print("identical" if len(set(digests)) == 1 else f"MISMATCH: {len(set(digests))} different results in {runs} runs")
