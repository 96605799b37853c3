#!/usr/bin/env python
"""VAE decode and encode of one 5-frame story at 512x512 (SD-1.5 AutoencoderKL, random-init) on the HIP path: ms per story."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rcdms_amd import vae
dev = torch.device("cuda", 0)
with torch.device("meta"):
    m = vae.AutoencoderKL()
m = m.to_empty(device=dev).eval()
bench.init_weights_(m)
z = torch.randn(5, 4, 64, 64, device=dev)
for _ in range(2):
    y = m.decode(z).sample
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    y = m.decode(z).sample
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"vae decode 5 x 512x512: {1e3 * dt:.2f} ms per story, output {tuple(y.shape)}, finite {bool(torch.isfinite(y).all())}, "
      f"6.2 TFLOP -> {6.2 / dt:.0f} TFLOP/s")

x = torch.rand(5, 3, 512, 512, device=dev) * 2 - 1
for _ in range(2):
    d = m.encode(x).latent_dist
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    d = m.encode(x).latent_dist
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"vae encode 5 x 512x512: {1e3 * dt:.2f} ms per story, mean {tuple(d.mean.shape)}, finite {bool(torch.isfinite(d.mean).all())}")
from collections import Counter
prog = m._enc_programs[(5, 512, 512)][1]
print("encode plan:", len(prog.plan.ops), "launches")
