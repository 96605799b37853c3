#!/usr/bin/env python
"""VAE decode of one 5-frame story at 512x512 (SD-1.5 AutoencoderKL decoder, random-init) on the HIP path: ms per story."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rcdms_amd import vae
dev = torch.device("cuda", 0)
with torch.device("meta"):
    m = vae.AutoencoderKLDecoder()
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
