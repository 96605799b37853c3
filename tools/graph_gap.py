#!/usr/bin/env python
"""Cost of a dependent kernel node in a replayed hipGraph: N launches of a one-thread kernel (rcdm_advance_step) and of
a small real kernel (LayerNorm of 640 x 1280), captured once, replayed; microseconds per node."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip
hip.load()
dev = torch.device("cuda", 0)
step = torch.zeros(1, dtype=torch.int32, device=dev)
x = torch.randn(640, 1280, device=dev).half(); y = torch.empty_like(x)
g_, b_ = torch.ones(1280, device=dev), torch.zeros(1280, device=dev)
s = torch.cuda.Stream(device=dev)
for name, fn, n in (("1-thread kernel", lambda: hip.advance_step(step.data_ptr()), 1000),
                    ("layernorm 640x1280", None, 1000)):
    with torch.cuda.stream(s):
        if fn is None:
            d = hip.LayerNormDesc(640, 1280, 1280, 1280, 1e-5, 640, 1)
            fn = lambda: hip.layernorm(d, x.data_ptr(), g_.data_ptr(), b_.data_ptr(), 0, y.data_ptr())
        fn(); s.synchronize()
        g = hip.Graph(); g.begin()
        for _ in range(n):
            fn()
        g.end()
        for _ in range(3):
            g.launch()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.launch()
        s.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name}: {n} nodes replay {dt * 1e3:.3f} ms -> {dt / n * 1e6:.2f} us per node")
