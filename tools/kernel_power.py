#!/usr/bin/env python
"""Which launches of the step are power-capped?  Every op kind of the bench workload's launch plan (the most expensive
instance per kind and level) is replayed alone in a tight loop for ~0.6 s while a thread samples the shader clock and the
socket power (rocm-smi); printed per op: time per launch, the clock and power it sustains.  The whole step sustains
~2.13-2.25 GHz at ~1145-1190 W (tools/clock_probe.sh); this table says which kernels pull the clock down and which run
at 2.4 GHz because they leave the chip idle.  Results are the ops' own (right) results on real data: clocks depend on the
operand values.
usage: python tools/kernel_power.py [--latent 64] [--seconds 0.6] [--top 24]"""
import argparse
import collections
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def level(tag):
    """Resolution level of a plan tag (rows of the b = 2, f = 5 step), as tools/level_summary.py."""
    kv = dict(re.findall(r"(\w+)=(-?\d+)", tag))
    kind = tag.split()[0]
    try:
        if kind in ("gemm", "gemm_ln", "rowchain", "layernorm", "ff_fused"):
            M = int(kv["M"])
        elif kind == "conv3x3":
            n, H, W = map(int, re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
            M = n * H * W * (4 if int(kv.get("up", 0)) else 1) // int(kv.get("s", 1)) ** 2
        elif kind in ("flash_attn", "xattn"):
            M = int(kv["B"]) * int(kv["Lq"])
        elif kind == "temporal_attn":
            M = int(kv["S"]) * int(kv["F"]) * int(kv["P"])
        elif kind in ("groupnorm", "groupnorm_stats"):
            M = int(kv["S"]) * int(kv["R"])
        else:
            return "other"
    except (KeyError, AttributeError):
        return "other"
    for name, rows in (("64^2", 40960), ("32^2", 10240), ("16^2", 2560), ("8^2", 640)):
        if M >= rows * 0.45:
            return name
    return "tiny"


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.on, self.stop_ = [], False, False

    def run(self):
        while not self.stop_:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                continue
            m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            m2 = re.search(r"Power \(W\): ([\d.]+)", out)
            if self.on and m1 and m2:
                self.rows.append((int(m1.group(1)), float(m2.group(1))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=0.6)
    ap.add_argument("--top", type=int, default=24)
    a = ap.parse_args()
    import bench
    from opprof import algorithmic_work
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    story = synth.synthetic_story(stories=1, latent_hw=(a.latent, a.latent), ctx_len=85, seed=42)
    loop = DenoiseLoop(model, 1, 5, a.latent, a.latent, 85, 2.0, sched, 4)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    loop.run(use_graph=False)
    torch.cuda.synchronize()
    plan = loop.prog.plan
    # one representative per (kind, level): the instance with the longest single-launch time
    quick = {}
    with torch.cuda.stream(loop.prog.stream):
        for i, (op, tag) in enumerate(zip(plan.ops, plan.tags)):
            op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                op()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) / 3 * 1e3
            key = (tag.split()[0], level(tag))
            if key not in quick or us > quick[key][0]:
                quick[key] = (us, i)
    picks = sorted(quick.items(), key=lambda kv: -kv[1][0])[:a.top]
    smp = Sampler()
    smp.start()
    print(f"{'us/launch':>10s} {'TFLOP/s':>8s} {'sclk MHz':>9s} {'power W':>8s} {'samples':>7s}  op")
    # the whole step first (graph replays), as the reference line
    g = loop.prog.capture()

    def measure(fn, label, gf=None):
        with torch.cuda.stream(loop.prog.stream):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            smp.rows, n, t0 = [], 0, time.perf_counter()
            smp.on = True
            while time.perf_counter() - t0 < a.seconds or len(smp.rows) < 3:
                for _ in range(20):
                    fn()
                n += 20
                torch.cuda.synchronize()
                if time.perf_counter() - t0 > 6 * a.seconds:
                    break
            dt = time.perf_counter() - t0
            smp.on = False
        rows = smp.rows[1:] or smp.rows   # (the first sample may straddle the start)
        clk = sum(r[0] for r in rows) / max(len(rows), 1)
        pw = sum(r[1] for r in rows) / max(len(rows), 1)
        us = dt / n * 1e6
        tf = f"{gf / us * 1e3:8.0f}" if gf else f"{'':8s}"
        print(f"{us:10.1f} {tf} {clk:9.0f} {pw:8.0f} {len(rows):7d}  {label}", flush=True)

    measure(g.launch, "WHOLE STEP (graph replay)", 11044.0 - 585.5)
    for (kind, lv), (us, i) in picks:
        gf, _ = algorithmic_work(plan.tags[i])
        measure(plan.ops[i], plan.tags[i], gf)
    smp.stop_ = True


if __name__ == "__main__":
    main()
