#!/usr/bin/env python
"""Per-op timing of one UNet call at the bench workload, aggregated by (kind, shape): every op of the launch plan is
run alone `--iters` times between HIP events (eager, same buffers), so the numbers include launch gaps the graph hides;
use them for RELATIVE weight per shape.  Next to every op INSTANCE: its algorithmic work per step (GFLOP for the
contractions — 2 MAC, the convention of SURVEY §8(d) — or MB for the HBM-bound norms / temporal attention), and the rate
that follows, so that every per-kernel fraction quoted in DESIGN.md can be recomputed from the committed table.
usage: python tools/opprof.py [--latent 64] [--iters 5] [--top 60]"""
import re
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def algorithmic_work(tag):
    """(GFLOP, MB) of ONE launch of the op named by a plan tag (engine.py emitters); None where not applicable."""
    kv = {k: float(v) for k, v in re.findall(r"(\w+)=(-?[\d.]+)", tag)}
    kind = tag.split()[0]
    if kind == "gemm":
        return 2e-9 * kv["M"] * kv["N"] * kv["K"], None
    if kind == "gemm_ln":
        return 2e-9 * kv["M"] * kv["N"] * kv["K"], None
    if kind == "conv3x3":
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        cin, cout = (float(v) for v in re.search(r"(\d+)->(\d+)", tag).groups())
        rows = n * H * W * (4.0 if kv.get("up") else 1.0) / (kv.get("s", 1.0) ** 2)
        return 2e-9 * rows * (9 * cin + kv.get("add1x1", 0.0)) * cout, None   # (add1x1: the folded conv_shortcut's channels)
    if kind == "conv3x3_wino":   # priced at the reference's nine taps; the Winograd form multiplies 16 / 36 of them
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        cin, cout = (float(v) for v in re.search(r"(\d+)->(\d+)", tag).groups())
        return 2e-9 * n * H * W * (9 * cin + kv.get("add1x1", 0.0)) * cout, None
    if kind == "conv_gather":
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        return None, 2e-6 * n * H * W * kv["C"] * (9 + 1)
    if kind == "upsample_gather":   # (its products are priced with the N = 9 c GEMM in front of it)
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        return None, 2e-6 * n * H * W * kv["C"] * (9 + 4)
    if kind in ("flash_attn", "xattn", "flash_attn_masked"):
        Lk = kv.get("Lk", kv.get("L", 0))
        Lq = kv.get("Lq", kv.get("L", 0))
        return 4e-9 * kv["B"] * kv["H"] * Lq * Lk * kv["d"], None
    if kind == "ff_fused":
        return 2e-9 * kv["M"] * 12 * kv["C"] ** 2, 4e-6 * kv["M"] * kv["C"]
    if kind == "rowchain":
        t = kv["tail"]   # 0: feed-forward (12 C^2 per row), 2: feed-forward + projection (13), 1 / 3: GEMM to t*C columns
        nmat = {0: 12, 2: 13}.get(t, t)
        # rows moved: a_in, (res), tok written, then q / qkv out | tok re-read + out | tok re-read + z_res + out
        return 2e-9 * kv["M"] * kv["C"] ** 2 * (1 + nmat), 2e-6 * kv["M"] * kv["C"] * (2 + kv["res"] + {0: 2, 2: 3}.get(t, t))
    if kind == "temporal_attn":
        C = kv["H"] * kv["d"]
        rows = kv["S"] * kv["F"] * kv["P"]
        return 4e-9 * rows * kv["F"] * C, 8e-6 * rows * C
    if kind == "groupnorm":
        return None, 6e-6 * kv["S"] * kv["R"] * kv["C"]
    if kind == "groupnorm_stats":
        return None, 2e-6 * kv["S"] * kv["R"] * kv["C"]
    if kind == "layernorm":
        return None, 4e-6 * kv["M"] * kv["C"]
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--context", choices=("dense", "reference"), default="dense",
                    help="reference: the context-row structure of SURVEY F6 (the rank-1-context plan)")
    ap.add_argument("--list", action="store_true", help="also print the plan's ops in launch order with their times")
    a = ap.parse_args()
    import bench
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    dev = torch.device("cuda", 0)
    model = bench.build_model(dev)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    story = synth.synthetic_story(stories=1, latent_hw=(a.latent, a.latent), ctx_len=85, seed=42, structure=a.context)
    loop = DenoiseLoop(model, 1, 5, a.latent, a.latent, 85, 2.0, sched, 4)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    loop.run(use_graph=False)
    torch.cuda.synchronize()
    plan = loop.prog.plan
    agg = collections.defaultdict(lambda: [0, 0.0])
    seq = []
    with torch.cuda.stream(loop.prog.stream):
        for op, tag in zip(plan.ops, plan.tags):
            op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                op()
            e1.record()
            e1.synchronize()
            r = agg[tag]
            r[0] += 1
            r[1] += e0.elapsed_time(e1) / a.iters * 1e3
            seq.append((tag, e0.elapsed_time(e1) / a.iters * 1e3))
    tot = sum(v[1] for v in agg.values())
    gf_tot = sum(n * (algorithmic_work(tag)[0] or 0.0) for tag, (n, _) in agg.items())
    print(f"total {tot / 1e3:.3f} ms over {len(plan.ops)} ops; algorithmic work of the listed contractions {gf_tot / 1e3:.3f} TFLOP per step")
    print(f"{'ms/step':>8s} {'%':>6s} {'n':>5s} {'avg us':>9s} {'GFLOP/op':>9s} {'TFLOP/s':>8s} {'MB/op':>8s} {'TB/s':>6s}  op")
    for tag, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        gf, mb = algorithmic_work(tag)
        avg = us / n
        c1 = f"{gf:9.2f} {gf / avg * 1e3:8.0f}" if gf else f"{'':9s} {'':8s}"
        c2 = f"{mb:8.1f} {mb / avg:6.2f}" if mb else f"{'':8s} {'':6s}"
        print(f"{us / 1e3:8.3f} {100 * us / tot:5.1f}% {n:5d} {avg:9.1f} {c1} {c2}  {tag}")
    if a.list:
        print("--- launch order")
        for i, (tag, us) in enumerate(seq):
            print(f"{i:4d} {us:8.1f} us  {tag}")


if __name__ == "__main__":
    main()

