#!/usr/bin/env python
"""Probe: the 16x16 / 8x8 levels of the b = 2 step (down block 2 .. up block 1: contiguous in the launch plan) as ONE
b = 2 chain vs TWO concurrent b = 1 chains on two HIP streams.  The two CFG halves are independent samples through the
whole UNet (GroupNorm statistics are per sample), the deep levels' launches do not fill the chip and are latency chains, so
two chains could overlap each other's ramps and tails; tools/dual_stream_probe.py measured the WHOLE step that way
(neutral) — this one isolates the deep part and the hybrid (b = 2 shallow levels, forked deep levels).
Timing only: the sub-graphs run on whatever the buffers hold after one full forward.
usage: python tools/deep_fork_probe.py [--iters 30] [--levels 16,8]"""
import argparse
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def op_rows(tag):
    kv = dict(re.findall(r"(\w+)=(-?\d+)", tag))
    kind = tag.split()[0]
    try:
        if kind in ("gemm", "gemm_ln", "rowchain", "layernorm", "ff_fused"):
            return int(kv["M"])
        if kind == "conv3x3":
            n, H, W = map(int, re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
            return n * H * W * (4 if int(kv.get("up", 0)) else 1) // int(kv.get("s", 1)) ** 2
        if kind in ("flash_attn", "xattn"):
            return int(kv["B"]) * int(kv["Lq"])
        if kind == "temporal_attn":
            return int(kv["S"]) * int(kv["F"]) * int(kv["P"])
        if kind in ("groupnorm", "groupnorm_stats"):
            return int(kv["S"]) * int(kv["R"])
    except (KeyError, AttributeError):
        pass
    return None


def deep_range(prog, max_rows):
    """[i0, i1): the contiguous run of body ops whose row count is <= max_rows (ops without a row count inherit)."""
    tags = prog.plan.tags
    idx = [i for i in range(prog.n_time_ops, len(tags)) if (op_rows(tags[i]) or 1 << 30) <= max_rows]
    i0, i1 = idx[0], idx[-1] + 1
    inside = [i for i in range(i0, i1) if (op_rows(tags[i]) or 0) > max_rows]
    return i0, i1, inside


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--deep", type=int, default=16, help="deepest-level side length from which the fork starts (16: 16^2 + 8^2)")
    a = ap.parse_args()
    import bench
    from rcdms_amd import engine, hip, synth
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
        torch.cuda.synchronize()
        return p

    def sub_graph(p, ops):
        torch.cuda.synchronize()
        with torch.cuda.stream(p.stream):
            g = hip.Graph()
            g.begin()
            try:
                for op in ops:
                    op()
            finally:
                g.end()
        torch.cuda.synchronize()
        return g

    with torch.no_grad():
        p2, pa, pb = make(2, [0, 1]), make(1, [0]), make(1, [1])
    side = a.deep
    progs = {}
    for name, p, b in (("b2", p2, 2), ("a", pa, 1), ("b", pb, 1)):
        i0, i1, inside = deep_range(p, b * 5 * side * side)
        ops = p.plan.ops
        progs[name] = dict(p=p, i0=i0, i1=i1,
                           pre=sub_graph(p, ops[p.n_time_ops:i0]), deep=sub_graph(p, ops[i0:i1]), post=sub_graph(p, ops[i1:]),
                           full=sub_graph(p, ops[p.n_time_ops:]))
        print(f"{name}: body ops {p.n_time_ops}..{len(ops)}, deep range [{i0}, {i1}) = {i1 - i0} ops "
              f"({len(inside)} of them above the row bound: {[p.plan.tags[i] for i in inside][:4]})", flush=True)
        print(f"   first deep op: {p.plan.tags[i0]} | last: {p.plan.tags[i1 - 1]}", flush=True)

    s2, sa, sb = p2.stream, pa.stream, pb.stream

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def on(stream, g):
        with torch.cuda.stream(stream):
            g.launch()

    def full2():
        on(s2, progs["b2"]["full"])

    def parts2():
        on(s2, progs["b2"]["pre"]); on(s2, progs["b2"]["deep"]); on(s2, progs["b2"]["post"])

    def deep2():
        on(s2, progs["b2"]["deep"])

    def deep_ab():
        on(sa, progs["a"]["deep"]); on(sb, progs["b"]["deep"])

    def deep_ab_serial():
        on(sa, progs["a"]["deep"]); on(sa, progs["b"]["deep"])

    def shallow2():
        on(s2, progs["b2"]["pre"]); on(s2, progs["b2"]["post"])

    def shallow_ab():
        on(sa, progs["a"]["pre"]); on(sb, progs["b"]["pre"]); on(sa, progs["a"]["post"]); on(sb, progs["b"]["post"])

    ev = [torch.cuda.Event() for _ in range(3)]

    def hybrid():
        on(s2, progs["b2"]["pre"])
        ev[0].record(s2)
        sa.wait_event(ev[0]); sb.wait_event(ev[0])
        on(sa, progs["a"]["deep"]); on(sb, progs["b"]["deep"])
        ev[1].record(sa); ev[2].record(sb)
        s2.wait_event(ev[1]); s2.wait_event(ev[2])
        on(s2, progs["b2"]["post"])

    for r in range(3):
        print(f"round {r}: full b=2 graph {timed(full2, a.iters):.3f} | b=2 as pre+deep+post {timed(parts2, a.iters):.3f} | "
              f"hybrid (b=2 pre, forked deep, b=2 post) {timed(hybrid, a.iters):.3f} ms", flush=True)
        print(f"         deep: b=2 {timed(deep2, a.iters):.3f} | two b=1 concurrent {timed(deep_ab, a.iters):.3f} | "
              f"two b=1 serial {timed(deep_ab_serial, a.iters):.3f} ms", flush=True)
        print(f"         shallow: b=2 {timed(shallow2, a.iters):.3f} | two b=1 concurrent {timed(shallow_ab, a.iters):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
