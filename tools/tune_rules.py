#!/usr/bin/env python
"""In-sequence tile tuning: every GEMM / conv3x3 of the bench workload's launch plan is timed INSIDE the step's launch
sequence (one execution per pass between two events, operands as cold as in the replayed graph — tools/autotune.py times
each shape back to back, operands hot, and mis-ranks the latency-bound shapes), once per candidate tile variant forced on
all shapes at a time through rcdm_set_shape_rules.  Prints, per shape, the library's choice against the best candidate
and a RCDM_SHAPE_RULES string of the gains; confirm with tools/ab_rules.sh before a rule goes into kShapeRules.
usage: python tools/tune_rules.py [--latent 64] [--stories 1] [--ctx-len 85] [--passes 5] [--min-gain-us 1.0]"""
import argparse
import os
import re
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402

# (variant, split-K factor; 0 = the variant's own heuristic) candidates per tap count
CAND = {1: [(v, 0) for v in (1, 2, 3, 4, 5, 9, 10)] + [(v, sp) for v in (1, 3, 5, 9, 10) for sp in (1, 2, 3, 4)],
        9: [(v, 0) for v in (1, 2, 5, 6, 7, 8, 9)] + [(v, sp) for v in (1, 2, 8, 9) for sp in (1, 2, 4, 8)]}


def shape_of(tag):
    if tag.startswith("gemm "):
        kv = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", tag)}
        return (1, kv["M"], kv["N"], kv["K"])
    if tag.startswith("conv3x3 "):
        n, H, W = (int(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        cin, cout = (int(v) for v in re.search(r"(\d+)->(\d+)", tag).groups())
        kv = {k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", tag)}
        if kv["up"] == 2:
            return None
        rows = n * H * W * (4 if kv["up"] else 1) // (kv["s"] * kv["s"])
        return (9, rows, cout, cin)
    return None


def time_passes(plan, passes):
    n = len(plan.ops)
    times = [[] for _ in range(n)]
    for p in range(passes + 1):
        evs = []
        for op in plan.ops:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            try:
                op()
            except hip.RcdmError:
                evs.append(None)
                continue
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        if p == 0:
            continue
        for i, ev in enumerate(evs):
            times[i].append(float("inf") if ev is None else ev[0].elapsed_time(ev[1]) * 1e3)
    return [statistics.median(t) for t in times]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--stories", type=int, default=1)
    ap.add_argument("--ctx-len", type=int, default=85)
    ap.add_argument("--prior", action="store_true", help="tune the stage-1 prior's step (config 5) instead of the UNet's")
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--min-gain-us", type=float, default=1.0)
    a = ap.parse_args()
    import bench
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    dev = torch.device("cuda", 0)
    if a.prior:
        plan, stream = prior_plan(dev)
        return tune(plan, stream, a)
    model = bench.build_model(dev)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    story = synth.synthetic_story(stories=a.stories, latent_hw=(a.latent, a.latent), ctx_len=a.ctx_len, seed=42)
    loop = DenoiseLoop(model, a.stories, 5, a.latent, a.latent, a.ctx_len, 2.0, sched, 4)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    loop.run(use_graph=False)
    tune(loop.prog.plan, loop.prog.stream, a)


def prior_plan(dev):
    """BASELINE config 5 (tools/bench_prior.py): the stage-1 prior's step plan at the full 20-layer shape."""
    import bench
    from rcdms_amd.sampler import PriorLoop
    from rcdms_amd.scheduler import UnCLIPScheduler
    from src.models.myprior_transformer import MyPriorTransformer
    mk = dict(num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
              temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)
    with torch.device("meta"):
        m = MyPriorTransformer(num_attention_heads=32, attention_head_dim=64, num_layers=20, embedding_dim=1280,
                               num_embeddings=91, additional_embeddings=6, unet_use_cross_frame_attention=False,
                               unet_use_temporal_attention=False, use_motion_module=True, motion_module_type="Vanilla",
                               motion_module_kwargs=mk)
    m = m.to_empty(device=dev).eval()
    bench.init_weights_(m)
    B, T, E = 10, 91, 1280
    g = torch.Generator(device=dev).manual_seed(42)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    mask = torch.ones(B, T, device=dev)
    mask[:, 20:] = 0
    loop = PriorLoop(m, 5, T, 4.0, UnCLIPScheduler(), 4)
    loop.load(rn(5, E), rn(B, E), rn(B, T, E), rn(B, E), rn(B, E), mask, generator=g)
    loop.run(use_graph=False)
    prior_plan.keep = (m, loop)
    return loop.prog.plan, torch.cuda.current_stream(dev)   # (the prior program launches on the caller's stream)


def tune(plan, stream, a):
    dev = torch.device("cuda", 0)
    for name in ("splitk_ws", "splitk_ws_side"):   # another variant may split where the planned one does not
        if name in plan.bufs:
            b = plan.bufs[name]
            b.t = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
            b.nbytes = 1 << 30
    shapes = [shape_of(t) for t in plan.tags]
    distinct = sorted({s for s in shapes if s})
    with torch.cuda.stream(stream):
        hip.set_shape_rules("")
        base = time_passes(plan, a.passes)
        res = {}
        for taps, cands in CAND.items():
            for v, sp in cands:
                hip.set_shape_rules(";".join(f"{t},{M},{N},{C},{v},{sp}" for t, M, N, C in distinct if t == taps))
                res[(taps, (v, sp))] = time_passes(plan, a.passes)
        hip.set_shape_rules(None)
    by_tag = {}
    for i, (tag, s) in enumerate(zip(plan.tags, shapes)):
        if s:
            by_tag.setdefault((tag, s), []).append(i)
    rules, total = {}, 0.0
    print(f"{'op':62s} {'n':>3s} {'library':>9s} | best candidate")
    for (tag, s), idx in sorted(by_tag.items(), key=lambda kv: -sum(base[i] for i in kv[1])):
        b = sum(base[i] for i in idx) / len(idx)
        best_v, best_t = None, b
        for v in CAND[s[0]]:
            t = sum(res[(s[0], v)][i] for i in idx) / len(idx)
            if t < best_t:
                best_v, best_t = v, t
        gain = (b - best_t) * len(idx)
        flag = ""
        if best_v is not None and b - best_t >= a.min_gain_us and (b - best_t) / b >= 0.03:
            flag = "  <--"
            # several tags can share a shape (epilogue variants): keep the rule only if no tag of the shape loses
            rules.setdefault(s, []).append((best_v, gain))
            total += gain
        print(f"{tag:62s} {len(idx):3d} {b:8.1f}us | " + (f"v{best_v[0]} split {best_v[1]}: {best_t:7.1f}us ({gain:6.1f} us per step){flag}" if best_v else "-"))
    print(f"sum of the flagged gains: {total / 1e3:.3f} ms per step (in-sequence timing)")
    out = []
    for s, lst in rules.items():
        vs = {v for v, _ in lst}
        if len(vs) == 1:
            v, sp = vs.pop()
            out.append(f"{s[0]},{s[1]},{s[2]},{s[3]},{v},{sp}")
    print("RCDM_SHAPE_RULES=" + ";".join(out))


if __name__ == "__main__":
    main()
