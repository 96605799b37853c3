#!/usr/bin/env python
"""profiles/r6_ceiling.txt — where the denoising step's time is, launch by launch, against what ANY tile of this library
reaches on each exact shape (task 1: "publish the ceiling table").

For every op of the bench workload's launch plan (BASELINE config 2: b = 2 x 5 frames, 64x64 latents, L = 85):
  t_seq     its time INSIDE the step's own launch sequence (one execution per pass between two events, operands as cold as in
            the replayed graph; median over --passes) — these sum to the eager step; the replayed graph's step is printed beside it
  GFLOP     algorithmic work (2 MAC; tools/opprof.py's convention = SURVEY 8d)
  rate      GFLOP / t_seq
and, for the contractions (rcdm_gemm* / rcdm_conv3x3*):
  tile      the library's choice for the shape: variant, BM x BN, tiles, split-K (rcdm_*_plan_query)
  q_eff     tile-quantisation efficiency of that choice = (useful outputs / outputs of the padded tile grid)
            x (tiles x splits) / (rounds x resident slots), slots = CUs x blocks per CU of the tile
  best_seq  the best (variant, split) candidate's time in the same position of the sequence (tools/tune_rules.py's sweep)
  warm      the op ISOLATED and WARM: repeated back to back on the same operands, the library's choice ...
  best_warm ... and the best candidate — "the best isolated warm rate of any tile at that exact shape"
  fixed     prologue + epilogue (+ launch) share of the op: T(K) = a + b K fitted from the isolated warm time at K and 2 K
            (same M, N, epilogue, tile family), fixed = a / T(K)
The bottom of the file sums the columns per kernel family and per resolution level: t_seq vs (GFLOP / best_warm rate) is the
part of the gap to the library's own best tile that is NOT the main loop's rate — cold operands, launch ramps, quantisation,
prologue / epilogue — and (GFLOP / 1221 TFLOP/s), the 4096^3 yardstick of profiles/r5_gemm_ladder.txt, what the step would
take if every contraction ran at the best rate this library reaches on ANY shape.
usage: python tools/ceiling.py [--passes 5] [--iters 8] > profiles/r6_ceiling.txt"""
import argparse
import collections
import ctypes
import os
import re
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rcdms_amd import hip  # noqa: E402
import opprof  # noqa: E402
import tune_rules  # noqa: E402

YARDSTICK_TFLOPS = 1221.0   # rcdm_gemm at 4096^3 on the 256x256 ping-pong tile (profiles/r5_gemm_ladder.txt)
VARIANT_NAMES = {0: "128x128", 1: "128x128", 2: "256x256", 3: "64x64", 4: "64x64/4", 5: "128x64", 6: "pp160x320", 7: "pp160x256",
                 8: "pp256x256", 9: "i16 160x160", 10: "128x64/3"}


def level_of(tag):
    """Resolution level of an op from its row count (b f = 10 images)."""
    kv = {k: float(v) for k, v in re.findall(r"(\w+)=(-?[\d.]+)", tag)}
    kind = tag.split()[0]
    rows = None
    if kind in ("gemm", "rowchain", "ff_fused", "layernorm"):
        rows = kv.get("M")
    elif kind == "upsample_gather":
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        rows = n * H * W * 4
    elif kind == "conv3x3_wino":
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        rows = n * H * W
    elif kind == "conv3x3":
        n, H, W = (float(v) for v in re.search(r"(\d+)x(\d+)x(\d+)", tag).groups())
        rows = n * H * W * (4.0 if kv.get("up") else 1.0) / (kv.get("s", 1.0) ** 2)
    elif kind in ("flash_attn", "xattn"):
        rows = kv["B"] * kv["Lq"]
    elif kind == "temporal_attn":
        rows = kv["S"] * kv["F"] * kv["P"]
    elif kind.startswith("groupnorm"):
        rows = kv["S"] * kv["R"]
    if rows is None:
        return "other"
    for name, r in (("64^2", 40960), ("32^2", 10240), ("16^2", 2560), ("8^2", 640)):
        if rows >= 0.45 * r:
            return name
    return "other"


def q_eff(pl, M, N, cus):
    useful = (M * N) / float(pl["tiles_m"] * pl["bm"] * pl["tiles_n"] * pl["bn"])
    work = pl["tiles_m"] * pl["tiles_n"] * pl["splits"]
    slots = cus * pl["blocks_per_cu"]
    rounds = (work + slots - 1) // slots
    return useful * work / float(rounds * slots)


def warm_time(op, iters):
    op()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def fixed_share(kind_desc, iters, dev):
    """a / T(K) from the isolated warm times at K and 2 K (scratch operands, the library's own tile rules for each)."""
    kind, d = kind_desc[0], kind_desc[1]
    try:
        if kind == "gemm":
            M, N, K = d.M, d.N, d.K
            ts = []
            for k in (K, 2 * K):
                A = torch.randn(M, k, device=dev).half()
                W = (torch.randn(N, k, device=dev) * k ** -0.5).half()
                out = torch.empty(M + d.dup_rows, max(N, 8), dtype=torch.float16, device=dev)
                bias = torch.zeros(N, device=dev)
                epi = d.epilogue & (hip.EPI_BIAS | hip.EPI_GEGLU | hip.EPI_GELU)
                dd = hip.GemmDesc(M, N, k, k, out.shape[1], 0, epi, 1, 0, 1.0, 0, 0)
                ws = torch.empty(max(hip.gemm_workspace_bytes(dd), 256), dtype=torch.uint8, device=dev)
                ts.append(warm_time(lambda: hip.gemm(dd, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(),
                                                     ws.data_ptr(), ws.numel()), iters))
        else:
            if d.upsample == 2 or d.c_in2:
                return None
            ts = []
            for c in (d.c_in, 2 * d.c_in):
                rows = d.n_img * d.h_in * d.w_in
                x = torch.randn(rows, c, device=dev).half()
                W = (torch.randn(d.c_out, 9 * c, device=dev) * (9 * c) ** -0.5).half()
                ho = ((d.h_in << (1 if d.upsample else 0)) - 1) // d.stride + 1
                wo = ((d.w_in << (1 if d.upsample else 0)) - 1) // d.stride + 1
                out = torch.empty(d.n_img * ho * wo, d.c_out, dtype=torch.float16, device=dev)
                bias = torch.zeros(d.c_out, device=dev)
                dd = hip.ConvDesc(d.n_img, d.h_in, d.w_in, c, d.c_out, d.stride, d.upsample, c, d.c_out, 0, hip.EPI_BIAS, 1, 0, 1.0,
                                  0, d.pad_after_only, 0, 0, 0)
                ws = torch.empty(max(hip.conv3x3_workspace_bytes(dd), 256), dtype=torch.uint8, device=dev)
                ts.append(warm_time(lambda: hip.conv3x3(dd, x.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(),
                                                        ws.data_ptr(), ws.numel()), iters))
        a = 2 * ts[0] - ts[1]
        return max(0.0, min(1.0, a / ts[0]))
    except hip.RcdmError:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    a = ap.parse_args()
    import bench
    from rcdms_amd import synth
    from rcdms_amd.sampler import DenoiseLoop
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    model = bench.build_model(dev)
    story = synth.synthetic_story(stories=1, latent_hw=(64, 64), ctx_len=85, seed=42)
    T = 8
    loop = DenoiseLoop(model, 1, 5, 64, 64, 85, 2.0, bench.ddim(), T)
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    ev0, ev1 = hip.Event(), hip.Event()
    loop.run()
    loop.load(story["latents"], story["mask"], story["masked_latents"], story["ctx"])
    sp = loop.prog.stream.cuda_stream
    ev0.record(sp)
    loop.run()
    ev1.record(sp)
    torch.cuda.synchronize()
    graph_ms = ev0.elapsed_ms(ev1) / T
    plan, stream = loop.prog.plan, loop.prog.stream
    body = list(range(loop.prog.n_time_ops, len(plan.ops)))     # the captured step replays the body (time rows come from a table)
    for name in ("splitk_ws",):      # another variant may split where the planned one does not
        b = plan.bufs[name]
        b.t = torch.zeros(1 << 30, dtype=torch.uint8, device=dev)
        b.nbytes = 1 << 30
    shapes = [tune_rules.shape_of(t) for t in plan.tags]
    distinct = sorted({s for s in shapes if s})
    first_of = {}
    for i in body:
        if shapes[i] and (plan.tags[i], shapes[i]) not in first_of:
            first_of[(plan.tags[i], shapes[i])] = i
    with torch.cuda.stream(stream):
        hip.set_shape_rules("")
        base = tune_rules.time_passes(plan, a.passes)
        warm = {k: warm_time(plan.ops[i], a.iters) for k, i in first_of.items()}
        seq_c, warm_c = {}, {}
        for taps, cands in tune_rules.CAND.items():
            for v, spl in cands:
                hip.set_shape_rules(";".join(f"{t},{M},{N},{C},{v},{spl}" for t, M, N, C in distinct if t == taps))
                seq_c[(taps, (v, spl))] = tune_rules.time_passes(plan, max(3, a.passes - 2))
                for k, i in first_of.items():
                    if k[1][0] != taps:
                        continue
                    try:
                        warm_c.setdefault(k, {})[(v, spl)] = warm_time(plan.ops[i], a.iters)
                    except hip.RcdmError:
                        pass
        hip.set_shape_rules(None)
        fixed = {k: fixed_share(plan.op_desc[i], a.iters, dev) for k, i in first_of.items() if i in plan.op_desc}
    torch.cuda.synchronize()

    groups = collections.OrderedDict()
    for i in body:
        groups.setdefault(plan.tags[i], []).append(i)
    rows = []
    for tag, idx in groups.items():
        n = len(idx)
        t_seq = sum(base[i] for i in idx) / n
        gf, mb = opprof.algorithmic_work(tag)
        r = dict(tag=tag, n=n, t_seq=t_seq, gf=gf, mb=mb, level=level_of(tag), kind=tag.split()[0])
        s = shapes[idx[0]]
        if s and idx[0] in plan.op_desc:
            od = plan.op_desc[idx[0]]
            pl = hip.gemm_plan_query(od[1], od[2], od[3]) if od[0] == "gemm" else hip.conv3x3_plan_query(od[1])
            k = (tag, s)
            r.update(pl=pl, q=q_eff(pl, s[1], s[2], cus), warm=warm.get(k), fixed=fixed.get(k))
            best_seq = min((sum(seq_c[c][i] for i in idx) / n, c[1]) for c in seq_c if c[0] == s[0])
            r["best_seq"] = best_seq if best_seq[0] < t_seq else (t_seq, "lib")
            wc = warm_c.get(k, {})
            if wc:
                bw = min((t, c) for c, t in wc.items())
                r["best_warm"] = bw if (r["warm"] is None or bw[0] < r["warm"]) else (r["warm"], "lib")
        rows.append(r)

    tot_seq = sum(r["t_seq"] * r["n"] for r in rows)
    print(f"# r6 ceiling table — BASELINE config 2 (b = 2 x 5 frames, 64x64 latents, L = 85), {len(body)} ops per step on {cus} CUs")
    print(f"# replayed graph: {graph_ms:.3f} ms per step;  sum of in-sequence op times (eager, events around every op): {tot_seq / 1e3:.3f} ms")
    print("# columns: ms/step | n | t_seq us | GFLOP | TFLOP/s || tile (variant BMxBN tiles x split) | q_eff | best_seq us (variant,split) | "
          "warm us | best_warm us (variant,split) -> TFLOP/s | fixed share")
    for r in sorted(rows, key=lambda r: -r["t_seq"] * r["n"]):
        line = f"{r['t_seq'] * r['n'] / 1e3:7.3f} {r['n']:3d} {r['t_seq']:8.1f} "
        line += f"{r['gf']:8.2f} {r['gf'] / r['t_seq'] * 1e3:6.0f} " if r["gf"] else f"{'':8s} {'':6s} "
        if "pl" in r:
            pl = r["pl"]
            line += (f"|| v{pl['variant']:<2d} {pl['bm']}x{pl['bn']} {pl['tiles_m']}x{pl['tiles_n']}x{pl['splits']} q={r['q']:.2f} | "
                     f"{r['best_seq'][0]:7.1f} {str(r['best_seq'][1]):9s} | ")
            line += f"{r['warm']:7.1f} " if r.get("warm") else f"{'':7s} "
            if "best_warm" in r:
                bw = r["best_warm"]
                line += f"{bw[0]:7.1f} {str(bw[1]):9s} -> {r['gf'] / bw[0] * 1e3:5.0f} "
            line += f"| fixed {100 * r['fixed']:3.0f}% " if r.get("fixed") is not None else "| fixed  n/a "
        else:
            line += f"|| {'(not a tile kernel)':24s} " + (f"{r['mb']:7.1f} MB {r['mb'] / r['t_seq']:5.2f} TB/s " if r["mb"] else "")
        print(line + " " + r["tag"])

    def summarise(key, title):
        print(f"\n# {title}: ms per step in sequence | at each op's best-warm rate | at the {YARDSTICK_TFLOPS:.0f} TFLOP/s yardstick | "
              "launches | GFLOP | mean q_eff (time-weighted) | mean fixed share (time-weighted)")
        agg = collections.OrderedDict()
        for r in rows:
            g = agg.setdefault(key(r), dict(t=0.0, tb=0.0, ty=0.0, n=0, gf=0.0, qw=0.0, fw=0.0, tw=0.0, tfw=0.0))
            t = r["t_seq"] * r["n"]
            g["t"] += t
            g["n"] += r["n"]
            if "pl" in r and r["gf"]:
                g["gf"] += r["gf"] * r["n"]
                bw = r.get("best_warm", (r["t_seq"],))[0]
                g["tb"] += min(bw, r["t_seq"]) * r["n"]
                g["ty"] += r["gf"] / YARDSTICK_TFLOPS * 1e3 * r["n"]
                g["qw"] += r["q"] * t
                g["tw"] += t
                if r.get("fixed") is not None:
                    g["fw"] += r["fixed"] * t
                    g["tfw"] += t
            else:
                g["tb"] += t
                g["ty"] += t
        for k, g in sorted(agg.items(), key=lambda kv: -kv[1]["t"]):
            q = f"{g['qw'] / g['tw']:.2f}" if g["tw"] else " n/a"
            f = f"{100 * g['fw'] / g['tfw']:.0f}%" if g["tfw"] else "n/a"
            print(f"{str(k):28s} {g['t'] / 1e3:7.3f} {g['tb'] / 1e3:7.3f} {g['ty'] / 1e3:7.3f} {g['n']:4d} {g['gf']:9.1f} {q:>5s} {f:>5s}")
        tt = sum(g["t"] for g in agg.values()) / 1e3
        tb = sum(g["tb"] for g in agg.values()) / 1e3
        ty = sum(g["ty"] for g in agg.values()) / 1e3
        print(f"{'TOTAL':28s} {tt:7.3f} {tb:7.3f} {ty:7.3f}")
        return tt, tb, ty

    summarise(lambda r: r["kind"] if "pl" not in r else ("gemm/conv " + ("split-K" if r["pl"]["splits"] > 1 else "unsplit")), "by kernel family")
    tt, tb, ty = summarise(lambda r: r["level"], "by resolution level")
    print(f"\n# reading: the step spends {tt:.2f} ms in sequence ({graph_ms:.2f} ms as a replayed graph).  With every contraction at the best "
          f"isolated WARM rate any tile of this library reaches on its exact shape it would spend {tb:.2f} ms (the {tt - tb:.2f} ms between the two "
          f"are cold operands + launch ramps, not tile choice); with every contraction at the {YARDSTICK_TFLOPS:.0f} TFLOP/s of the 4096^3 yardstick, "
          f"{ty:.2f} ms = {11044 / ty:.0f} TFLOP/s = {100 * 11044 / ty / 2500:.1f} % of the 2.5 PFLOP/s peak — the ceiling of this decomposition "
          "(one launch per GEMM-shaped op at b f = 10 images, non-contraction launches as they are).")


if __name__ == "__main__":
    main()
