#!/usr/bin/env python
"""Micro-benchmark of single librcdm_hip kernels at the UNet's hot shapes (HIP-event timed, interleaved rounds).
usage: python tools/kbench.py [gemm|conv|attn|norm|ff|chain|ladder|all] [--variants 0,1,2] [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402

DEV = "cuda"
SPLIT_K = 0   # --split: forced split-K factor of the GEMM rows (0 = the library's heuristic)


def timeit(fn, rounds=5, inner=10):
    # warm-up long enough to sit out the clock ramp after an idle gap (the first milliseconds run at ~1.5 GHz,
    # tools/mfma_peak.py): otherwise whichever variant is timed first looks 10 % slow
    import time
    t_end = time.perf_counter() + 0.03
    while time.perf_counter() < t_end:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    best = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / inner * 1e3)
    best.sort()
    return best[len(best) // 2], best[0]


GEMMS = [  # (name, M, N, K, epi)
    ("L0 CxC K=320 +bias+res", 40960, 320, 320, 5),
    ("L0 qkv N=960", 40960, 960, 320, 0),
    ("L0 geglu N=2560", 40960, 2560, 320, 9),
    ("L0 ff-out K=1280", 40960, 320, 1280, 5),
    ("L1 CxC K=640", 10240, 640, 640, 5),
    ("L1 geglu N=5120", 10240, 5120, 640, 9),
    ("L1 ff-out K=2560", 10240, 640, 2560, 5),
    ("L2 CxC K=1280", 2560, 1280, 1280, 5),
    ("L2 geglu N=10240", 2560, 10240, 1280, 9),
    ("L2 ff-out K=5120", 2560, 1280, 5120, 5),
    ("L2 qkv N=3840", 2560, 3840, 1280, 0),
    ("L3 CxC K=1280", 640, 1280, 1280, 5),
    ("L3 qkv N=3840", 640, 3840, 1280, 0),
    ("L3 geglu N=10240", 640, 10240, 1280, 9),
    ("L3 ff-out K=5120", 640, 1280, 5120, 5),
    ("L1 ctx kv N=1280 K=768", 850, 1280, 768, 0),
    ("L2 ctx kv N=2560 K=768", 850, 2560, 768, 0),
    ("L1 qkv N=1920", 10240, 1920, 640, 0),
    ("prior CxC 2048", 970, 2048, 2048, 5),
    ("prior qkv N=6144", 970, 6144, 2048, 1),
    ("prior ff1 gelu N=8192", 970, 8192, 2048, 17),
    ("prior ff2 K=8192", 970, 2048, 8192, 5),
    ("prior geglu N=16384", 970, 16384, 2048, 9),
]
CONVS = [  # (name, n_img, H, W, cin, cout)
    ("L0 320->320 @64", 10, 64, 64, 320, 320),
    ("L0 640->320 @64", 10, 64, 64, 640, 320),
    ("L0 960->320 @64", 10, 64, 64, 960, 320),
    ("L1 640->640 @32", 10, 32, 32, 640, 640),
    ("L1 1280->640 @32", 10, 32, 32, 1280, 640),
    ("L1 1920->640 @32", 10, 32, 32, 1920, 640),
    ("L2 1280->1280 @16", 10, 16, 16, 1280, 1280),
    ("L2 2560->1280 @16", 10, 16, 16, 2560, 1280),
    ("L3 1280->1280 @8", 10, 8, 8, 1280, 1280),
    ("L3 2560->1280 @8", 10, 8, 8, 2560, 1280),
]


def bench_gemm(variants, rounds):
    for name, M, N, K, epi in GEMMS:
        A = torch.randn(M, K, device=DEV).half()
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
        bias = torch.randn(N, device=DEV)
        nout = N // 2 if epi & 8 else N
        res = torch.randn(M, nout, device=DEV).half()
        out = torch.empty(M, nout, device=DEV, dtype=torch.float16)
        line = f"gemm {name:28s} {2.0 * M * N * K / 1e9:8.1f} GF |"
        for v in variants:
            hip.set_igemm_pingpong(v != -2)      # -2: the shape heuristic with the ping-pong kernel switched off
            hip.set_igemm_variant(max(v, -1))
            d = hip.GemmDesc(M, N, K, K, nout, nout, epi, 1, 0, 1.0, SPLIT_K)
            wsb = hip.gemm_workspace_bytes(d)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
            fn = lambda: hip.gemm(d, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(),
                                  ws.data_ptr(), ws.numel())
            med, mn = timeit(fn, rounds)
            line += f" v{v}: {med:7.1f}us {2.0 * M * N * K / med / 1e6:6.0f}TF |"
        if N <= 320 and not (epi & (8 | 16 | 2)):   # the LayerNorm that follows, fused (rcdm_gemm_ln) vs as its own launch
            gam, bet = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
            y = torch.empty(M, N, device=DEV, dtype=torch.float16)
            d1 = hip.GemmDesc(M, N, K, K, N, N, epi, 1, 0, 1.0, 1, 0)
            ln = hip.LnFuse(gam.data_ptr(), bet.data_ptr(), 0, y.data_ptr(), N, 1, 1, 1e-5)
            lnd = hip.LayerNormDesc(M, N, N, N, 1e-5, 1, 1)
            ws1 = torch.empty(16, dtype=torch.uint8, device=DEV)
            hip.set_igemm_variant(-1)

            def pair():
                hip.gemm(d1, A.data_ptr(), W.data_ptr(), bias.data_ptr(), 0, res.data_ptr(), out.data_ptr(), ws1.data_ptr(), 16)
                hip.layernorm(lnd, out.data_ptr(), gam.data_ptr(), bet.data_ptr(), 0, y.data_ptr())
            t_pair, _ = timeit(pair, rounds)
            t_fused, _ = timeit(lambda: hip.gemm_ln(d1, ln, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                                    out.data_ptr()), rounds)
            line += f" gemm + layernorm {t_pair:6.1f}us, rcdm_gemm_ln {t_fused:6.1f}us |"
        print(line, flush=True)
    hip.set_igemm_variant(-1)
    hip.set_igemm_pingpong(True)


def bench_conv(variants, rounds):
    for name, n, H, W, cin, cout in CONVS:
        x = torch.randn(n * H * W, cin, device=DEV).half()
        w = (torch.randn(cout, 9 * cin, device=DEV) * (9 * cin) ** -0.5).half()
        bias = torch.randn(cout, device=DEV)
        out = torch.empty(n * H * W, cout, device=DEV, dtype=torch.float16)
        fl = 2.0 * n * H * W * 9 * cin * cout
        line = f"conv {name:28s} {fl / 1e9:8.1f} GF |"
        for v in variants:
            hip.set_igemm_pingpong(v != -2)
            hip.set_igemm_variant(max(v, -1))
            d = hip.ConvDesc(n, H, W, cin, cout, 1, 0, cin, cout, 0, 1, 1, 0, 1.0, 0)
            wsb = hip.conv3x3_workspace_bytes(d)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
            fn = lambda: hip.conv3x3(d, x.data_ptr(), w.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(), ws.data_ptr(),
                                     ws.numel())
            med, mn = timeit(fn, rounds)
            line += f" v{v}: {med:7.1f}us {fl / med / 1e6:6.0f}TF |"
        print(line, flush=True)
    hip.set_igemm_variant(-1)


def bench_attn(rounds, only=""):
    for name, batch, heads, L, Lk, d in [("L0 self d=40", 10, 8, 4096, 4096, 40), ("L1 self d=80", 10, 8, 1024, 1024, 80),
                                          ("L2 self d=160", 10, 8, 256, 256, 160), ("L0 cross Lk=85", 10, 8, 4096, 85, 40),
                                          ("L1 cross Lk=85", 10, 8, 1024, 85, 80), ("L2 cross Lk=85", 10, 8, 256, 85, 160),
                                          ("L3 cross Lk=85", 10, 8, 64, 85, 160)]:
        if only and only not in name:
            continue
        C = heads * d
        qkv = torch.randn(batch * L, 3 * C, device=DEV).half()
        kv = torch.randn(batch * Lk, 2 * C, device=DEV).half()
        out = torch.empty(batch * L, C, device=DEV, dtype=torch.float16)
        if L == Lk:
            desc = hip.AttnDesc(batch, heads, L, Lk, d, 3 * C, 3 * C, 3 * C, C, d ** -0.5)
            fn = lambda: hip.flash_attn(desc, qkv.data_ptr(), qkv.data_ptr() + 2 * C, qkv.data_ptr() + 4 * C, out.data_ptr())
        else:
            desc = hip.AttnDesc(batch, heads, L, Lk, d, 3 * C, 2 * C, 2 * C, C, d ** -0.5)
            fn = lambda: hip.flash_attn(desc, qkv.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * C, out.data_ptr())
        med, mn = timeit(fn, rounds)
        fl = 4.0 * batch * heads * L * Lk * d
        line = f"attn {name:28s} {fl / 1e9:8.1f} GF | {med:7.1f}us {fl / med / 1e6:6.0f}TF"
        if L != Lk and Lk <= 96:   # the short-key kernel on the same data (K / V image packed once, outside the timing)
            img = torch.empty(hip.xattn_image_bytes(batch, heads, d), dtype=torch.uint8, device=DEV)
            hip.xattn_pack_kv(kv.data_ptr(), kv.data_ptr() + 2 * C, batch, Lk, heads, d, 2 * C, 2 * C, img.data_ptr())
            med2, _ = timeit(lambda: hip.xattn(desc, qkv.data_ptr(), img.data_ptr(), out.data_ptr()), rounds)
            line += f" | rcdm_xattn {med2:7.1f}us {fl / med2 / 1e6:6.0f}TF"
        print(line, flush=True)
    for name, b, f, px, heads, d in [("L0 temporal", 2, 5, 4096, 8, 40), ("L1 temporal", 2, 5, 1024, 8, 80)]:
        if only and only not in name:
            continue
        C = heads * d
        qkv = torch.randn(b * f * px, 3 * C, device=DEV).half()
        out = torch.empty(b * f * px, C, device=DEV, dtype=torch.float16)
        desc = hip.TemporalAttnDesc(b, f, px, heads, d, 3 * C, C, d ** -0.5)
        med, mn = timeit(lambda: hip.temporal_attn(desc, qkv.data_ptr(), out.data_ptr()), rounds)
        by = qkv.numel() * 2 + out.numel() * 2
        print(f"attn {name:28s} {by / 1e6:8.1f} MB | {med:7.1f}us {by / med / 1e6:6.2f}TB/s", flush=True)


def bench_norm(rounds, only=""):
    for name, M, C in [("copy 26 MB (torch)", 40960, 320), ("copy 13 MB (torch)", 10240, 640), ("copy 6.5 MB (torch)", 2560, 1280),
                       ("copy 79 MB (torch)", 40960, 960)]:
        if only and only not in name:
            continue
        x = torch.randn(M, C, device=DEV).half()
        y = torch.empty_like(x)
        med, mn = timeit(lambda: y.copy_(x), rounds)  # bandwidth yardstick for the norm kernels below (same bytes)
        print(f"norm {name:28s} {4.0 * M * C / 1e6:8.1f} MB | {med:7.1f}us {4.0 * M * C / med / 1e6:6.2f}TB/s", flush=True)
    for name, M, C in [("L0 LN C=320", 40960, 320), ("L1 LN C=640", 10240, 640), ("L2 LN C=1280", 2560, 1280)]:
        if only and only not in name:
            continue
        x = torch.randn(M, C, device=DEV).half()
        y = torch.empty_like(x)
        g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        d = hip.LayerNormDesc(M, C, C, C, 1e-5, 1, 1)
        med, mn = timeit(lambda: hip.layernorm(d, x.data_ptr(), g.data_ptr(), b.data_ptr(), 0, y.data_ptr()), rounds)
        print(f"norm {name:28s} {4.0 * M * C / 1e6:8.1f} MB | {med:7.1f}us {4.0 * M * C / med / 1e6:6.2f}TB/s", flush=True)
    for name, smp, rps, C, silu in [("L0 GN cross C=320", 2, 20480, 320, 1), ("L0 GN cross C=960", 2, 20480, 960, 1),
                                    ("L0 GN frame C=320", 10, 4096, 320, 0), ("L2 GN cross C=1280", 2, 1280, 1280, 1),
                                    ("L1 GN cross C=640", 2, 5120, 640, 1), ("L1 GN frame C=640", 10, 1024, 640, 0)]:
        if only and only not in name:
            continue
        x = torch.randn(smp * rps, C, device=DEV).half()
        y = torch.empty_like(x)
        g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        d = hip.GroupNormDesc(smp, rps, C, 32, C, C, 1e-5, silu)
        ws = torch.empty(hip.groupnorm_workspace_bytes(d), dtype=torch.uint8, device=DEV)
        med, mn = timeit(lambda: hip.groupnorm_silu(d, x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                                    ws.numel()), rounds)
        by = 6.0 * smp * rps * C
        print(f"norm {name:28s} {by / 1e6:8.1f} MB | {med:7.1f}us {by / med / 1e6:6.2f}TB/s (2 reads + 1 write)", flush=True)


def bench_ff(rounds, only=""):
    """LayerNorm -> GEGLU feed-forward -> + residual: the three-launch chain vs the row-stationary fused kernel."""
    for name, M, C in [("L0 FF C=320", 40960, 320), ("L0 FF C=320 half", 20480, 320), ("L1 FF C=640", 10240, 640),
                       ("L2 FF C=1280", 2560, 1280)]:
        if only and only not in name:
            continue
        x = torch.randn(M, C, device=DEV).half()
        g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        w1 = torch.randn(8 * C, C, device=DEV) * C ** -0.5
        b1 = torch.randn(8 * C, device=DEV)
        w2 = torch.randn(C, 4 * C, device=DEV) * (4 * C) ** -0.5
        b2 = torch.randn(C, device=DEV)
        a = torch.empty_like(x)
        hid = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
        y = torch.empty_like(x)
        wp = torch.empty(8 * C, C, dtype=torch.float16, device=DEV)
        bp = torch.empty(8 * C, dtype=torch.float32, device=DEV)
        hip.pack_geglu_rows(w1.data_ptr(), b1.data_ptr(), 8 * C, C, wp.data_ptr(), bp.data_ptr())
        w2h = w2.half().contiguous()
        lnd = hip.LayerNormDesc(M, C, C, C, 1e-5, 1, 1)
        d1 = hip.GemmDesc(M, 8 * C, C, C, 4 * C, 0, 9, 1, 0, 1.0, 0)
        d2 = hip.GemmDesc(M, C, 4 * C, 4 * C, C, C, 5, 1, 0, 1.0, 0)
        ws = torch.zeros(max(hip.gemm_workspace_bytes(d1), hip.gemm_workspace_bytes(d2), 16), dtype=torch.uint8, device=DEV)

        def chain():
            hip.layernorm(lnd, x.data_ptr(), g.data_ptr(), b.data_ptr(), 0, a.data_ptr())
            hip.gemm(d1, a.data_ptr(), wp.data_ptr(), bp.data_ptr(), 0, 0, hid.data_ptr(), ws.data_ptr(), ws.numel())
            hip.gemm(d2, hid.data_ptr(), w2h.data_ptr(), b2.data_ptr(), 0, x.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel())
        fl = 2.0 * M * 12 * C * C
        med, mn = timeit(chain, rounds)
        line = f"ff   {name:28s} {fl / 1e9:8.1f} GF | chain {med:7.1f}us {fl / med / 1e6:6.0f}TF"
        if hip.ff_fused_supported(C):
            wsr = torch.empty(hip.ff_stream_bytes(C), dtype=torch.uint8, device=DEV)
            b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
            hip.pack_ff_stream(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), C, wsr.data_ptr(), b1p.data_ptr())
            fd = hip.FFDesc(M, C, C, C, 1e-5)
            med2, mn2 = timeit(lambda: hip.ff_fused(fd, x.data_ptr(), g.data_ptr(), b.data_ptr(), wsr.data_ptr(), b1p.data_ptr(),
                                                    b2.data_ptr(), y.data_ptr()), rounds)
            line += f" | rcdm_ff_fused {med2:7.1f}us (min {mn2:.1f}) {fl / med2 / 1e6:6.0f}TF"
        print(line, flush=True)


def bench_chain(rounds, only=""):
    """C x C GEMM (+bias +residual) -> LayerNorm (+pe) -> q | k | v GEMM: the three-launch chain vs rcdm_rowchain."""
    for name, M, C, tail in [("L0 o+res -> LN -> qkv", 40960, 320, 3), ("L0 o+res -> LN -> q", 40960, 320, 1),
                             ("L0 o+res -> LN -> FF", 40960, 320, 0)]:
        if only and only not in name:
            continue
        a = torch.randn(M, C, device=DEV).half()
        tok = torch.randn(M, C, device=DEV).half()
        wa = torch.randn(C, C, device=DEV) * C ** -0.5
        ba, g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        pe = torch.randn(5, C, device=DEV)
        ncol = tail * C if tail else C
        wt = torch.randn(max(tail, 1) * C, C, device=DEV) * C ** -0.5
        w1, b1 = torch.randn(8 * C, C, device=DEV) * C ** -0.5, torch.randn(8 * C, device=DEV)
        w2, b2 = torch.randn(C, 4 * C, device=DEV) * (4 * C) ** -0.5, torch.randn(C, device=DEV)
        y = torch.empty(M, C, device=DEV, dtype=torch.float16)
        out = torch.empty(M, ncol, device=DEV, dtype=torch.float16)
        hid = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
        wah, wth, w2h = wa.half().contiguous(), wt.half().contiguous(), w2.half().contiguous()
        wp = torch.empty(8 * C, C, dtype=torch.float16, device=DEV)
        bp = torch.empty(8 * C, dtype=torch.float32, device=DEV)
        hip.pack_geglu_rows(w1.data_ptr(), b1.data_ptr(), 8 * C, C, wp.data_ptr(), bp.data_ptr())
        d0 = hip.GemmDesc(M, C, C, C, C, C, 5, 1, 0, 1.0, 0)
        lnd = hip.LayerNormDesc(M, C, C, C, 1e-5, 4096, 5)
        dt = hip.GemmDesc(M, ncol, C, C, ncol, 0, 0, 1, 0, 1.0, 0)
        d1 = hip.GemmDesc(M, 8 * C, C, C, 4 * C, 0, 9, 1, 0, 1.0, 0)
        d2 = hip.GemmDesc(M, C, 4 * C, 4 * C, C, C, 5, 1, 0, 1.0, 0)
        ws = torch.zeros(1 << 24, dtype=torch.uint8, device=DEV)

        def chain():
            hip.gemm(d0, a.data_ptr(), wah.data_ptr(), ba.data_ptr(), 0, tok.data_ptr(), tok.data_ptr(), ws.data_ptr(), ws.numel())
            hip.layernorm(lnd, tok.data_ptr(), g.data_ptr(), b.data_ptr(), pe.data_ptr(), y.data_ptr())
            if tail:
                hip.gemm(dt, y.data_ptr(), wth.data_ptr(), 0, 0, 0, out.data_ptr(), ws.data_ptr(), ws.numel())
            else:
                hip.gemm(d1, y.data_ptr(), wp.data_ptr(), bp.data_ptr(), 0, 0, hid.data_ptr(), ws.data_ptr(), ws.numel())
                hip.gemm(d2, hid.data_ptr(), w2h.data_ptr(), b2.data_ptr(), 0, tok.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel())
        fl = 2.0 * M * C * C * (1 + (tail if tail else 12))
        med, mn = timeit(chain, rounds)
        wsr = torch.empty(hip.rowchain_stream_bytes(C, tail), dtype=torch.uint8, device=DEV)
        b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
        hip.pack_rowchain(wa.data_ptr(), C, tail, wt.data_ptr() if tail else 0, 0 if tail else w1.data_ptr(),
                          0 if tail else b1.data_ptr(), 0 if tail else w2.data_ptr(), wsr.data_ptr(), 0 if tail else b1p.data_ptr())
        rd = hip.RowChainDesc(M, C, C, C, C, ncol, tail, 4096, 5, 1e-5)
        med2, mn2 = timeit(lambda: hip.rowchain(rd, a.data_ptr(), tok.data_ptr(), tok.data_ptr(), ba.data_ptr(), g.data_ptr(),
                                                b.data_ptr(), pe.data_ptr(), wsr.data_ptr(), 0 if tail else b1p.data_ptr(),
                                                0 if tail else b2.data_ptr(), out.data_ptr()), rounds)
        print(f"chain {name:27s} {fl / 1e9:8.1f} GF | launches {med:7.1f}us {fl / med / 1e6:6.0f}TF | rcdm_rowchain {med2:7.1f}us "
              f"(min {mn2:.1f}) {fl / med2 / 1e6:6.0f}TF", flush=True)
        if tail == 0 and (not only or "proj_out" in only or only in name):
            # ... -> proj_out + bias + the block's input: one more C x C GEMM launch, or tail 2 of the chain
            wz, bz = torch.randn(C, C, device=DEV) * C ** -0.5, torch.randn(C, device=DEV)
            wzh, xin, fin = wz.half().contiguous(), torch.randn(M, C, device=DEV).half(), torch.empty(M, C, device=DEV, dtype=torch.float16)

            def chain_z():
                chain()
                hip.gemm(d0, out.data_ptr(), wzh.data_ptr(), bz.data_ptr(), 0, xin.data_ptr(), fin.data_ptr(), ws.data_ptr(), ws.numel())
            flz = fl + 2.0 * M * C * C
            medz, _ = timeit(chain_z, rounds)
            wsz = torch.empty(hip.rowchain_stream_bytes(C, 2), dtype=torch.uint8, device=DEV)
            hip.pack_rowchain(wa.data_ptr(), C, 2, wz.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), wsz.data_ptr(), b1p.data_ptr())
            rz = hip.RowChainDesc(M, C, C, C, C, C, 2, 1, 1, 1e-5, 0, 0, C)
            medz2, mnz2 = timeit(lambda: hip.rowchain(rz, a.data_ptr(), tok.data_ptr(), tok.data_ptr(), ba.data_ptr(), g.data_ptr(),
                                                      b.data_ptr(), 0, wsz.data_ptr(), b1p.data_ptr(), b2.data_ptr(), fin.data_ptr(),
                                                      z_res=xin.data_ptr(), z_bias=bz.data_ptr()), rounds)
            print(f"chain {'L0 o+res -> LN -> FF -> proj_out':27s} {flz / 1e9:8.1f} GF | launches {medz:7.1f}us {flz / medz / 1e6:6.0f}TF | "
                  f"rcdm_rowchain {medz2:7.1f}us (min {mnz2:.1f}) {flz / medz2 / 1e6:6.0f}TF", flush=True)
        if tail == 3 and (not only or "GN" in only or only in name):
            # GroupNorm (per frame) -> proj_in -> LN + pe -> qkv: four + three launches, or statistics + the chain with the apply inside
            gd = hip.GroupNormDesc(M // 4096, 4096, C, 32, C, C, 1e-6, 0)
            gws = torch.zeros(hip.groupnorm_workspace_bytes(gd), dtype=torch.uint8, device=DEV)
            stat = torch.zeros(M // 4096 * 64, dtype=torch.float32, device=DEV)
            xn = torch.empty(M, C, device=DEV, dtype=torch.float16)
            d00 = hip.GemmDesc(M, C, C, C, C, 0, 1, 1, 0, 1.0, 0)

            def chain_gn():
                hip.groupnorm_silu(gd, a.data_ptr(), g.data_ptr(), b.data_ptr(), xn.data_ptr(), gws.data_ptr(), gws.numel())
                hip.gemm(d00, xn.data_ptr(), wah.data_ptr(), ba.data_ptr(), 0, 0, tok.data_ptr(), ws.data_ptr(), ws.numel())
                hip.layernorm(lnd, tok.data_ptr(), g.data_ptr(), b.data_ptr(), pe.data_ptr(), y.data_ptr())
                hip.gemm(dt, y.data_ptr(), wth.data_ptr(), 0, 0, 0, out.data_ptr(), ws.data_ptr(), ws.numel())
            medg, _ = timeit(chain_gn, rounds)
            rg = hip.RowChainDesc(M, C, C, 0, C, ncol, 3, 4096, 5, 1e-5, 32, 4096)

            def fused_gn():
                hip.groupnorm_stats(gd, a.data_ptr(), stat.data_ptr(), gws.data_ptr(), gws.numel())
                hip.rowchain(rg, a.data_ptr(), 0, tok.data_ptr(), ba.data_ptr(), g.data_ptr(), b.data_ptr(), pe.data_ptr(), wsr.data_ptr(),
                             0, 0, out.data_ptr(), gn_stat=stat.data_ptr(), gn_g=g.data_ptr(), gn_b=b.data_ptr())
            medg2, mng2 = timeit(fused_gn, rounds)
            print(f"chain {'L0 GN -> proj_in -> LN -> qkv':27s} {fl / 1e9:8.1f} GF | launches {medg:7.1f}us {fl / medg / 1e6:6.0f}TF | "
                  f"stats + rcdm_rowchain {medg2:7.1f}us (min {mng2:.1f}) {fl / medg2 / 1e6:6.0f}TF", flush=True)


def bench_ladder(rounds):
    """The guide's yardstick (cdna_hip_programming.md, "optimization ladder at 4096^3": 874 TFLOP/s for the 128x128 two-barrier
    structure, ~1330 for the 256x256 8-phase + swizzle template, uniform random operands): rcdm_gemm at M = N = K = 4096 (and
    8192), f16, operands uniform in [-1, 1), no epilogue, every tile variant of the library, unsplit and at its best split."""
    names = {1: "dma 128x128", 2: "dma 256x256", 3: "dma 64x64 x4", 4: "dma 64x64", 5: "dma 128x64", 6: "pp 160x320", 7: "pp 160x256",
             8: "pp 256x256", 9: "i16 160x160", 10: "dma 128x64 ring3", -1: "library's pick"}
    for n in (4096, 8192):
        A = (torch.rand(n, n, device=DEV) * 2 - 1).half()
        W = (torch.rand(n, n, device=DEV) * 2 - 1).half()
        out = torch.empty(n, n, device=DEV, dtype=torch.float16)
        fl = 2.0 * n ** 3
        print(f"ladder rcdm_gemm M = N = K = {n}, f16 uniform [-1, 1), epi 0   (guide: 874 step-3 / ~1330 8-phase @4096; ~1470 8-phase @8192)", flush=True)
        for v in (-1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
            hip.set_igemm_pingpong(True)
            hip.set_igemm_variant(v)
            best = None
            line = f"ladder {n:5d} v{v:<2d} {names[v]:18s} |"
            for split in (1, 2):
                d = hip.GemmDesc(n, n, n, n, n, 0, 0, 1, 0, 1.0, split)
                try:
                    wsb = hip.gemm_workspace_bytes(d)
                    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=DEV)
                    fn = lambda: hip.gemm(d, A.data_ptr(), W.data_ptr(), 0, 0, 0, out.data_ptr(), ws.data_ptr(), ws.numel())
                    fn()
                    med, mn = timeit(fn, rounds, inner=5)
                except hip.RcdmError as e:
                    line += f" split {split}: {str(e)[:30]} |"
                    continue
                line += f" split {split}: {med:8.1f}us {fl / med / 1e6:6.0f}TF (best {fl / mn / 1e6:6.0f}) |"
                best = max(best or 0.0, fl / med / 1e6)
            print(line + (f"  => {best:.0f} TF" if best else ""), flush=True)
    hip.set_igemm_variant(-1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--variants", default="1,2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--split", type=int, default=0, help="force the GEMMs' split-K factor")
    a = ap.parse_args()
    if a.only:
        GEMMS[:] = [g for g in GEMMS if a.only in g[0]]
        CONVS[:] = [c for c in CONVS if a.only in c[0]]
    SPLIT_K = a.split
    vs = [int(v) for v in a.variants.split(",")]
    if a.what in ("gemm", "all"):
        bench_gemm(vs, a.rounds)
    if a.what in ("conv", "all"):
        bench_conv(vs, a.rounds)
    if a.what in ("attn", "all"):
        bench_attn(a.rounds, a.only)
    if a.what in ("norm", "all"):
        bench_norm(a.rounds, a.only)
    if a.what in ("ff", "all"):
        bench_ff(a.rounds, a.only)
    if a.what in ("chain", "all"):
        bench_chain(a.rounds, a.only)
    if a.what == "ladder":
        bench_ladder(a.rounds)
