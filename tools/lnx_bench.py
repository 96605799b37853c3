#!/usr/bin/env python
"""Deferred LayerNorm (rcdm_gemm_lnx) against the three launches it replaces, at the UNet's deep-level shapes:
  separate:  producer GEMM (to_out + residual) -> rcdm_layernorm -> consumer GEMM (q|k|v / to_q / GEGLU)
  deferred:  producer GEMM with row statistics -> consumer GEMM on the raw rows
HIP-event timed, interleaved rounds, each op alone and the sequence.  usage: python tools/lnx_bench.py [--rounds 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rcdms_amd import hip  # noqa: E402
from tools.kbench import timeit  # noqa: E402

DEV = "cuda"
LEVELS = [("32x32", 10240, 640), ("16x16", 2560, 1280), ("8x8", 640, 1280)]
CONSUMERS = [("qkv", 3, 0), ("to_q", 1, 0), ("geglu", 8, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    hip.load()
    for lname, M, C in LEVELS:
        x_in = torch.randn(M, C, device=DEV).half()
        res = torch.randn(M, C, device=DEV).half()
        Wp = (torch.randn(C, C, device=DEV) * C ** -0.5).half()
        bp = torch.randn(C, device=DEV)
        tok = torch.empty(M, C, dtype=torch.float16, device=DEV)
        nrm = torch.empty(M, C, dtype=torch.float16, device=DEV)
        g, b = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        dp = hip.GemmDesc(M, C, C, C, C, C, 1 | 4, 1, 0, 1.0, 0, 0)
        parts = hip.gemm_stat_parts(dp)
        stat = torch.empty(M * max(parts, 1) * 2, dtype=torch.float32, device=DEV)
        lxp = hip.Lnx(stat.data_ptr(), parts, M, 0, 0, 0, 0, 1e-5, 0)
        ld = hip.LayerNormDesc(M, C, C, C, 1e-5, 1, 1)

        def prod():
            hip.gemm(dp, x_in.data_ptr(), Wp.data_ptr(), bp.data_ptr(), 0, res.data_ptr(), tok.data_ptr(), 0, 0)

        def prod_stat():
            hip.gemm_lnx(dp, lxp, x_in.data_ptr(), Wp.data_ptr(), bp.data_ptr(), 0, res.data_ptr(), tok.data_ptr(), 0, 0)

        def lnorm():
            hip.layernorm(ld, tok.data_ptr(), g.data_ptr(), b.data_ptr(), 0, nrm.data_ptr())

        t_p, t_ps, t_ln = timeit(prod, a.rounds)[0], timeit(prod_stat, a.rounds)[0], timeit(lnorm, a.rounds)[0]
        print(f"{lname} M={M} C={C}: producer {t_p:.1f} us, with statistics {t_ps:.1f} us ({parts} slots), layernorm {t_ln:.1f} us")
        for cname, mult, epi_extra in CONSUMERS:
            N = mult * C
            Wc = (torch.randn(N, C, device=DEV) * C ** -0.5).half()
            bc = torch.randn(N, device=DEV)
            S = Wc.float().sum(dim=1).contiguous()
            out = torch.empty(M, N // 2 if epi_extra else N, dtype=torch.float16, device=DEV)
            plain_epi = (1 | 8) if epi_extra else 0
            dc0 = hip.GemmDesc(M, N, C, C, out.shape[1], 0, plain_epi, 1, 0, 1.0, 0, 0)
            dc1 = hip.GemmDesc(M, N, C, C, out.shape[1], 0, 1 | epi_extra, 1, 0, 1.0, 0, 0)
            lxc = hip.Lnx(0, 0, 0, stat.data_ptr(), parts, M, S.data_ptr(), 1e-5, C)
            w0 = torch.empty(max(hip.gemm_workspace_bytes(dc0), 16), dtype=torch.uint8, device=DEV)

            def cons():
                hip.gemm(dc0, nrm.data_ptr(), Wc.data_ptr(), bc.data_ptr(), 0, 0, out.data_ptr(), w0.data_ptr(), w0.numel())

            def cons_lnx():
                hip.gemm_lnx(dc1, lxc, tok.data_ptr(), Wc.data_ptr(), bc.data_ptr(), 0, 0, out.data_ptr(), 0, 0)

            dcb = hip.GemmDesc(M, N, C, C, out.shape[1], 0, 1 | epi_extra, 1, 0, 1.0, 0, 0)

            def cons_bias():   # the separate-launch consumer WITH a bias epilogue (what the deferred form's b' costs alone)
                hip.gemm(dcb, nrm.data_ptr(), Wc.data_ptr(), bc.data_ptr(), 0, 0, out.data_ptr(), w0.data_ptr(), w0.numel())

            def seq0():
                prod(); lnorm(); cons()

            def seq1():
                prod_stat(); cons_lnx()

            prod_stat()
            t_c, t_cl = timeit(cons, a.rounds)[0], timeit(cons_lnx, a.rounds)[0]
            t_cb = timeit(cons_bias, a.rounds)[0]
            t_s0, t_s1 = timeit(seq0, a.rounds)[0], timeit(seq1, a.rounds)[0]
            print(f"   {cname:6s} N={N:5d}: consumer {t_c:6.1f} (+bias {t_cb:6.1f}) -> deferred {t_cl:6.1f} us;   sequence {t_s0:6.1f} -> {t_s1:6.1f} us")


if __name__ == "__main__":
    main()
