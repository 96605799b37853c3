"""GPU: seeded random-shape sweep of the C-ABI kernels against torch fp32 on f16-rounded inputs — ragged M / N / K,
padded leading dimensions, every epilogue combination, automatic tile-variant and split-K choice.  Complements the
hand-picked cases of test_hip_kernels.py; shapes are drawn from a fixed seed so failures reproduce."""
import random

import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O
from tests.test_hip_kernels import DEV, close, h16, rows_from_5d, rows_to_5d, ws

pytestmark = pytest.mark.gpu


def test_gemm_random_shapes(hiplib):
    from rcdms_amd import hip
    rnd = random.Random(1234)
    for case in range(40):
        M = rnd.choice([1, 7, 64, 97, 130, 257, 640, 970, 1500])
        N = 8 * rnd.randint(1, 90)
        K = 8 * rnd.randint(1, 70)
        geglu = rnd.random() < 0.2
        if geglu:
            N = 64 * rnd.randint(1, 10)
        epi = (hip.EPI_BIAS if rnd.random() < 0.7 or geglu else 0) | (hip.EPI_GEGLU if geglu else 0)
        if not geglu:
            epi |= rnd.choice([0, hip.EPI_ROWVEC]) | rnd.choice([0, hip.EPI_RESIDUAL]) | rnd.choice([0, 0, hip.EPI_GELU])
        split = rnd.choice([0, 0, 1, 2, 3])
        scale = rnd.choice([1.0, 1.0, 0.5, 1 / 1.3])      # 1 / output_scale_factor (resnet.py:210)
        g = torch.Generator().manual_seed(case)
        A = h16(torch.randn(M, K, generator=g))
        W = h16(torch.randn(N, K, generator=g) * K ** -0.5)
        bias = torch.randn(N, generator=g)
        rps = rnd.choice([1, 50, 128])
        Nout = N // 2 if geglu else N
        rowvec = torch.randn((M + rps - 1) // rps, Nout, generator=g)
        res = h16(torch.randn(M, Nout, generator=g))
        lda, ldc, ldr = K + 8 * rnd.randint(0, 2), Nout + 8 * rnd.randint(0, 2), Nout + 8 * rnd.randint(0, 2)
        Ad = torch.zeros(M, lda, dtype=torch.float16); Ad[:, :K] = A.half(); Ad = Ad.to(DEV)
        Rd = torch.zeros(M, ldr, dtype=torch.float16); Rd[:, :Nout] = res.half(); Rd = Rd.to(DEV)
        out = torch.full((M, ldc), float("nan"), dtype=torch.float16, device=DEV)
        if geglu:
            w32, b32 = W.to(DEV), bias.to(DEV)
            Wd = torch.empty(N, K, dtype=torch.float16, device=DEV)
            bd = torch.empty(N, dtype=torch.float32, device=DEV)
            hip.pack_geglu_rows(w32.data_ptr(), b32.data_ptr(), N, K, Wd.data_ptr(), bd.data_ptr())
            hg = F.linear(A, W, bias)
            hid, gate = hg.chunk(2, dim=-1)
            ref = hid * F.gelu(gate)
        else:
            Wd, bd = W.half().to(DEV), bias.to(DEV)
            ref = A @ W.t()
            if epi & hip.EPI_BIAS:
                ref = ref + bias
            if epi & hip.EPI_ROWVEC:
                ref = ref + rowvec[torch.arange(M) // rps]
            if epi & hip.EPI_GELU:
                ref = F.gelu(ref)
            if epi & hip.EPI_RESIDUAL:
                ref = ref + res
        ref = ref * scale
        rvd = rowvec.to(DEV)
        d = hip.GemmDesc(M, N, K, lda, ldc, ldr, epi, rps, Nout, scale, split)
        w = ws(hip.gemm_workspace_bytes(d))
        hip.gemm(d, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), rvd.data_ptr(), Rd.data_ptr(), out.data_ptr(),
                 w.data_ptr(), w.numel())
        torch.cuda.synchronize()
        try:
            close(out[:, :Nout], ref)
        except AssertionError as e:
            raise AssertionError(f"case {case}: M={M} N={N} K={K} epi={epi} split={split} scale={scale:.3f} rps={rps} "
                                 f"lda={lda} ldc={ldc}: {e}")
        assert torch.isnan(out[:, Nout:].float()).all(), f"case {case}: wrote outside the N columns"


def test_conv_random_shapes(hiplib):
    from rcdms_amd import hip
    rnd = random.Random(4321)
    for case in range(16):
        n = rnd.choice([1, 2, 5])
        H, W = rnd.choice([4, 6, 8, 10, 16]), rnd.choice([4, 8, 12, 16])
        cin, cout = 8 * rnd.randint(1, 40), 8 * rnd.randint(1, 40)
        stride, up = rnd.choice([(1, 0), (1, 0), (2, 0), (1, 1)])
        if stride == 2 and (H % 2 or W % 2):
            H, W = H + H % 2, W + W % 2
        g = torch.Generator().manual_seed(100 + case)
        x = h16(torch.randn(1, cin, n, H, W, generator=g))
        w = h16(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
        bias = torch.randn(cout, generator=g)
        xin = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest") if up else x
        ref = O.conv_frames(xin, w, bias, stride=stride, padding=1)
        Ho, Wo = ref.shape[-2:]
        lda = cin + 8
        xd = rows_from_5d(x, lda)
        wp = torch.empty(cout, 9 * cin, dtype=torch.float16, device=DEV)
        w32 = w.to(DEV)
        hip.pack_conv3x3(w32.data_ptr(), cout, cin, cin, wp.data_ptr())
        out = torch.empty(n * Ho * Wo, cout, dtype=torch.float16, device=DEV)
        d = hip.ConvDesc(n, H, W, cin, cout, stride, up, lda, cout, 0, hip.EPI_BIAS, 1, 0, 1.0, rnd.choice([0, 0, 1, 2]))
        wsb = ws(hip.conv3x3_workspace_bytes(d))
        bd = bias.to(DEV)
        hip.conv3x3(d, xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), 0, 0, out.data_ptr(), wsb.data_ptr(), wsb.numel())
        torch.cuda.synchronize()
        try:
            close(rows_to_5d(out, 1, cout, n, Ho, Wo), ref)
        except AssertionError as e:
            raise AssertionError(f"case {case}: n={n} {H}x{W} {cin}->{cout} s={stride} up={up}: {e}")


def test_flash_random_shapes(hiplib):
    from rcdms_amd import hip
    rnd = random.Random(777)
    for case in range(20):
        batch, heads = rnd.choice([1, 2, 3]), rnd.choice([1, 2, 8])
        d = 8 * rnd.randint(1, 20)
        Lq, Lk = rnd.randint(1, 300), rnd.randint(1, 300)
        causal = rnd.random() < 0.3
        pad = rnd.random() < 0.3
        if causal:
            Lk = Lq
        g = torch.Generator().manual_seed(200 + case)
        C = heads * d
        q = h16(torch.randn(batch, Lq, C, generator=g))
        k = h16(torch.randn(batch, Lk, C, generator=g))
        v = h16(torch.randn(batch, Lk, C, generator=g))
        valid = torch.ones(batch, Lk, dtype=torch.uint8)
        if pad and Lk > 3:
            valid[:, 1 + rnd.randint(0, Lk - 3):Lk - 1] = 0       # key 0 and the last key stay visible
        mask = None
        if causal or pad:
            mask = ((1.0 - valid.float())[:, None, :] * -10000.0).expand(batch, Lq, Lk).clone()
            if causal:
                mask = mask + torch.full((Lq, Lk), -10000.0).triu_(1)[None]
        ref = O.attention_core(q, k, v, heads, mask=mask)
        qd, kd, vd = (t.reshape(-1, C).half().to(DEV) for t in (q, k, v))
        vm = valid.to(DEV)
        out = torch.empty(batch * Lq, C, dtype=torch.float16, device=DEV)
        desc = hip.AttnDesc(batch, heads, Lq, Lk, d, C, C, C, C, d ** -0.5)
        if causal or pad:
            hip.flash_attn_masked(desc, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), vm.data_ptr() if pad else 0, causal,
                                  out.data_ptr())
        else:
            hip.flash_attn(desc, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        try:
            close(out.reshape(batch, Lq, C), ref)
        except AssertionError as e:
            raise AssertionError(f"case {case}: B={batch} H={heads} Lq={Lq} Lk={Lk} d={d} causal={causal} pad={pad}: {e}")


def test_xattn_random_shapes(hiplib):
    """rcdm_xattn_pack_kv + rcdm_xattn: any head dim the flash kernel takes, 1..96 keys, ragged query counts, K and V in
    separate buffers with padded leading dimensions."""
    from rcdms_amd import hip
    rnd = random.Random(4242)
    for case in range(24):
        batch, heads = rnd.choice([1, 2, 5]), rnd.choice([1, 2, 3, 8])
        d = 8 * rnd.randint(1, 20)
        Lq, Lk = rnd.randint(1, 700), rnd.randint(1, 96)
        g = torch.Generator().manual_seed(900 + case)
        C = heads * d
        q = h16(torch.randn(batch, Lq, C, generator=g))
        k = h16(torch.randn(batch, Lk, C, generator=g) * 1.5)
        v = h16(torch.randn(batch, Lk, C, generator=g))
        ref = O.attention_core(q, k, v, heads)
        ldq, ldk, ldv, ldo = (C + 8 * rnd.randint(0, 3) for _ in range(4))

        def padded(t, rows, ld):
            buf = torch.zeros(rows, ld, dtype=torch.float16)
            buf[:, :C] = t.reshape(rows, C).half()
            return buf.to(DEV)
        qd, kd, vd = padded(q, batch * Lq, ldq), padded(k, batch * Lk, ldk), padded(v, batch * Lk, ldv)
        img = torch.empty(hip.xattn_image_bytes(batch, heads, d), dtype=torch.uint8, device=DEV)
        out = torch.full((batch * Lq, ldo), float("nan"), dtype=torch.float16, device=DEV)
        desc = hip.AttnDesc(batch, heads, Lq, Lk, d, ldq, ldk, ldv, ldo, d ** -0.5)
        hip.xattn_pack_kv(kd.data_ptr(), vd.data_ptr(), batch, Lk, heads, d, ldk, ldv, img.data_ptr())
        hip.xattn(desc, qd.data_ptr(), img.data_ptr(), out.data_ptr())
        torch.cuda.synchronize()
        try:
            close(out[:, :C].reshape(batch, Lq, C), ref)
            assert torch.isnan(out[:, C:].float()).all(), "wrote outside the C columns"
        except AssertionError as e:
            raise AssertionError(f"case {case}: B={batch} H={heads} Lq={Lq} Lk={Lk} d={d}: {e}")


def test_layernorm_random_shapes(hiplib):
    """rcdm_layernorm (row-group kernel: every lanes-per-row / chunks-per-lane combination, masked chunks, ragged last
    wave), with and without the positional-encoding table, padded leading dimensions."""
    from rcdms_amd import hip
    rnd = random.Random(99)
    for case in range(30):
        M = rnd.choice([1, 3, 31, 64, 257, 1000, 4099])
        C = 8 * rnd.randint(1, 256)
        pe = rnd.random() < 0.4
        frames, rpf = rnd.choice([1, 5, 8]), rnd.choice([1, 7, 64])
        g = torch.Generator().manual_seed(300 + case)
        x = h16(torch.randn(M, C, generator=g) * 2 + 0.3)
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        table = torch.randn(frames, C, generator=g)
        ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
        if pe:
            ref = ref + table[(torch.arange(M) // rpf) % frames]
        ldx, ldy = C + 8 * rnd.randint(0, 2), C + 8 * rnd.randint(0, 2)
        xd = torch.zeros(M, ldx, dtype=torch.float16); xd[:, :C] = x.half(); xd = xd.to(DEV)
        y = torch.full((M, ldy), float("nan"), dtype=torch.float16, device=DEV)
        gd, bd, td = gamma.to(DEV), beta.to(DEV), table.to(DEV)
        d = hip.LayerNormDesc(M, C, ldx, ldy, 1e-5, rpf, frames)
        hip.layernorm(d, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), td.data_ptr() if pe else 0, y.data_ptr())
        torch.cuda.synchronize()
        try:
            close(y[:, :C], ref)
            assert torch.isnan(y[:, C:].float()).all(), "wrote outside the C columns"
        except AssertionError as e:
            raise AssertionError(f"case {case}: M={M} C={C} pe={pe} frames={frames} rpf={rpf}: {e}")
