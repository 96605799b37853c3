"""GPU parity tests of the Winograd F(2x2, 3x3) form of the stride-1 3x3 convolution (rcdms_amd/csrc/wino.hip, round 6):
rcdm_conv3x3_wino against the oracle's fp32 convolution (oracle/unet_oracle.py conv_frames = InflatedConv3d.forward,
src/models/resnet.py:10-18) on the same seeded f16-rounded inputs, next to the library's own nine-tap form.

Tolerance: the kernel-level tolerance of tests/test_hip_kernels.py (|hip - oracle| <= 4e-3 max|oracle| + 2e-3 |oracle|); on
top of that the Winograd form's rel-RMS error against the oracle may be at most 3x the nine-tap form's on the same operands
(measured ~1.3-2x: its MFMA operands are f16(B^T d B) and f16(G g G^T) instead of the f16 pixels and weights)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O
from tests.test_hip_kernels import DEV, close, h16, rows_from_5d, rows_to_5d, ws

pytestmark = pytest.mark.gpu

G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=np.float64)


def test_wino_pack(hiplib):
    """rcdm_pack_conv3x3_wino against the definition U[4 i + j][co][ci] = (G g G^T)[i][j]."""
    from rcdms_amd import hip
    g = torch.Generator().manual_seed(5)
    w = torch.randn(24, 64, 3, 3, generator=g)
    U = torch.empty(16, 24, 64, dtype=torch.float16, device=DEV)
    wd = w.to(DEV)
    hip.pack_conv3x3_wino(wd.data_ptr(), 24, 64, U.data_ptr())
    torch.cuda.synchronize()
    want = np.einsum("ik,ockl,jl->ijoc", G, w.double().numpy(), G).reshape(16, 24, 64)
    got = U.float().cpu().numpy()
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()


def _rel_rms(got, ref):
    return ((got - ref).double().pow(2).mean().sqrt() / ref.double().pow(2).mean().sqrt()).item()


@pytest.mark.parametrize("b,f,H,W,cin,cin2,cout,epi,split,gn", [
    (1, 2, 4, 4, 64, 0, 64, 0, 0, False),          # 8 tiles: one partly filled row tile, no epilogue
    (2, 1, 8, 8, 128, 0, 192, 1 | 2, 0, False),    # bias + per-sample row vector (conv1 of a ResNet block), N tail of a tile
    (2, 1, 8, 8, 128, 0, 192, 1 | 4, 3, False),    # bias + residual (conv2), forced split-K with an uneven last slice (2 k-steps / 3)
    (2, 5, 6, 10, 64, 0, 64, 1, 2, False),         # non-square image
    (2, 1, 8, 8, 64, 128, 64, 1 | 4, 0, False),    # second input: the 1x1 conv_shortcut's four parity entries
    (2, 1, 8, 8, 128, 64, 320, 1 | 2 | 4, 2, False),
    (2, 5, 8, 8, 128, 0, 64, 1 | 2, 0, True),      # GroupNorm + SiLU applied by the input transform (cross-frame statistics)
    (2, 5, 16, 16, 1280, 0, 1280, 1 | 2, 0, True),   # the 16x16-level conv1 of the UNet
    (2, 5, 16, 16, 1280, 2560, 1280, 1, 0, True),    # ... conv2 + shortcut of the 2560-channel concat
    (2, 5, 8, 8, 1280, 0, 1280, 1 | 4, 0, False),    # the 8x8 level: 160 tiles, heuristic split-K
    (2, 5, 8, 8, 2560, 0, 1280, 1 | 2, 0, True),
])
@pytest.mark.parametrize("slab16", [0, 1])
def test_conv3x3_wino(hiplib, b, f, H, W, cin, cin2, cout, epi, split, gn, slab16):
    """slab16: rcdm_set_wino_slab_f16 — the transform-domain sums between the batched GEMM and the output transform in f16."""
    from rcdms_amd import hip
    hip.set_wino_slab_f16(slab16)
    try:
        _conv3x3_wino(hip, b, f, H, W, cin, cin2, cout, epi, split, gn, slab16)
    finally:
        hip.set_wino_slab_f16(-1)


def _conv3x3_wino(hip, b, f, H, W, cin, cin2, cout, epi, split, gn, slab16):
    g = torch.Generator().manual_seed(900 + cin + cin2 + cout + H)
    x = h16(torch.randn(b, cin, f, H, W, generator=g) * 1.5 + 0.3)
    w = h16(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
    bias = torch.randn(cout, generator=g)
    temb = torch.randn(b, cout, generator=g)
    res = h16(torch.randn(b, cout, f, H, W, generator=g))
    gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
    xin = F.silu(O.group_norm_cross_frame(x, gamma, beta, 32, 1e-5)) if gn else x
    ref = O.conv_frames(xin, w, bias if epi & 1 else None, stride=1, padding=1)
    x2 = w2 = None
    if cin2:
        x2 = h16(torch.randn(b, cin2, f, H, W, generator=g))
        w2 = h16(torch.randn(cout, cin2, generator=g) * cin2 ** -0.5)
        ref = ref + torch.einsum("oc,bcfhw->bofhw", w2, x2)
    if epi & 2:
        ref = ref + temb[:, :, None, None, None]
    if epi & 4:
        ref = ref + res
    ref = ref * 0.5
    lda, lda2 = cin + 8, cin2 + 16
    xd, rd = rows_from_5d(x, lda), rows_from_5d(res)
    x2d = rows_from_5d(x2, lda2) if cin2 else None
    w2d = w2.half().to(DEV).contiguous() if cin2 else None
    w32 = w.to(DEV)
    U = torch.empty(16, cout, cin, dtype=torch.float16, device=DEV)
    hip.pack_conv3x3_wino(w32.data_ptr(), cout, cin, U.data_ptr())
    bd, td, gd, bed = bias.to(DEV), temb.to(DEV), gamma.to(DEV), beta.to(DEV)
    gd2, bd2 = (torch.rand(cout, generator=g) + 0.5).to(DEV), (torch.randn(cout, generator=g) * 0.3).to(DEV)
    M = b * f * H * W
    out = torch.full((M, cout), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.ConvDesc(b * f, H, W, cin, cout, 1, 0, lda, cout, cout if epi & 4 else 0, epi, f * H * W, cout, 0.5, split, 0, 0, cin2,
                     lda2 if cin2 else 0)
    assert hip.conv3x3_wino_supported(d)
    gnd, stat = None, None
    if gn:
        gnd = hip.GroupNormDesc(b, f * H * W, cin, 32, lda, lda, 1e-5, 1)
        stat = torch.empty(b * 32 * 2, dtype=torch.float32, device=DEV)
        gws = ws(hip.groupnorm_workspace_bytes(gnd))
        hip.groupnorm_stats(gnd, xd.data_ptr(), stat.data_ptr(), gws.data_ptr(), gws.numel())
    wsb = ws(hip.conv3x3_wino_workspace_bytes(d))
    # the statistics of the GroupNorm that reads `out` next (cross-frame, 32 groups), left by the output transform per tile
    god = hip.GroupNormDesc(b, f * H * W, cout, 32, cout, cout, 1e-5, 0) if cout % 32 == 0 else None
    part = ws(b * 32 * (f * H * W // 4) * 3 * 4) if god is not None else None
    hip.conv3x3_wino(d, xd.data_ptr(), U.data_ptr(), bd.data_ptr() if epi & 1 else 0, td.data_ptr() if epi & 2 else 0,
                     rd.data_ptr() if epi & 4 else 0, out.data_ptr(), wsb.data_ptr(), wsb.numel(),
                     x2=x2d.data_ptr() if cin2 else 0, W2=w2d.data_ptr() if cin2 else 0, gn=gnd,
                     gn_stat=stat.data_ptr() if gn else 0, gn_gamma=gd.data_ptr() if gn else 0, gn_beta=bed.data_ptr() if gn else 0,
                     gn_out=god, gn_out_partial=part.data_ptr() if god is not None else 0)
    torch.cuda.synchronize()
    got = rows_to_5d(out, b, cout, f, H, W)
    close(got, ref)
    if god is not None:
        st_t = torch.full((b * 32 * 2,), float("nan"), dtype=torch.float32, device=DEV)
        hip.groupnorm_finalize(b, 32, f * H * W // 4, 1e-5, part.data_ptr(), st_t.data_ptr())
        st_r = torch.empty_like(st_t)
        gws2 = ws(hip.groupnorm_workspace_bytes(god))
        hip.groupnorm_stats(god, out.data_ptr(), st_r.data_ptr(), gws2.data_ptr(), gws2.numel())
        torch.cuda.synchronize()
        assert torch.allclose(st_t, st_r, rtol=2e-5, atol=1e-6), (st_t - st_r).abs().max()
        # ... and for a PER-FRAME norm behind the conv (attention.py:328 / motion_module.py:162: samples = images): same launch
        gof = hip.GroupNormDesc(b * f, H * W, cout, 32, cout, cout, 1e-6, 0)
        partf = ws(b * f * 32 * (H * W // 4) * 3 * 4)
        out.fill_(float("nan"))
        hip.conv3x3_wino(d, xd.data_ptr(), U.data_ptr(), bd.data_ptr() if epi & 1 else 0, td.data_ptr() if epi & 2 else 0,
                         rd.data_ptr() if epi & 4 else 0, out.data_ptr(), wsb.data_ptr(), wsb.numel(),
                         x2=x2d.data_ptr() if cin2 else 0, W2=w2d.data_ptr() if cin2 else 0, gn=gnd,
                         gn_stat=stat.data_ptr() if gn else 0, gn_gamma=gd.data_ptr() if gn else 0, gn_beta=bed.data_ptr() if gn else 0,
                         gn_out=gof, gn_out_partial=partf.data_ptr())
        stf_t = torch.full((b * f * 32 * 2,), float("nan"), dtype=torch.float32, device=DEV)
        hip.groupnorm_finalize(b * f, 32, H * W // 4, 1e-6, partf.data_ptr(), stf_t.data_ptr())
        stf_r = torch.empty_like(stf_t)
        gws3 = ws(hip.groupnorm_workspace_bytes(gof))
        hip.groupnorm_stats(gof, out.data_ptr(), stf_r.data_ptr(), gws3.data_ptr(), gws3.numel())
        # finalize + apply from those statistics == the norm's own three / single launch form
        yn, yr = torch.empty_like(out), torch.empty_like(out)
        hip.groupnorm_apply(gof, out.data_ptr(), stf_t.data_ptr(), gd2.data_ptr(), bd2.data_ptr(), yn.data_ptr())
        gws4 = ws(hip.groupnorm_workspace_bytes(gof))
        hip.groupnorm_silu(gof, out.data_ptr(), gd2.data_ptr(), bd2.data_ptr(), yr.data_ptr(), gws4.data_ptr(), gws4.numel())
        torch.cuda.synchronize()
        assert torch.allclose(stf_t, stf_r, rtol=2e-5, atol=1e-6), (stf_t - stf_r).abs().max()
        assert (yn.float() - yr.float()).abs().max().item() <= 2e-3 * yr.float().abs().max().item() + 1e-3
    # the library's nine-tap form on the same operands (the norm + activation applied by its own launch)
    a1 = xd
    if gn:
        a1 = torch.empty_like(xd)
        gws = ws(hip.groupnorm_workspace_bytes(gnd))
        hip.groupnorm_silu(gnd, xd.data_ptr(), gd.data_ptr(), bed.data_ptr(), a1.data_ptr(), gws.data_ptr(), gws.numel())
    wp = torch.empty(cout, 9 * cin, dtype=torch.float16, device=DEV)
    hip.pack_conv3x3(w32.data_ptr(), cout, cin, cin, wp.data_ptr())
    out1 = torch.full((M, cout), float("nan"), dtype=torch.float16, device=DEV)
    d1 = hip.ConvDesc(b * f, H, W, cin, cout, 1, 0, lda, cout, cout if epi & 4 else 0, epi, f * H * W, cout, 0.5, 0, 0, 0, cin2,
                      lda2 if cin2 else 0)
    w1 = ws(hip.conv3x3_workspace_bytes(d1))
    args = (bd.data_ptr() if epi & 1 else 0, td.data_ptr() if epi & 2 else 0, rd.data_ptr() if epi & 4 else 0, out1.data_ptr(),
            w1.data_ptr(), w1.numel())
    if cin2:
        wk = torch.cat([wp, w2d], dim=1).contiguous()
        hip.conv3x3_add1x1(d1, a1.data_ptr(), x2d.data_ptr(), wk.data_ptr(), *args)
    else:
        hip.conv3x3(d1, a1.data_ptr(), wp.data_ptr(), *args)
    torch.cuda.synchronize()
    e_w, e_d = _rel_rms(got, ref), _rel_rms(rows_to_5d(out1, b, cout, f, H, W), ref)
    print(f"rel-RMS vs fp32 oracle: winograd{' (f16 slabs)' if slab16 else ''} {e_w:.3e}, nine-tap {e_d:.3e}, ratio {e_w / max(e_d, 1e-12):.2f}")
    assert e_w <= 3.0 * e_d + 1e-4, (e_w, e_d)


def test_conv3x3_wino_refusals(hiplib):
    """Stride 2, upsampling, odd image sizes, channel counts that are not whole k-steps, GEGLU-type epilogues: not this form."""
    from rcdms_amd import hip
    for bad in (hip.ConvDesc(2, 8, 8, 64, 64, 2, 0, 64, 64, 0, 0, 1, 0, 1.0, 0),
                hip.ConvDesc(2, 8, 8, 64, 64, 1, 1, 64, 64, 0, 0, 1, 0, 1.0, 0),
                hip.ConvDesc(2, 7, 8, 64, 64, 1, 0, 64, 64, 0, 0, 1, 0, 1.0, 0),
                hip.ConvDesc(2, 8, 8, 72, 64, 1, 0, 72, 64, 0, 0, 1, 0, 1.0, 0),
                hip.ConvDesc(2, 8, 8, 64, 64, 1, 0, 64, 64, 0, 8, 1, 0, 1.0, 0),
                hip.ConvDesc(2, 8, 8, 64, 64, 1, 0, 64, 64, 0, 0, 1, 0, 1.0, 0, 0, 16)):
        assert not hip.conv3x3_wino_supported(bad)
        assert hip.conv3x3_wino_workspace_bytes(bad) == 0
    ok = hip.ConvDesc(2, 8, 8, 64, 64, 1, 0, 64, 64, 0, 0, 1, 0, 1.0, 0)
    assert hip.conv3x3_wino_supported(ok)
    x = torch.zeros(4096, dtype=torch.float16, device=DEV)
    with pytest.raises(hip.RcdmError):   # workspace too small
        hip.conv3x3_wino(ok, x.data_ptr(), x.data_ptr(), 0, 0, 0, x.data_ptr(), x.data_ptr(), 16)


def test_conv3x3_wino_random_shapes(hiplib):
    """Seeded sweep: image counts and sides that leave partly filled 160-row tiles, channel counts with N tails, padded row
    strides, every epilogue combination, forced split-K, second input, the norm in the input transform — against fp32."""
    import random
    from rcdms_amd import hip
    rnd = random.Random(606)
    for case in range(24):
        b, f = rnd.choice([(1, 1), (1, 3), (2, 1), (2, 5), (3, 2)])
        H, W = 2 * rnd.randint(1, 6), 2 * rnd.randint(1, 6)
        cin, cout = 64 * rnd.randint(1, 4), 8 * rnd.randint(1, 40)
        cin2 = rnd.choice([0, 0, 64, 128])
        gn = rnd.random() < 0.5
        epi = rnd.choice([0, 1]) | rnd.choice([0, 2]) | rnd.choice([0, 4])
        split = rnd.choice([0, 0, 1, 2, 3])
        scale = rnd.choice([1.0, 0.5, 1 / 1.3])
        g = torch.Generator().manual_seed(7000 + case)
        x = h16(torch.randn(b, cin, f, H, W, generator=g) * 1.3 - 0.2)
        w = h16(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5)
        bias, temb = torch.randn(cout, generator=g), torch.randn(b, cout, generator=g)
        res = h16(torch.randn(b, cout, f, H, W, generator=g))
        gamma, beta = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g) * 0.3
        xin = F.silu(O.group_norm_cross_frame(x, gamma, beta, 32, 1e-5)) if gn else x
        ref = O.conv_frames(xin, w, bias if epi & 1 else None, stride=1, padding=1)
        x2d = w2d = None
        lda, lda2, ldc, ldr = cin + 8 * rnd.randint(0, 2), cin2 + 8 * rnd.randint(0, 2), cout + 8 * rnd.randint(0, 2), cout + 8 * rnd.randint(0, 2)
        if cin2:
            x2 = h16(torch.randn(b, cin2, f, H, W, generator=g))
            w2 = h16(torch.randn(cout, cin2, generator=g) * cin2 ** -0.5)
            ref = ref + torch.einsum("oc,bcfhw->bofhw", w2, x2)
            x2d, w2d = rows_from_5d(x2, lda2), w2.half().to(DEV).contiguous()
        if epi & 2:
            ref = ref + temb[:, :, None, None, None]
        if epi & 4:
            ref = ref + res
        ref = ref * scale
        xd, rd = rows_from_5d(x, lda), rows_from_5d(res, ldr)
        w32 = w.to(DEV)
        U = torch.empty(16, cout, cin, dtype=torch.float16, device=DEV)
        hip.pack_conv3x3_wino(w32.data_ptr(), cout, cin, U.data_ptr())
        bd, td, gd, bed = bias.to(DEV), temb.to(DEV), gamma.to(DEV), beta.to(DEV)
        M = b * f * H * W
        out = torch.full((M, ldc), float("nan"), dtype=torch.float16, device=DEV)
        d = hip.ConvDesc(b * f, H, W, cin, cout, 1, 0, lda, ldc, ldr if epi & 4 else 0, epi, f * H * W, cout, scale, split, 0, 0, cin2,
                         lda2 if cin2 else 0)
        assert hip.conv3x3_wino_supported(d), (case, b, f, H, W, cin, cout)
        gnd = stat = None
        if gn:
            gnd = hip.GroupNormDesc(b, f * H * W, cin, 32, lda, lda, 1e-5, 1)
            stat = torch.empty(b * 32 * 2, dtype=torch.float32, device=DEV)
            gws = ws(hip.groupnorm_workspace_bytes(gnd))
            hip.groupnorm_stats(gnd, xd.data_ptr(), stat.data_ptr(), gws.data_ptr(), gws.numel())
        wsb = ws(hip.conv3x3_wino_workspace_bytes(d))
        hip.conv3x3_wino(d, xd.data_ptr(), U.data_ptr(), bd.data_ptr() if epi & 1 else 0, td.data_ptr() if epi & 2 else 0,
                         rd.data_ptr() if epi & 4 else 0, out.data_ptr(), wsb.data_ptr(), wsb.numel(),
                         x2=x2d.data_ptr() if cin2 else 0, W2=w2d.data_ptr() if cin2 else 0, gn=gnd,
                         gn_stat=stat.data_ptr() if gn else 0, gn_gamma=gd.data_ptr() if gn else 0, gn_beta=bed.data_ptr() if gn else 0)
        torch.cuda.synchronize()
        assert torch.isnan(out[:, cout:]).all() if ldc > cout else True, "the pad columns of the output rows were written"
        close(rows_to_5d(out, b, cout, f, H, W), ref, rel=3e-3, abs_frac=6e-3)


@pytest.mark.parametrize("n_img,H,W,c", [(3, 4, 6, 64), (10, 8, 8, 640), (10, 16, 16, 1280)])
def test_upsample_tap_planes(hiplib, n_img, H, W, c):
    """Upsample3D (resnet.py:60-79) as one GEMM over the SOURCE pixels with the nine taps' weights stacked (N = 9 c) + the gather
    rcdm_upsample_taps_gather, against F.interpolate(nearest, x2) + conv3x3 in fp32 and next to the four-phase form."""
    from rcdms_amd import hip
    g = torch.Generator().manual_seed(40 + c + H)
    x = h16(torch.randn(1, c, n_img, H, W, generator=g))
    w = h16(torch.randn(c, c, 3, 3, generator=g) * (9 * c) ** -0.5)
    bias = torch.randn(c, generator=g)
    ref = O.conv_frames(F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest"), w, bias, stride=1, padding=1)
    lda = c + 8
    xd = rows_from_5d(x, lda)
    W9 = w.permute(2, 3, 0, 1).reshape(9 * c, c).half().to(DEV).contiguous()
    M = n_img * H * W
    ldp = 9 * c + 8
    P = torch.full((M, ldp), float("nan"), dtype=torch.float16, device=DEV)
    d = hip.GemmDesc(M, 9 * c, c, lda, ldp, 0, 0, 1, 0, 1.0, 0, 0)
    wsb = ws(hip.gemm_workspace_bytes(d))
    hip.gemm(d, xd.data_ptr(), W9.data_ptr(), 0, 0, 0, P.data_ptr(), wsb.data_ptr(), wsb.numel())
    out = torch.full((4 * M, c), float("nan"), dtype=torch.float16, device=DEV)
    bd = bias.to(DEV)
    hip.upsample_taps_gather(P.data_ptr(), ldp, n_img, H, W, c, bd.data_ptr(), out.data_ptr(), c)
    torch.cuda.synchronize()
    got = rows_to_5d(out, 1, c, n_img, 2 * H, 2 * W)
    close(got, ref)
    if c % 64 == 0 and c >= 640:
        d2 = hip.ConvDesc(n_img, H, W, c, c, 1, 2, lda, c, 0, hip.EPI_BIAS, 1, 0, 1.0, 0)
        if hip.conv3x3_up2_supported(d2):
            wp2 = torch.empty(4, c, 4 * c, dtype=torch.float16, device=DEV)
            w32 = w.to(DEV)
            hip.pack_conv3x3_up2(w32.data_ptr(), c, c, wp2.data_ptr())
            out2 = torch.full((4 * M, c), float("nan"), dtype=torch.float16, device=DEV)
            w2 = ws(hip.conv3x3_workspace_bytes(d2))
            hip.conv3x3(d2, xd.data_ptr(), wp2.data_ptr(), bd.data_ptr(), 0, 0, out2.data_ptr(), w2.data_ptr(), w2.numel())
            torch.cuda.synchronize()
            e9, e4 = _rel_rms(got, ref), _rel_rms(rows_to_5d(out2, 1, c, n_img, 2 * H, 2 * W), ref)
            print(f"rel-RMS vs fp32: tap planes {e9:.3e}, four phases {e4:.3e}")
            assert e9 <= 3.0 * e4 + 1e-4
