"""GPU parity of the row-stationary fused feed-forward (rcdm_ff_fused, rowff.hip) against the reference arithmetic
out = x + FeedForward_geglu(LayerNorm(x))  (attention.py:514 / motion_module.py:243; diffusers FeedForward "geglu":
Linear(C -> 8C), hidden * gelu(gate) with the exact erf GELU, Linear(4C -> C)) on f16-rounded inputs, and against the
unfused three-launch path of the library (rcdm_layernorm + rcdm_gemm GEGLU + rcdm_gemm bias + residual).

Tolerance: |hip - ref| <= 4e-3 * max|ref| + 2e-3 * |ref| (tests/test_hip_kernels.py)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_hip_kernels import DEV, close, h16

pytestmark = pytest.mark.gpu


def _weights(C, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(
        ln_g=1.0 + 0.2 * torch.randn(C, generator=g), ln_b=0.1 * torch.randn(C, generator=g),
        w1=h16(torch.randn(8 * C, C, generator=g) * C ** -0.5), b1=0.1 * torch.randn(8 * C, generator=g),
        w2=h16(torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5), b2=0.1 * torch.randn(C, generator=g)), g


def _reference(x, w):
    C = x.shape[1]
    a = h16(F.layer_norm(x, (C,), w["ln_g"], w["ln_b"], 1e-5))       # the MFMA operand is f16
    hg = F.linear(a, w["w1"], w["b1"])
    hidden, gate = hg.chunk(2, dim=-1)
    return x + F.linear(h16(hidden * F.gelu(gate)), w["w2"], w["b2"])


def _run_fused(x16, ld, w, out=None, M=None):
    from rcdms_amd import hip
    M, C = M or x16.shape[0], w["ln_g"].numel()
    dev = {k: v.to(DEV).contiguous() for k, v in w.items()}
    ws = torch.empty(hip.ff_stream_bytes(C), dtype=torch.uint8, device=DEV)
    b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    hip.pack_ff_stream(dev["w1"].data_ptr(), dev["b1"].data_ptr(), dev["w2"].data_ptr(), C, ws.data_ptr(), b1p.data_ptr())
    if out is None:
        out = torch.zeros_like(x16)
    d = hip.FFDesc(M, C, ld, ld, 1e-5)
    hip.ff_fused(d, x16.data_ptr(), dev["ln_g"].data_ptr(), dev["ln_b"].data_ptr(), ws.data_ptr(), b1p.data_ptr(),
                 dev["b2"].data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("M", [160, 16, 333, 2560])   # one block; one wave; ragged tail; many blocks
def test_ff_fused_vs_reference(hiplib, M):
    from rcdms_amd import hip
    C = 320
    assert hip.ff_fused_supported(C)
    w, g = _weights(C, 11 + M)
    x = h16(torch.randn(M, C, generator=g) * 1.5 + 0.3)
    ref = _reference(x, w)
    ld = C + 64                                           # rows inside a wider buffer
    xb = torch.full((M + 8, ld), 7.0, dtype=torch.float16, device=DEV)   # guard rows / columns must stay untouched
    xb[:M, :C] = x.half().to(DEV)
    out = torch.full((M + 8, ld), 7.0, dtype=torch.float16, device=DEV)
    _run_fused(xb, ld, w, out, M=M)
    close(out[:M, :C], ref)
    assert (out[M:] == 7.0).all() and (out[:, C:] == 7.0).all(), "stores outside the M x C result"


def test_ff_fused_in_place_and_vs_unfused_path(hiplib):
    """out aliasing x (how the engine calls it), and agreement with the three-launch path to f16 rounding."""
    from rcdms_amd import hip
    M, C = 1280, 320
    w, g = _weights(C, 5)
    x = h16(torch.randn(M, C, generator=g))
    dev = {k: v.to(DEV).contiguous() for k, v in w.items()}
    x16 = x.half().to(DEV)
    fused = x16.clone()
    _run_fused(fused, C, w, fused)
    # unfused: LayerNorm -> GEGLU GEMM -> out GEMM (+bias +residual)
    a = torch.empty_like(x16)
    hip.layernorm(hip.LayerNormDesc(M, C, C, C, 1e-5, 1, 1), x16.data_ptr(), dev["ln_g"].data_ptr(), dev["ln_b"].data_ptr(), 0,
                  a.data_ptr())
    wp = torch.empty(8 * C, C, dtype=torch.float16, device=DEV)
    bp = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    hip.pack_geglu_rows(dev["w1"].data_ptr(), dev["b1"].data_ptr(), 8 * C, C, wp.data_ptr(), bp.data_ptr())
    hid = torch.empty(M, 4 * C, dtype=torch.float16, device=DEV)
    wsb = torch.full((1 << 24,), 0xFF, dtype=torch.uint8, device=DEV)   # NaN-filled workspace
    hip.gemm(hip.GemmDesc(M, 8 * C, C, C, 4 * C, 0, hip.EPI_BIAS | hip.EPI_GEGLU, 1, 0, 1.0, 1), a.data_ptr(), wp.data_ptr(),
             bp.data_ptr(), 0, 0, hid.data_ptr(), wsb.data_ptr(), wsb.numel())
    w2h = dev["w2"].half().contiguous()
    unf = torch.empty_like(x16)
    hip.gemm(hip.GemmDesc(M, C, 4 * C, 4 * C, C, C, hip.EPI_BIAS | hip.EPI_RESIDUAL, 1, 0, 1.0, 1), hid.data_ptr(), w2h.data_ptr(),
             dev["b2"].data_ptr(), 0, x16.data_ptr(), unf.data_ptr(), wsb.data_ptr(), wsb.numel())
    torch.cuda.synchronize()
    close(fused, _reference(x, w))
    close(fused, unf.float(), rel=3e-3, abs_frac=3e-3)


def test_ff_fused_deterministic_and_rejects(hiplib):
    from rcdms_amd import hip
    M, C = 640, 320
    w, g = _weights(C, 3)
    x16 = h16(torch.randn(M, C, generator=g)).half().to(DEV)
    a = _run_fused(x16, C, w)
    b = _run_fused(x16, C, w)
    assert torch.equal(a, b)
    assert not hip.ff_fused_supported(64)
    with pytest.raises(hip.RcdmError):
        hip.ff_fused(hip.FFDesc(M, 64, 64, 64, 1e-5), x16.data_ptr(), x16.data_ptr(), x16.data_ptr(), x16.data_ptr(),
                     x16.data_ptr(), x16.data_ptr(), x16.data_ptr())


# ------------------------------------------------------------------------------------------------
# rcdm_rowchain: tok = a W_a^T + b_a (+ res); y = LayerNorm(tok) (+ pe); tail GEMM (q / qkv) or feed-forward

def _chain_weights(C, tail, frames, seed):
    g = torch.Generator().manual_seed(seed)
    w = dict(wa=h16(torch.randn(C, C, generator=g) * C ** -0.5), ba=0.1 * torch.randn(C, generator=g),
             ln_g=1.0 + 0.2 * torch.randn(C, generator=g), ln_b=0.1 * torch.randn(C, generator=g),
             pe=(0.3 * torch.randn(frames, C, generator=g)) if frames else None)
    if tail:
        w["wt"] = h16(torch.randn(tail * C, C, generator=g) * C ** -0.5)
    else:
        w.update(w1=h16(torch.randn(8 * C, C, generator=g) * C ** -0.5), b1=0.1 * torch.randn(8 * C, generator=g),
                 w2=h16(torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5), b2=0.1 * torch.randn(C, generator=g))
    return w, g


def _chain_reference(a, res, w, tail, rpf):
    C = a.shape[1]
    tok = F.linear(a, w["wa"], w["ba"])
    if res is not None:
        tok = tok + res
    tok = h16(tok)                                         # the token rows are stored (and chained) as f16
    y = F.layer_norm(tok, (C,), w["ln_g"], w["ln_b"], 1e-5)
    if w["pe"] is not None:
        frames = w["pe"].shape[0]
        y = y + w["pe"][(torch.arange(a.shape[0]) // rpf) % frames]
    y = h16(y)
    if tail:
        return tok, F.linear(y, w["wt"])
    hidden, gate = F.linear(y, w["w1"], w["b1"]).chunk(2, dim=-1)
    return tok, tok + F.linear(h16(hidden * F.gelu(gate)), w["w2"], w["b2"])


def _run_chain(a16, res16, w, tail, M, rpf, in_place_tok=False, gn=None):
    from rcdms_amd import hip
    C = w["ln_g"].numel()
    dev = {k: (v.to(DEV).contiguous() if v is not None else None) for k, v in w.items()}
    ws = torch.empty(hip.rowchain_stream_bytes(C, tail), dtype=torch.uint8, device=DEV)
    b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    hip.pack_rowchain(dev["wa"].data_ptr(), C, tail, dev["wt"].data_ptr() if tail else 0,
                      0 if tail else dev["w1"].data_ptr(), 0 if tail else dev["b1"].data_ptr(),
                      0 if tail else dev["w2"].data_ptr(), ws.data_ptr(), 0 if tail else b1p.data_ptr())
    ncol = tail * C if tail else C
    tok = res16 if in_place_tok else torch.full((M + 4, C + 8), 3.0, dtype=torch.float16, device=DEV)
    out = torch.full((M + 4, ncol + 8), 5.0, dtype=torch.float16, device=DEV)
    frames = w["pe"].shape[0] if w["pe"] is not None else 1
    d = hip.RowChainDesc(M, C, a16.stride(0), res16.stride(0) if res16 is not None else 0, tok.stride(0), out.stride(0), tail,
                         rpf, frames, 1e-5, gn[3] if gn else 0, gn[4] if gn else 0)
    hip.rowchain(d, a16.data_ptr(), res16.data_ptr() if res16 is not None else 0, tok.data_ptr(), dev["ba"].data_ptr(),
                 dev["ln_g"].data_ptr(), dev["ln_b"].data_ptr(), dev["pe"].data_ptr() if dev["pe"] is not None else 0,
                 ws.data_ptr(), 0 if tail else b1p.data_ptr(), 0 if tail else dev["b2"].data_ptr(), out.data_ptr(),
                 gn_stat=gn[0].data_ptr() if gn else 0, gn_g=gn[1].data_ptr() if gn else 0, gn_b=gn[2].data_ptr() if gn else 0)
    torch.cuda.synchronize()
    return tok, out, ncol


@pytest.mark.parametrize("tail,has_res,frames,M", [
    (3, False, 0, 320),     # proj_in -> norm1 -> qkv (spatial)
    (3, False, 5, 640),     # proj_in -> norms[0] + pe -> qkv (motion; rows_per_frame 64: frames change inside a block)
    (3, True, 5, 333),      # to_out + res -> norms[1] + pe -> qkv, ragged tail
    (1, True, 0, 480),      # attn1.to_out + res -> norm2 -> attn2.to_q
    (0, True, 0, 352),      # attn2.to_out + res -> norm3 -> ff -> + res
    (0, False, 0, 160),
])
def test_rowchain_vs_reference(hiplib, tail, has_res, frames, M):
    from rcdms_amd import hip
    C, rpf = 320, 64
    assert hip.rowchain_supported(C)
    w, g = _chain_weights(C, tail, frames, 100 + 7 * tail + M)
    a = h16(torch.randn(M, C, generator=g))
    res = h16(torch.randn(M, C, generator=g) * 1.5) if has_res else None
    tok_ref, out_ref = _chain_reference(a, res, w, tail, rpf)
    a16 = a.half().to(DEV)
    res16 = res.half().to(DEV) if has_res else None
    tok, out, ncol = _run_chain(a16, res16, w, tail, M, rpf)
    close(tok[:M, :C], tok_ref)
    close(out[:M, :ncol], out_ref)
    assert (tok[M:] == 3.0).all() and (tok[:, C:] == 3.0).all(), "token stores outside M x C"
    assert (out[M:] == 5.0).all() and (out[:, ncol:] == 5.0).all(), "tail stores outside the result"


def test_rowchain_in_place_token_rows(hiplib):
    """tok aliasing res (how the engine updates the residual stream), feed-forward tail written over tok as well."""
    M, C = 800, 320
    w, g = _chain_weights(C, 0, 0, 77)
    a = h16(torch.randn(M, C, generator=g))
    res = h16(torch.randn(M, C, generator=g))
    tok_ref, out_ref = _chain_reference(a, res, w, 0, 1)
    from rcdms_amd import hip
    dev = {k: (v.to(DEV).contiguous() if v is not None else None) for k, v in w.items()}
    ws = torch.empty(hip.rowchain_stream_bytes(C, 0), dtype=torch.uint8, device=DEV)
    b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    hip.pack_rowchain(dev["wa"].data_ptr(), C, 0, 0, dev["w1"].data_ptr(), dev["b1"].data_ptr(), dev["w2"].data_ptr(),
                      ws.data_ptr(), b1p.data_ptr())
    tokb = res.half().to(DEV)
    a16 = a.half().to(DEV)
    d = hip.RowChainDesc(M, C, C, C, C, C, 0, 1, 1, 1e-5)
    hip.rowchain(d, a16.data_ptr(), tokb.data_ptr(), tokb.data_ptr(), dev["ba"].data_ptr(), dev["ln_g"].data_ptr(),
                 dev["ln_b"].data_ptr(), 0, ws.data_ptr(), b1p.data_ptr(), dev["b2"].data_ptr(), tokb.data_ptr())
    torch.cuda.synchronize()
    close(tokb, out_ref)


@pytest.mark.parametrize("frames,rows,samples", [(0, 176, 3), (5, 160, 5), (0, 1024, 2)])
def test_rowchain_groupnorm_prologue(hiplib, frames, rows, samples):
    """GroupNorm -> proj_in -> norm -> qkv with the norm's APPLY inside the chain launch (attention.py:328-330,
    motion_module.py:162-166): against the reference arithmetic, and bit-identical to rcdm_groupnorm_silu followed by the
    chain without the prologue.  rows = 176: blocks of 160 rows meet two samples, also in the middle of the tensor."""
    from rcdms_amd import hip
    C, G, M = 320, 32, rows * samples
    w, g = _chain_weights(C, 3, frames, 900 + rows)
    x = h16(torch.randn(M, C, generator=g) * 2.0 + 0.5)
    gn_g = (1.0 + 0.2 * torch.randn(C, generator=g)).float()
    gn_b = (0.3 * torch.randn(C, generator=g)).float()
    xs = x.view(samples, rows, G, C // G)
    mean = xs.mean(dim=(1, 3), keepdim=True)
    var = xs.var(dim=(1, 3), unbiased=False, keepdim=True)
    a_ref = h16((((xs - mean) / torch.sqrt(var + 1e-6)).reshape(M, C) * gn_g + gn_b))
    tok_ref, out_ref = _chain_reference(a_ref, None, w, 3, rows)

    x16 = x.half().to(DEV)
    gd = hip.GroupNormDesc(samples, rows, C, G, C, C, 1e-6, 0)
    wsb = hip.groupnorm_workspace_bytes(gd)
    gws = torch.full((max(wsb, 16),), 0xFF, dtype=torch.uint8, device=DEV)
    gg, gb = gn_g.to(DEV), gn_b.to(DEV)
    # (1) separate launches
    a16 = torch.empty_like(x16)
    hip.groupnorm_silu(gd, x16.data_ptr(), gg.data_ptr(), gb.data_ptr(), a16.data_ptr(), gws.data_ptr(), gws.numel())
    tok1, out1, ncol = _run_chain(a16, None, w, 3, M, rows)
    # (2) statistics only + the apply in the chain's prologue
    stat = torch.zeros(samples * G * 2, dtype=torch.float32, device=DEV)
    hip.groupnorm_stats(gd, x16.data_ptr(), stat.data_ptr(), gws.data_ptr(), gws.numel())
    tok2, out2, _ = _run_chain(x16, None, w, 3, M, rows, gn=(stat, gg, gb, G, rows))
    close(tok2[:M, :C], tok_ref)
    close(out2[:M, :ncol], out_ref)
    bad = (tok1 != tok2).any(dim=1).nonzero().flatten().tolist()
    assert not bad, f"prologue GroupNorm differs from the separate launch in token rows {bad[:8]} ... ({len(bad)} rows)"
    assert torch.equal(out1, out2)
    # rejected: with a stage-A residual, or with samples that are not whole 16-row fragments
    d = hip.RowChainDesc(M, C, C, C, C, 3 * C, 3, rows, max(frames, 1), 1e-5, G, rows)
    with pytest.raises(hip.RcdmError):
        hip.rowchain(d, x16.data_ptr(), x16.data_ptr(), tok2.data_ptr(), gg.data_ptr(), gg.data_ptr(), gb.data_ptr(), 0,
                     x16.data_ptr(), 0, 0, out2.data_ptr(), gn_stat=stat.data_ptr(), gn_g=gg.data_ptr(), gn_b=gb.data_ptr())
    d.gn_rows = rows + 8
    with pytest.raises(hip.RcdmError):
        hip.rowchain(d, x16.data_ptr(), 0, tok2.data_ptr(), gg.data_ptr(), gg.data_ptr(), gb.data_ptr(), 0,
                     x16.data_ptr(), 0, 0, out2.data_ptr(), gn_stat=stat.data_ptr(), gn_g=gg.data_ptr(), gn_b=gb.data_ptr())


@pytest.mark.parametrize("M", [352, 2560])
def test_rowchain_ff_then_projection(hiplib, M):
    """tail 2: attn.to_out + res -> norm -> ff -> + res -> proj_out + bias -> + the block's input, one launch
    (attention.py:505-514,352-363).  Against the reference arithmetic and against tail 0 followed by rcdm_gemm."""
    from rcdms_amd import hip
    C = 320
    w, g = _chain_weights(C, 0, 0, 4242 + M)
    wz = h16(torch.randn(C, C, generator=g) * C ** -0.5)
    bz = 0.1 * torch.randn(C, generator=g)
    a = h16(torch.randn(M, C, generator=g))
    res = h16(torch.randn(M, C, generator=g) * 1.5)
    zres = h16(torch.randn(M, C, generator=g))
    tok_ref, ff_ref = _chain_reference(a, res, w, 0, 1)
    out_ref = h16(F.linear(h16(ff_ref), wz, bz)) + zres

    dev = {k: (v.to(DEV).contiguous() if v is not None else None) for k, v in w.items()}
    wzd, bzd = wz.to(DEV).contiguous(), bz.to(DEV)
    ws = torch.empty(hip.rowchain_stream_bytes(C, 2), dtype=torch.uint8, device=DEV)
    assert ws.numel() == 2 * C * C * 14
    b1p = torch.empty(8 * C, dtype=torch.float32, device=DEV)
    hip.pack_rowchain(dev["wa"].data_ptr(), C, 2, wzd.data_ptr(), dev["w1"].data_ptr(), dev["b1"].data_ptr(),
                      dev["w2"].data_ptr(), ws.data_ptr(), b1p.data_ptr())
    a16, z16 = a.half().to(DEV), zres.half().to(DEV)
    tok = res.half().to(DEV)                         # in place on the residual stream, as the engine runs it
    out = torch.full((M + 4, C + 8), 5.0, dtype=torch.float16, device=DEV)
    d = hip.RowChainDesc(M, C, C, C, C, out.stride(0), 2, 1, 1, 1e-5, 0, 0, C)
    args = (a16.data_ptr(), tok.data_ptr(), tok.data_ptr(), dev["ba"].data_ptr(), dev["ln_g"].data_ptr(), dev["ln_b"].data_ptr(), 0,
            ws.data_ptr(), b1p.data_ptr(), dev["b2"].data_ptr(), out.data_ptr())
    hip.rowchain(d, *args, z_res=z16.data_ptr(), z_bias=bzd.data_ptr())
    torch.cuda.synchronize()
    close(tok, tok_ref)                              # tok keeps stage A's rows (the feed-forward output is not stored)
    close(out[:M, :C], out_ref)
    assert (out[M:] == 5.0).all() and (out[:, C:] == 5.0).all(), "stores outside the result"
    # the two-launch form: tail 0 in place, then proj_out as rcdm_gemm with bias + residual
    _, ff2, _ = _run_chain(a16, res.half().to(DEV), w, 0, M, 1)
    out2 = torch.empty(M, C, dtype=torch.float16, device=DEV)
    gd = hip.GemmDesc(M, C, C, ff2.stride(0), C, C, 1 | 4, 1, 0, 1.0, 0, 0)
    wsz = torch.full((max(hip.gemm_workspace_bytes(gd), 16),), 0xFF, dtype=torch.uint8, device=DEV)
    hip.gemm(gd, ff2.data_ptr(), wzd.half().data_ptr(), bzd.data_ptr(), 0, z16.data_ptr(), out2.data_ptr(), wsz.data_ptr(), wsz.numel())
    torch.cuda.synchronize()
    close(out[:M, :C], out2.float().cpu())
    # rejected without the residual / bias of the projection, or without a stage-A residual
    with pytest.raises(hip.RcdmError):
        hip.rowchain(d, *args)
    with pytest.raises(hip.RcdmError):
        hip.rowchain(d, a16.data_ptr(), 0, *args[2:], z_res=z16.data_ptr(), z_bias=bzd.data_ptr())
