"""GPU parity tests of the deferred LayerNorm (rcdm_gemm_lnx): the GEMM that writes the token rows emits per-row partial
(sum, sum of squares); the GEMM behind the LayerNorm takes the RAW rows with gamma / beta folded into its weights and
applies rstd (x W'^T) - rstd mean S + b' in its epilogue.  Reference (fp32 torch, the arithmetic of the reference's
nn.LayerNorm + nn.Linear, attention.py:482-514 / motion_module.py:236-243 with PositionalEncoding :265-267):
    x = f16(A_p W_p^T + b_p + res);  y = epi(LayerNorm(x) (+ pe_f) W^T + b).
Every tile shape (variants 1-10 as consumer — 8, the 256x256 ping-pong tile, refuses —, 1-5 and 10 as producer), plain / GEGLU / row-vector epilogues, ragged M, dup rows,
rows whose mean is large against their spread (the cancellation case of E[x^2] - mean^2 and of the f16-staged x W'^T)."""
import pytest
import torch

from tests.test_hip_kernels import DEV, close, h16, ws

pytestmark = pytest.mark.gpu


def _fold(W, gamma, beta, bias):
    """Packer.lnx_mat's arithmetic on the host: f16(W diag(gamma)), its row sums, bias + W beta."""
    Wg = (W * gamma[None, :]).half()
    return Wg, Wg.float().sum(dim=1), (W * beta[None, :]).sum(dim=1) + (bias if bias is not None else 0)


def _producer(hip, Ap, Wp, bp, res, M, C, Kp, variant, dup=0):
    """x = f16(Ap Wp^T + bp + res) through rcdm_gemm_lnx with statistics; returns (x rows on the device [M(+dup)][C], stat, parts)."""
    hip.set_igemm_variant(variant)
    d = hip.GemmDesc(M, C, Kp, Kp, C, C, 1 | 4, 1, 0, 1.0, 1, dup)
    parts = hip.gemm_stat_parts(d)
    assert parts > 0
    x = torch.full((M + dup, C), float("nan"), dtype=torch.float16, device=DEV)
    stat = torch.full(((M + dup) * parts * 2,), float("nan"), dtype=torch.float32, device=DEV)
    lx = hip.Lnx(stat.data_ptr(), parts, M + dup, 0, 0, 0, 0, 1e-5, 0)
    Ad, Wd, bd, Rd = Ap.half().to(DEV), Wp.half().to(DEV), bp.to(DEV), res.half().to(DEV)
    hip.gemm_lnx(d, lx, Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), 0, Rd.data_ptr(), x.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    hip.set_igemm_variant(-1)
    return x, stat, parts


@pytest.mark.parametrize("pvariant", [1, 2, 3, 4, 5, 10])
@pytest.mark.parametrize("M,C", [(640, 1280), (1000, 640), (300, 64)])
def test_row_statistics(hiplib, M, C, pvariant):
    """Producer side alone: the partial slots of a row sum to (sum, sum of squares) of the f16 values stored to that row."""
    from rcdms_amd import hip
    g = torch.Generator().manual_seed(5 + M + C + pvariant)
    Kp = 128
    Ap = h16(torch.randn(M, Kp, generator=g))
    Wp = h16(torch.randn(C, Kp, generator=g) * Kp ** -0.5)
    bp = torch.randn(C, generator=g)
    res = h16(torch.randn(M, C, generator=g) * 3 + 2.0)
    x, stat, parts = _producer(hip, Ap, Wp, bp, res, M, C, Kp, pvariant)
    xs = x.float().cpu()
    ref = h16(Ap @ Wp.t() + bp + res)
    close(xs, ref)
    st = stat.cpu().view(parts, M, 2).sum(dim=0)
    assert torch.isfinite(st).all(), "a statistics slot was not written"
    r1, r2 = xs.double().sum(dim=1), (xs.double() ** 2).sum(dim=1)
    e1 = ((st[:, 0].double() - r1).abs() / (xs.double().abs().sum(dim=1) + 1e-6)).max().item()
    e2 = ((st[:, 1].double() - r2).abs() / (r2 + 1e-6)).max().item()
    print(f"row statistics vs float64: sum {e1:.2e} (of sum |x|), sum of squares {e2:.2e} relative")
    assert e1 < 2e-6 and e2 < 2e-6      # fp32 accumulation of exact f16 products (v_dot2c_f32_f16), a few hundred terms


CASES = [
    # M, C (= K of the consumer), N, form, row offset (mean / spread of the rows)
    (640, 1280, 3840, "plain", 0.0),      # 8x8 level q|k|v
    (2560, 1280, 1280, "plain", 0.5),     # 16x16 level attn2.to_q
    (1000, 640, 1920, "plain", 4.0),      # ragged M, rows with mean 4x their spread
    (1280, 640, 5120, "geglu", 0.5),      # GEGLU projection
    (640, 320, 960, "rowvec", 0.3),       # motion module: per-frame row table (64 rows per frame)
    (300, 64, 192, "rowvec", 0.3),        # the block goldens' width, row table with fewer rows per sample than a tile
]


@pytest.mark.parametrize("cvariant", [-1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
@pytest.mark.parametrize("M,C,N,form,shift", CASES)
def test_gemm_lnx_vs_reference(hiplib, M, C, N, form, shift, cvariant):
    from rcdms_amd import hip
    g = torch.Generator().manual_seed(99 + M + C + N)
    Kp = 64
    Ap = h16(torch.randn(M, Kp, generator=g))
    Wp = h16(torch.randn(C, Kp, generator=g) * Kp ** -0.5)
    bp = torch.randn(C, generator=g) * 0.1
    res = h16(torch.randn(M, C, generator=g) * 2 + shift * 2 * torch.randn(M, 1, generator=g).abs())
    x, stat, parts = _producer(hip, Ap, Wp, bp, res, M, C, Kp, -1)
    xs = x.float().cpu()
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    W = torch.randn(N, C, generator=g) * C ** -0.5
    bias = 0.1 * torch.randn(N, generator=g)
    ln = torch.nn.functional.layer_norm(xs, (C,), gamma, beta, 1e-5)
    rps, rowvec_d, epi = 1, 0, 1
    if form == "rowvec":
        frames, rpf = 5, 64 if M == 640 else 20
        pe = 0.5 * torch.randn(frames, C, generator=g)
        fidx = (torch.arange(M) // rpf) % frames
        ref = (ln + pe[fidx]) @ h16(W).t() + bias
        Wg, S, bf = _fold(W, gamma, beta, bias)
        nsamp = (M + rpf - 1) // rpf
        tab = torch.stack([bf + (W * pe[s % frames][None, :]).sum(dim=1) for s in range(nsamp)]).contiguous()
        rowvec_t = tab.to(DEV)
        rowvec_d, rps, epi = rowvec_t.data_ptr(), rpf, 2
        bias_d = 0
    else:
        ref = ln @ h16(W).t() + bias
        Wg, S, bf = _fold(W, gamma, beta, bias)
        bias_t = bf.to(DEV)
        bias_d = bias_t.data_ptr()
    n_out = N
    if form == "geglu":
        # hidden | gate halves; the library wants rows interleaved 16 | 16 (rcdm_pack_geglu_rows) — pack the FOLDED matrix
        Wf32 = (W * gamma[None, :]).contiguous().to(DEV)
        bf32 = bf.contiguous().to(DEV)
        Wd = torch.empty(N, C, dtype=torch.float16, device=DEV)
        bias_t = torch.empty(N, dtype=torch.float32, device=DEV)
        hip.pack_geglu_rows(Wf32.data_ptr(), bf32.data_ptr(), N, C, Wd.data_ptr(), bias_t.data_ptr())
        torch.cuda.synchronize()
        Sd = Wd.float().sum(dim=1).contiguous()
        bias_d, epi, n_out = bias_t.data_ptr(), 1 | 8, N // 2
        # reference: ln W^T with the UNFOLDED matrix (what the LayerNorm + GEGLU of the reference computes)
        full = ln @ W.t() + bias
        hid, gate = full[:, :N // 2], full[:, N // 2:]
        ref = hid * torch.nn.functional.gelu(gate)
    else:
        Wd = Wg.to(DEV)
        Sd = S.to(DEV)
    ldc = n_out + 8
    out = torch.full((M, ldc), float("nan"), dtype=torch.float16, device=DEV)
    hip.set_igemm_variant(cvariant)
    d = hip.GemmDesc(M, N, C, C, ldc, 0, epi, rps, N if form == "rowvec" else 0, 1.0, 1, 0)
    lx = hip.Lnx(0, 0, 0, stat.data_ptr(), parts, M, Sd.data_ptr(), 1e-5, C)
    if cvariant == 8:   # the 256x256 ping-pong tile has no consumer epilogue: refused loudly, never computed wrongly
        with pytest.raises(hip.RcdmError):
            hip.gemm_lnx(d, lx, x.data_ptr(), Wd.data_ptr(), bias_d, rowvec_d, 0, out.data_ptr(), 0, 0)
        hip.set_igemm_variant(-1)
        return
    hip.gemm_lnx(d, lx, x.data_ptr(), Wd.data_ptr(), bias_d, rowvec_d, 0, out.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    hip.set_igemm_variant(-1)
    got = out[:, :n_out].float().cpu()
    assert torch.isnan(out[:, n_out:].float()).all(), "wrote outside the output columns"
    # against the LayerNorm -> Linear of the reference in fp32 (f16 storage of W and of the output are the only roundings
    # the separate-launch path has too; the deferred form adds the f16 staging of x W'^T before the mean term is removed)
    close(got, ref, rel=4e-3, abs_frac=6e-3)


@pytest.mark.parametrize("cvariant", [-1, 1, 5, 9, 10])
@pytest.mark.parametrize("form", ["plain", "geglu"])
@pytest.mark.parametrize("mean,sigma", [(800.0, 8.0), (20000.0, 200.0)])
def test_gemm_lnx_rows_far_from_zero(hiplib, mean, sigma, form, cvariant):
    """Rows whose mean is 100x their spread, and rows of large magnitude: the raw projection x W'^T is then
    ~ mean S[n] — thousands where the normalised result is O(1), and beyond the f16 range (65504) in the second case — so a
    consumer that rounded it to f16 before removing the mean term would be off by ~2^-11 |mean S| rstd (0.1 and more) or
    produce inf.  Every tile variant applies rstd acc - (mean rstd) S + b' to the fp32 accumulators; the tolerance below is
    the ordinary one plus the cancellation of var = E[x^2] - mean^2 in fp32 at mean / sigma = 100 (1e4 x 2^-24 x a few)."""
    from rcdms_amd import hip
    M, C, N = 960, 640, 1920 if form == "plain" else 2560
    g = torch.Generator().manual_seed(int(mean) + (form == "geglu"))
    Kp = 64
    Ap = h16(torch.randn(M, Kp, generator=g) * 0.01)
    Wp = h16(torch.randn(C, Kp, generator=g) * Kp ** -0.5)
    bp = torch.zeros(C)
    sign = torch.where(torch.rand(M, 1, generator=g) < 0.5, -1.0, 1.0)
    res = h16(sign * mean + sigma * torch.randn(M, C, generator=g))
    x, stat, parts = _producer(hip, Ap, Wp, bp, res, M, C, Kp, -1)
    xs = x.float().cpu()
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    W = torch.randn(N, C, generator=g) * C ** -0.5
    bias = 0.1 * torch.randn(N, generator=g)
    ln = torch.nn.functional.layer_norm(xs.double(), (C,), gamma.double(), beta.double(), 1e-5).float()
    Wg, S, bf = _fold(W, gamma, beta, bias)
    assert (xs.abs().max() * S.abs().max()).item() > (3e3 if mean < 1e4 else 65504), "the case must stress the raw projection"
    if form == "geglu":
        Wf32 = (W * gamma[None, :]).contiguous().to(DEV)
        bf32 = bf.contiguous().to(DEV)
        Wd = torch.empty(N, C, dtype=torch.float16, device=DEV)
        bias_t = torch.empty(N, dtype=torch.float32, device=DEV)
        hip.pack_geglu_rows(Wf32.data_ptr(), bf32.data_ptr(), N, C, Wd.data_ptr(), bias_t.data_ptr())
        torch.cuda.synchronize()
        Sd = Wd.float().sum(dim=1).contiguous()
        epi, n_out = 1 | 8, N // 2
        full = ln @ W.t() + bias
        ref = full[:, :N // 2] * torch.nn.functional.gelu(full[:, N // 2:])
    else:
        Wd, Sd, bias_t = Wg.to(DEV), S.to(DEV), bf.to(DEV)
        epi, n_out = 1, N
        ref = ln @ h16(W).t() + bias
    out = torch.full((M, n_out), float("nan"), dtype=torch.float16, device=DEV)
    hip.set_igemm_variant(cvariant)
    d = hip.GemmDesc(M, N, C, C, n_out, 0, epi, 1, 0, 1.0, 1, 0)
    lx = hip.Lnx(0, 0, 0, stat.data_ptr(), parts, M, Sd.data_ptr(), 1e-5, C)
    hip.gemm_lnx(d, lx, x.data_ptr(), Wd.data_ptr(), bias_t.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    hip.set_igemm_variant(-1)
    got = out.float().cpu()
    assert torch.isfinite(got).all(), "inf / NaN: a raw x W'^T value left the f16 range before the mean term was removed"
    close(got, ref, rel=8e-3, abs_frac=8e-3)


def test_gemm_lnx_dup_rows_and_both_sides(hiplib):
    """One call that is consumer AND producer, with dup_rows: out rows (and their statistics) are stored twice — the shared
    CFG prefix form (engine.emit_basic_block shared_half)."""
    from rcdms_amd import hip
    M, C, Kp = 512, 640, 64
    g = torch.Generator().manual_seed(3)
    Ap = h16(torch.randn(M, Kp, generator=g))
    Wp = h16(torch.randn(C, Kp, generator=g) * Kp ** -0.5)
    bp = torch.randn(C, generator=g) * 0.1
    res = h16(torch.randn(M, C, generator=g))
    x, stat, parts = _producer(hip, Ap, Wp, bp, res, M, C, Kp, -1, dup=M)
    xs = x.float().cpu()
    assert torch.equal(xs[:M], xs[M:]), "dup rows differ"
    st = stat.cpu().view(parts, 2 * M, 2)
    assert torch.equal(st[:, :M], st[:, M:]), "statistics of the dup rows differ"
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    W = torch.randn(C, C, generator=g) * C ** -0.5
    Wg, S, bf = _fold(W, gamma, beta, None)
    d = hip.GemmDesc(M, C, C, C, C, 0, 1, 1, 0, 1.0, 1, M)
    parts2 = hip.gemm_stat_parts(d)
    out = torch.full((2 * M, C), float("nan"), dtype=torch.float16, device=DEV)
    stat2 = torch.full((2 * M * parts2 * 2,), float("nan"), dtype=torch.float32, device=DEV)
    Wd, Sd, bd = Wg.to(DEV), S.to(DEV), bf.to(DEV)
    lx = hip.Lnx(stat2.data_ptr(), parts2, 2 * M, stat.data_ptr(), parts, 2 * M, Sd.data_ptr(), 1e-5, C)
    hip.gemm_lnx(d, lx, x.data_ptr(), Wd.data_ptr(), bd.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(xs[:M], (C,), gamma, beta, 1e-5) @ h16(W).t()
    o = out.float().cpu()
    close(o[:M], ref, rel=4e-3, abs_frac=6e-3)
    assert torch.equal(o[:M], o[M:])
    s2 = stat2.cpu().view(parts2, 2 * M, 2).sum(dim=0)
    assert torch.allclose(s2[:M, 0], o[:M].sum(dim=1), rtol=1e-4, atol=1e-2) and torch.equal(s2[:M], s2[M:])


def test_gemm_lnx_refusals(hiplib):
    """Loud failures: a statistics slot count that is not the launch's column-tile count, split-K on either side."""
    from rcdms_amd import hip
    M, C = 256, 640
    x = torch.zeros(M, C, dtype=torch.float16, device=DEV)
    W = torch.zeros(C, C, dtype=torch.float16, device=DEV)
    stat = torch.zeros(M * 64, dtype=torch.float32, device=DEV)
    S = torch.zeros(C, dtype=torch.float32, device=DEV)
    d = hip.GemmDesc(M, C, C, C, C, 0, 0, 1, 0, 1.0, 1, 0)
    parts = hip.gemm_stat_parts(d)
    with pytest.raises(hip.RcdmError):
        hip.gemm_lnx(d, hip.Lnx(stat.data_ptr(), parts + 1, M, 0, 0, 0, 0, 1e-5, 0), x.data_ptr(), W.data_ptr(), 0, 0, 0, x.data_ptr(), 0, 0)
    d2 = hip.GemmDesc(M, C, C, C, C, 0, 0, 1, 0, 1.0, 2, 0)
    w = ws(hip.gemm_workspace_bytes(d2))
    with pytest.raises(hip.RcdmError):
        hip.gemm_lnx(d2, hip.Lnx(0, 0, 0, stat.data_ptr(), parts, M, S.data_ptr(), 1e-5, C), x.data_ptr(), W.data_ptr(), 0, 0, 0,
                     x.data_ptr(), w.data_ptr(), w.numel())
    with pytest.raises(hip.RcdmError):
        hip.gemm_lnx(d, hip.Lnx(0, 0, 0, stat.data_ptr(), 21, M, S.data_ptr(), 1e-5, C), x.data_ptr(), W.data_ptr(), 0, 0, 0,
                     x.data_ptr(), 0, 0)
    torch.cuda.synchronize()
