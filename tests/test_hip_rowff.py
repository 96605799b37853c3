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


@pytest.mark.parametrize("variant", [0, 1])           # ten waves x 16 rows | four waves x 48 rows per block
@pytest.mark.parametrize("M", [160, 16, 333, 2560])   # one block; one wave; ragged tail; many blocks
def test_ff_fused_vs_reference(hiplib, M, variant):
    from rcdms_amd import hip
    hip.set_ff_variant(variant)
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
    hip.set_ff_variant(-1)
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
    wsb = torch.empty(1 << 24, dtype=torch.uint8, device=DEV)
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
