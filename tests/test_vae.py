"""VAE: rcdms_amd.vae.AutoencoderKLDecoder / AutoencoderKL vs the CPU restatement oracle/vae_oracle.py.
PARITY UNPINNED — diffusers' AutoencoderKL is third-party code absent from /root/reference; the oracle restates the
published architecture and is the only check available (stated in both headers)."""
import pytest
import torch

from oracle import vae_oracle as V
from rcdms_amd import hip, synth, vae


def _cfg(c):
    d = dict(c)
    d["norm_num_groups"] = d.pop("groups")
    return d


def test_state_dict_keys_are_the_diffusers_decoder_keys():
    m = vae.AutoencoderKLDecoder()
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == V.decoder_shapes(V.SD15_VAE)
    assert len(m.state_dict()) == 140


def test_cpu_module_fails_loudly():
    m = vae.AutoencoderKLDecoder(**_cfg(V.tiny_vae_config()))
    with pytest.raises(hip.RcdmError):
        m.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(NotImplementedError):
        m.encode(torch.zeros(1, 3, 64, 64))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,n,hw", [("tiny", V.tiny_vae_config(), 2, 8), ("sd15", V.SD15_VAE, 1, 32)])
def test_hip_vae_decode_vs_oracle(name, cfg, n, hw):
    m = vae.AutoencoderKLDecoder(**_cfg(cfg)).eval()
    sd = synth.procedural_state_dict(V.decoder_shapes(cfg), 31)
    m.load_state_dict(sd)
    m = m.to("cuda")
    z = synth.normal_tensor(f"vae.{name}.z", (n, 4, hw, hw), 32)
    got = m.decode(z.cuda()).sample.float().cpu()
    with torch.no_grad():
        want = V.vae_decode(sd, cfg, z)
    assert got.shape == want.shape == (n, 3, 8 * hw, 8 * hw) and torch.isfinite(got).all()
    rel = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    mx = float((got - want).abs().max() / want.abs().max())
    print(f"vae {name}: rel-RMS {rel:.3e} max {mx:.3e}")
    assert rel <= 4e-3 and mx <= 4.2e-3, (rel, mx)           # measured 2.0e-3 / 2.1e-3 (tiny), 1.8e-3 / 2.0e-3 (SD-1.5 shape)
    got2 = m.decode(z.cuda()).sample.float().cpu()          # cached program, deterministic
    assert torch.equal(got, got2)


def test_full_vae_state_dict_keys():
    m = vae.AutoencoderKL()
    want = dict(V.decoder_shapes(V.SD15_VAE))
    want.update(V.encoder_shapes(V.SD15_VAE))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want and len(want) == 248
    assert m.config.scaling_factor == 0.18215 and tuple(m.config.block_out_channels) == (128, 256, 512, 512)
    with pytest.raises(hip.RcdmError):
        m.encode(torch.zeros(1, 3, 64, 64))


def test_oracle_downsample_is_pad_after_only():
    """The asymmetric padding is what distinguishes diffusers' VAE downsampler from the UNet's; pin the restatement's
    arithmetic on a hand-checkable case: an all-ones 3x3 kernel over a 4x4 ramp."""
    import torch.nn.functional as F
    x = torch.arange(16.0).reshape(1, 1, 4, 4)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), torch.ones(1, 1, 3, 3), stride=2)
    assert y.flatten().tolist() == [45.0, 39.0, 66.0, 50.0]   # windows at (0,0),(0,2),(2,0),(2,2); last col/row zero


@pytest.mark.gpu
@pytest.mark.parametrize("n,H,W,cin,cout", [(2, 16, 16, 64, 64), (1, 32, 24, 128, 128), (3, 8, 8, 320, 64)])
def test_conv3x3_pad_after_only(n, H, W, cin, cout):
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(H * W + cin)
    x = (torch.randn(n, cin, H, W, generator=g)).half().float()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half().float()
    bias = torch.randn(cout, generator=g)
    want = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, bias, stride=2)
    xd = x.permute(0, 2, 3, 1).reshape(-1, cin).half().cuda().contiguous()
    wp = torch.empty(cout, 9 * cin, dtype=torch.float16, device="cuda")
    wd, bd = w.cuda(), bias.cuda()
    hip.pack_conv3x3(wd.data_ptr(), cout, cin, cin, wp.data_ptr())
    out = torch.empty(n * (H // 2) * (W // 2), cout, dtype=torch.float16, device="cuda")
    d = hip.ConvDesc(n, H, W, cin, cout, 2, 0, cin, cout, 0, hip.EPI_BIAS, 1, 0, 1.0, 0, 1)
    wsb = torch.full((max(hip.conv3x3_workspace_bytes(d), 256),), 0xFF, dtype=torch.uint8, device="cuda")   # NaN-filled
    hip.conv3x3(d, xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), 0, 0, out.data_ptr(), wsb.data_ptr(), wsb.numel())
    torch.cuda.synchronize()
    got = out.float().cpu().reshape(n, H // 2, W // 2, cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, want, rtol=5e-3, atol=5e-3), float((got - want).abs().max())
    # the symmetric form on the same input differs (so the flag is really honoured) ...
    d0 = hip.ConvDesc(n, H, W, cin, cout, 2, 0, cin, cout, 0, hip.EPI_BIAS, 1, 0, 1.0, 0, 0)
    out0 = torch.empty_like(out)
    hip.conv3x3(d0, xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), 0, 0, out0.data_ptr(), wsb.data_ptr(), wsb.numel())
    torch.cuda.synchronize()
    assert not torch.allclose(out0.float(), out.float(), atol=5e-2)
    # ... and the form is refused where diffusers does not define it: stride 1, odd sizes
    for bad in (hip.ConvDesc(n, H, W, cin, cout, 1, 0, cin, cout, 0, hip.EPI_BIAS, 1, 0, 1.0, 0, 1),
                hip.ConvDesc(n, H + 1, W, cin, cout, 2, 0, cin, cout, 0, hip.EPI_BIAS, 1, 0, 1.0, 0, 1)):
        with pytest.raises(hip.RcdmError):
            hip.conv3x3(bad, xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), 0, 0, out.data_ptr(), wsb.data_ptr(), wsb.numel())


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,n,px", [("tiny", V.tiny_vae_config(), 2, 64), ("sd15", V.SD15_VAE, 1, 256)])
def test_hip_vae_encode_vs_oracle(name, cfg, n, px):
    m = vae.AutoencoderKL(**_cfg(cfg)).eval()
    shapes = dict(V.decoder_shapes(cfg))
    shapes.update(V.encoder_shapes(cfg))
    sd = synth.procedural_state_dict(shapes, 37)
    m.load_state_dict(sd)
    m = m.to("cuda")
    x = synth.normal_tensor(f"vae.{name}.px", (n, 3, px, px), 38).clamp(-1, 1)
    dist = m.encode(x.cuda()).latent_dist
    with torch.no_grad():
        mean, logvar = V.vae_encode_moments(sd, cfg, x)
    for nm, got, want in (("mean", dist.mean, mean), ("logvar", dist.logvar, logvar)):
        got = got.float().cpu()
        assert got.shape == want.shape == (n, 4, px // 8, px // 8) and torch.isfinite(got).all()
        rel = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
        print(f"vae encode {name} {nm}: rel-RMS {rel:.3e}")
        assert rel <= 3.5e-3, (nm, rel)                      # measured 1.5e-3 .. 1.7e-3
    # posterior sampling: same generator -> same draw; equals mean + std * noise of the oracle's formula
    gen = torch.Generator(device="cuda").manual_seed(5)
    s1 = dist.sample(generator=gen)
    noise = torch.randn(dist.mean.shape, generator=torch.Generator(device="cuda").manual_seed(5), device="cuda")
    assert torch.allclose(s1, dist.mean + torch.exp(0.5 * dist.logvar) * noise)
    assert torch.equal(dist.mode(), dist.mean)
    # round trip through the decoder half of the same module (cached programs of both halves)
    rec = m.decode(dist.mode()).sample
    assert rec.shape == (n, 3, px, px) and torch.isfinite(rec).all()


@pytest.mark.gpu
def test_softmax_rows():
    M, N = 130, 4096
    x = (torch.randn(M, N, generator=torch.Generator().manual_seed(3)) * 3).half()
    xd = x.cuda()
    y = torch.empty_like(xd)
    hip.softmax_rows(M, N, N, N, 0.25, xd.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    want = torch.softmax(x.float() * 0.25, dim=-1)
    assert torch.allclose(y.float().cpu(), want, rtol=2e-3, atol=1e-6)
