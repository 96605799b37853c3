"""VAE decoder: rcdms_amd.vae.AutoencoderKLDecoder vs the CPU restatement oracle/vae_oracle.py.
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
    assert rel <= 1e-2 and mx <= 5e-2, (rel, mx)
    got2 = m.decode(z.cuda()).sample.float().cpu()          # cached program, deterministic
    assert torch.equal(got, got2)


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
