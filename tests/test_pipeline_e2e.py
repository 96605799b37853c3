"""GPU: `RCDMsPipeline.__call__` end to end on the HIP path — prompt encoding and VAE are tiny stand-in torch modules
(the reference's are CLIP / AutoencoderKL checkpoints that do not exist here), everything between them is the product:
HIP context builders (rcdms_amd.context), the captured denoising loop on the tiny UNet, the reference's mask / context
bookkeeping.  Checked against the same flow computed with the oracles on CPU."""
import types

import pytest
import torch
from torch import nn

from oracle import context_oracle as CO
from oracle import unet_oracle as O
from rcdms_amd import context, synth
from rcdms_amd.scheduler import DDIMScheduler
from tests.test_hip_unet import build, rel_rms
from tests.test_oracle_golden import SEEDS, mirrored, shapes_of

pytestmark = pytest.mark.gpu
D = 64          # cross-attention dim of the tiny UNet
T = 13          # text tokens


class _Tok:
    model_max_length = T

    def __call__(self, texts, padding=None, max_length=T, truncation=False, return_tensors="pt"):
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, s in enumerate(texts):
            for j, ch in enumerate(s[:max_length]):
                ids[i, j] = 1 + (ord(ch) % 97)
        return types.SimpleNamespace(input_ids=ids)


class _Text(nn.Module):
    max_position_embeddings = T

    def __init__(self):
        super().__init__()
        self.emb = nn.Embedding(100, D)
        with torch.no_grad():
            self.emb.weight.copy_(synth.normal_tensor("e2e.text_emb", (100, D), 1))

    def forward(self, ids):
        return types.SimpleNamespace(last_hidden_state=self.emb(ids))


class _Vae(nn.Module):
    """encode: 8x average pool, 3 -> 4 channels; decode: nearest 8x upsample of 3 channels."""

    def encode(self, x):
        z = nn.functional.avg_pool2d(x, 8)
        z = torch.cat([z, z.mean(1, keepdim=True)], dim=1)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda generator=None: z))

    def decode(self, z):
        return types.SimpleNamespace(sample=nn.functional.interpolate(z[:, :3], scale_factor=8, mode="nearest"))


@pytest.mark.parametrize("hip_vae", [False, True], ids=["stub_vae", "hip_vae"])
def test_pipeline_five_captions_matches_oracle_flow(hiplib, hip_vae):
    """The driver's call pattern (stage2_batchtest_rcdms_model.py:364-376): one caption per frame.  With hip_vae the
    `vae` argument is rcdms_amd.vae.AutoencoderKL (tiny shape), so encode, sampling loop and decode all run on HIP."""
    from oracle import vae_oracle as VO
    from rcdms_amd.vae import AutoencoderKL
    from src.pipelines.RCDMs_pipeline import RCDMsPipeline
    dev = "cuda"
    unet = build("unet_tiny")
    sd_unet = synth.procedural_state_dict(shapes_of(mirrored("unet_tiny")), SEEDS["unet_tiny"])
    cfg = O.tiny_config(width=64, cross_dim=D, layers_per_block=2)
    local = context.fine_stack(text_dim=D, vis_dim=32, hidden_dim=D, num_heads=8)
    glob = context.semantic_stack(text_dim=D, vis_dim=24, hidden_dim=D, num_heads=8)
    sd_l = synth.procedural_state_dict({k: v.shape for k, v in local.state_dict().items()}, 11)
    sd_g = synth.procedural_state_dict({k: v.shape for k, v in glob.state_dict().items()}, 12)
    local.load_state_dict(sd_l)
    glob.load_state_dict(sd_g)
    text, vae, tok = _Text(), _Vae(), _Tok()
    if hip_vae:
        vcfg = VO.tiny_vae_config()
        shapes = dict(VO.decoder_shapes(vcfg))
        shapes.update(VO.encoder_shapes(vcfg))
        sd_vae = synth.procedural_state_dict(shapes, 13)
        kw = dict(vcfg)
        kw["norm_num_groups"] = kw.pop("groups")
        vae = AutoencoderKL(**kw).eval()
        vae.load_state_dict(sd_vae)
    pipe = RCDMsPipeline(vae=vae, text_encoder=text, tokenizer=tok, unet=unet, local_module=local, global_module=glob,
                         scheduler=DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")).to(dev)
    H = W = 128
    caps = ["pororo waves", "loopy sings", "eddy builds", "crong jumps", "poby fishes"]
    src = synth.normal_tensor("e2e.src", (5, 3, H, W), 2) * 0.5
    mask_label = torch.zeros(1, 5, H // 8, W // 8)
    mask_label[:, 0] = 1.0
    img1 = synth.normal_tensor("e2e.img1", (1, 9, 32), 3)
    proj0 = synth.normal_tensor("e2e.proj0", (4, 1, 24), 4)
    lat0 = synth.normal_tensor("e2e.lat", (1, 4, 5, H // 8, W // 8), 5)
    steps, gs = 3, 2.0
    gen = torch.Generator(device=dev).manual_seed(9)           # consumed by latent_dist.sample only (latents are given)
    out = pipe(caps, src.to(dev), image_embeds_1=img1.to(dev), proj_embeds_0=proj0.to(dev), mask_label=mask_label.to(dev),
               video_length=5, height=H, width=W, num_inference_steps=steps, guidance_scale=gs, latents=lat0.to(dev),
               generator=gen).videos
    assert tuple(out.shape) == (1, 3, 5, H, W) and torch.isfinite(out).all()

    emb = text.emb.weight.detach().cpu()
    te = torch.cat([emb[tok([""] * 5).input_ids], emb[tok(caps).input_ids]])          # (10, T, D), unconditional half first
    ml = torch.cat([mask_label[0], mask_label[0]])                                     # encode_mask
    seen = (ml.reshape(10, -1) == 1).all(1)
    f1 = CO.context_stack_forward(sd_l, torch.cat([img1] * 2), te[seen])              # local module on the seen rows
    f0 = CO.context_stack_forward(sd_g, torch.cat([proj0] * 2), te[~seen])            # global module on the rest
    ctx = torch.cat([f1, f0])                                                          # reference order: seen rows first (F5)
    if hip_vae:
        noise = torch.randn(5, 4, H // 8, W // 8, generator=torch.Generator(device=dev).manual_seed(9), device=dev).cpu()
        with torch.no_grad():
            z = VO.vae_encode_sample(sd_vae, vcfg, src, noise)
    else:
        z = nn.functional.avg_pool2d(src, 8)
        z = torch.cat([z, z.mean(1, keepdim=True)], dim=1)                             # (5,4,h,w)
    masked = (z.reshape(1, 5, 4, H // 8, W // 8).permute(0, 2, 1, 3, 4) * 0.18215)
    masked = torch.cat([masked] * 2)
    mask5 = ml.view(2, 1, 5, H // 8, W // 8)
    with torch.no_grad():
        lat = O.denoise_loop(sd_unet, cfg, lat0, mask5, masked, ctx, steps, gs)
    zf = (lat / 0.18215).permute(0, 2, 1, 3, 4).reshape(5, 4, H // 8, W // 8)
    with torch.no_grad():
        want = VO.vae_decode(sd_vae, vcfg, zf) if hip_vae else vae.decode(zf).sample
    want = (want.reshape(1, 5, 3, H, W).permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
    r = rel_rms(out.float(), want)
    print(f"pipeline e2e ({'hip' if hip_vae else 'stub'} vae): rel-RMS {r:.3e}")
    assert r <= (2.4e-3 if hip_vae else 9.4e-3), r            # measured 1.2e-3 (HIP VAE) / 4.7e-3 (fp32 torch stub VAE)
