"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU fp32 restatement of the VAE the stage-2 pipeline calls: the *decoder* once per
frame (`self.vae.decode(latents[frame_idx:frame_idx+1]).sample`, src/pipelines/RCDMs_pipeline.py:274-287) and the
*encoder* once per story (`self.vae.encode(source_img).latent_dist.sample(generator)`, RCDMs_pipeline.py:429); the
module is the SD-1.5 `AutoencoderKL` loaded at stage2_batchtest_rcdms_model.py:205.  SURVEY §8f N3.

PARITY UNPINNED: AutoencoderKL is diffusers==0.24.0 code (requirements.txt:12), absent from /root/reference and not
installed; there is no reference output to pin against.  This file restates the published architecture — Decoder(conv_in,
UNetMidBlock2D[ResnetBlock2D, Attention(1 head), ResnetBlock2D], 4 x UpDecoderBlock2D[3 ResnetBlock2D (+ nearest-2x
Upsample2D conv)], GroupNorm(32, eps 1e-6) + SiLU + conv_out) behind post_quant_conv, and Encoder(conv_in, 4 x
DownEncoderBlock2D[2 ResnetBlock2D (+ Downsample2D: F.pad (0,1,0,1) then a stride-2 conv)], the same mid block,
GroupNorm + SiLU + conv_out) in front of quant_conv and the diagonal-Gaussian posterior — with the diffusers 0.24
state-dict key names, so a real `vae/diffusion_pytorch_model.bin` loads.  Only tests/ may import it."""
import torch
import torch.nn.functional as F

SD15_VAE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3, groups=32)


def tiny_vae_config():
    return dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1, latent_channels=4, out_channels=3, groups=32)


def decoder_shapes(cfg):
    """name -> shape of every decoder-side parameter (post_quant_conv + decoder.*), diffusers 0.24 naming."""
    boc, lc = list(cfg["block_out_channels"]), cfg["latent_channels"]
    top = boc[-1]
    sh = {"post_quant_conv.weight": (lc, lc, 1, 1), "post_quant_conv.bias": (lc,),
          "decoder.conv_in.weight": (top, lc, 3, 3), "decoder.conv_in.bias": (top,)}

    def resnet(p, cin, cout):
        sh.update({p + "norm1.weight": (cin,), p + "norm1.bias": (cin,), p + "conv1.weight": (cout, cin, 3, 3),
                   p + "conv1.bias": (cout,), p + "norm2.weight": (cout,), p + "norm2.bias": (cout,),
                   p + "conv2.weight": (cout, cout, 3, 3), p + "conv2.bias": (cout,)})
        if cin != cout:
            sh.update({p + "conv_shortcut.weight": (cout, cin, 1, 1), p + "conv_shortcut.bias": (cout,)})

    m = "decoder.mid_block."
    resnet(m + "resnets.0.", top, top)
    resnet(m + "resnets.1.", top, top)
    a = m + "attentions.0."
    sh.update({a + "group_norm.weight": (top,), a + "group_norm.bias": (top,)})
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh.update({a + n + ".weight": (top, top), a + n + ".bias": (top,)})
    rev = boc[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c)
        if i < len(rev) - 1:
            sh.update({f"decoder.up_blocks.{i}.upsamplers.0.conv.weight": (c, c, 3, 3),
                       f"decoder.up_blocks.{i}.upsamplers.0.conv.bias": (c,)})
        prev = c
    sh.update({"decoder.conv_norm_out.weight": (rev[-1],), "decoder.conv_norm_out.bias": (rev[-1],),
               "decoder.conv_out.weight": (cfg["out_channels"], rev[-1], 3, 3), "decoder.conv_out.bias": (cfg["out_channels"],)})
    return sh


def encoder_shapes(cfg):
    """name -> shape of every encoder-side parameter (encoder.* + quant_conv), diffusers 0.24 naming."""
    boc, lc = list(cfg["block_out_channels"]), cfg["latent_channels"]
    cin = cfg.get("in_channels", 3)
    sh = {"encoder.conv_in.weight": (boc[0], cin, 3, 3), "encoder.conv_in.bias": (boc[0],)}

    def resnet(p, ci, co):
        sh.update({p + "norm1.weight": (ci,), p + "norm1.bias": (ci,), p + "conv1.weight": (co, ci, 3, 3),
                   p + "conv1.bias": (co,), p + "norm2.weight": (co,), p + "norm2.bias": (co,),
                   p + "conv2.weight": (co, co, 3, 3), p + "conv2.bias": (co,)})
        if ci != co:
            sh.update({p + "conv_shortcut.weight": (co, ci, 1, 1), p + "conv_shortcut.bias": (co,)})

    prev = boc[0]
    for i, c in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}.", prev if j == 0 else c, c)
        if i < len(boc) - 1:
            sh.update({f"encoder.down_blocks.{i}.downsamplers.0.conv.weight": (c, c, 3, 3),
                       f"encoder.down_blocks.{i}.downsamplers.0.conv.bias": (c,)})
        prev = c
    top = boc[-1]
    m = "encoder.mid_block."
    resnet(m + "resnets.0.", top, top)
    resnet(m + "resnets.1.", top, top)
    a = m + "attentions.0."
    sh.update({a + "group_norm.weight": (top,), a + "group_norm.bias": (top,)})
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh.update({a + n + ".weight": (top, top), a + n + ".bias": (top,)})
    sh.update({"encoder.conv_norm_out.weight": (top,), "encoder.conv_norm_out.bias": (top,),
               "encoder.conv_out.weight": (2 * lc, top, 3, 3), "encoder.conv_out.bias": (2 * lc,),
               "quant_conv.weight": (2 * lc, 2 * lc, 1, 1), "quant_conv.bias": (2 * lc,)})
    return sh


def resnet2d(sd, p, x, groups):
    """diffusers ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1)."""
    h = F.conv2d(F.silu(F.group_norm(x, groups, sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)),
                 sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(F.group_norm(h, groups, sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)),
                 sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h


def mid_attention(sd, p, x, groups):
    """diffusers Attention(heads=1, dim_head=C, norm_num_groups, residual_connection=True, bias=True)."""
    n, c, hh, ww = x.shape
    h = F.group_norm(x, groups, sd[p + "group_norm.weight"], sd[p + "group_norm.bias"], 1e-6)
    t = h.reshape(n, c, hh * ww).transpose(1, 2)
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    pr = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
    o = F.linear(pr @ v, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, hh, ww) + x


def vae_decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample: z (n, 4, h, w) already divided by the scaling factor -> (n, 3, 8h, 8w)."""
    sd = {k: v.float() for k, v in sd.items()}
    g = cfg["groups"]
    x = F.conv2d(z.float(), sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    m = "decoder.mid_block."
    x = resnet2d(sd, m + "resnets.0.", x, g)
    x = mid_attention(sd, m + "attentions.0.", x, g)
    x = resnet2d(sd, m + "resnets.1.", x, g)
    nb = len(cfg["block_out_channels"])
    for i in range(nb):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet2d(sd, f"decoder.up_blocks.{i}.resnets.{j}.", x, g)
        if i < nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def vae_encode_moments(sd, cfg, x):
    """AutoencoderKL.encode(x).latent_dist as (mean, logvar): x (n, 3, H, W) in [-1, 1] -> two (n, 4, H/8, W/8) tensors;
    logvar already clamped to [-30, 20] as DiagonalGaussianDistribution does."""
    sd = {k: v.float() for k, v in sd.items()}
    g = cfg["groups"]
    x = F.conv2d(x.float(), sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    nb = len(cfg["block_out_channels"])
    for i in range(nb):
        for j in range(cfg["layers_per_block"]):
            x = resnet2d(sd, f"encoder.down_blocks.{i}.resnets.{j}.", x, g)
        if i < nb - 1:
            # Downsample2D(padding=0): zero row / column appended at the bottom / right only, then stride 2
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    m = "encoder.mid_block."
    x = resnet2d(sd, m + "resnets.0.", x, g)
    x = mid_attention(sd, m + "attentions.0.", x, g)
    x = resnet2d(sd, m + "resnets.1.", x, g)
    x = F.silu(F.group_norm(x, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], 1e-6))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    x = F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_encode_sample(sd, cfg, x, noise):
    """latent_dist.sample(): mean + exp(0.5 logvar) * noise."""
    mean, logvar = vae_encode_moments(sd, cfg, x)
    return mean + torch.exp(0.5 * logvar) * noise
