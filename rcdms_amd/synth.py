"""Procedural (name-seeded) weights and the synthetic stage-2 story of SURVEY.md §8(d).

There are no checkpoints or datasets in this environment, so every parity fixture and the bench use
weights generated from the parameter NAME: the same tensor is regenerated bit-identically on the build
container (where the reference model is imported to mint golden outputs) and on the GPU box, without
shipping 5 GB of weights.  numpy's Philox bit generator is platform-stable.
"""
import math
import zlib

import numpy as np
import torch


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=[zlib.crc32(name.encode()) & 0xFFFFFFFF, seed & 0xFFFFFFFF]))


def sinusoid_table(d_model, max_len):
    """The fixed (non-learned) `pos_encoder.pe` buffer of the reference's PositionalEncoding
    (src/models/motion_module.py:258-263) — part of the 1286-key state dict."""
    pos = torch.arange(max_len, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(pos * div)
    pe[0, :, 1::2] = torch.cos(pos * div)
    return pe


def procedural_tensor(name, shape, seed=0, style="unit"):
    """Value of parameter `name`:
      *.pe                         -> the sinusoid table (a buffer, not random)
      norm weights (1-D "*.weight" of a norm) -> 1 + 0.1 N(0,1); 1-D biases -> 0.05 N(0,1)
      matrices / conv kernels      -> N(0, 1/fan_in)  (unit gain, keeps activations O(1)).
    style "skewed" — a second weight family for the parity tests, with the features of trained weights the unit-gain
    family lacks: every matrix row (output channel) and column (input channel) carries its own log-normal gain
    (sigma 0.5 / 0.35: outlier channels, sharper attention logits), one entry in 512 is 6x larger (heavy tail), norm
    weights are 1 + 0.3 N(0,1), norm biases 0.2 N(0,1), other biases 0.1 N(0,1)."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("pos_encoder.pe"):
        return sinusoid_table(shape[2], shape[1])
    if style == "sdlike":
        return _sdlike_tensor(name, shape, seed)
    if style not in ("unit", "skewed"):
        raise ValueError(f"unknown weight style {style!r}")
    g = _rng(name, seed)
    x = g.standard_normal(size=shape, dtype=np.float32)
    skew = style == "skewed"
    if len(shape) == 1:
        if name.endswith(".weight") and _is_norm_param(name):
            x = 1.0 + (0.3 if skew else 0.1) * x
        elif skew:
            x = (0.2 if _is_norm_param(name) else 0.1) * x
        else:
            x = 0.05 * x
    else:
        fan_in = int(np.prod(shape[1:]))
        x = x * (1.0 / math.sqrt(fan_in))
        if skew:
            bshape_o = (shape[0],) + (1,) * (len(shape) - 1)
            bshape_i = (1, shape[1]) + (1,) * (len(shape) - 2)
            x = x * np.exp(0.5 * g.standard_normal(size=bshape_o, dtype=np.float32))
            x = x * np.exp(0.35 * g.standard_normal(size=bshape_i, dtype=np.float32))
            x = np.where(g.random(size=shape, dtype=np.float32) < 1.0 / 512, 6.0 * x, x).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


SDLIKE_OUTLIER_CHANNELS, SDLIKE_OUTLIER_GAIN, SDLIKE_QK_GAIN = 2, (50.0, 100.0), 3.0


def _sdlike_tensor(name, shape, seed):
    """style "sdlike" — the third weight family: the unit-gain family plus the two features of trained
    Stable-Diffusion weights that decide whether an f16 path survives them:
      * "massive activations": conv1 and conv2 of every ResnetBlock3D have SDLIKE_OUTLIER_CHANNELS output channels whose
        kernel (and bias) carry a gain drawn from U[50, 100) — channels 50-100x above their neighbours feeding the
        GroupNorm behind them (the group that holds one is dominated by it) and accumulating on the residual stream;
      * sharp attention: to_q and to_k of every spatial self-attention (attn1) carry a gain of SDLIKE_QK_GAIN each, so
        the logits are 25x the unit family's — in the hundreds at the tails, a near-one-hot softmax.
    Everything else as "unit".  (Trained checkpoints are not available here; this family is a stress shape, not a model.)"""
    x = procedural_tensor(name, shape, seed, "unit").numpy().copy()
    owner = name.rsplit(".", 1)[0]
    # ResnetBlock3D.conv1 / conv2 (resnet.py:147,167) only: conv1 feeds norm2, conv2 ADDS to the residual stream.  The layers
    # ON the residual path (conv_shortcut, the samplers' conv, conv_in) keep unit gain — outlier gains there multiply from
    # level to level (75^6 over the six samplers: the reference itself reaches 1e12) instead of modelling outlier channels
    if owner.rsplit(".", 1)[-1] in ("conv1", "conv2") and name.endswith((".weight", ".bias")):
        g = _rng(owner + "#sdlike", seed)          # weight and bias of one layer share their outlier channels
        ch = g.choice(shape[0], size=min(SDLIKE_OUTLIER_CHANNELS, shape[0]), replace=False)
        gain = g.uniform(*SDLIKE_OUTLIER_GAIN, size=len(ch)).astype(np.float32)
        for c, a in zip(ch, gain):
            x[c] *= a
    elif ".attn1.to_q.weight" in name or ".attn1.to_k.weight" in name:
        x *= SDLIKE_QK_GAIN
    return torch.from_numpy(np.ascontiguousarray(x))


def normal_tensor(name, shape, seed=0):
    """Standard-normal test INPUT named `name` (platform-stable Philox stream, like the weights)."""
    g = _rng(name, seed)
    return torch.from_numpy(g.standard_normal(size=tuple(int(s) for s in shape), dtype=np.float32))


def _is_norm_param(name):
    parts = name.split(".")
    owner = parts[-2]
    if owner.isdigit():  # "...norms.0.weight"
        owner = parts[-3]
    return "norm" in owner


def procedural_state_dict(shapes, seed=0, style="unit"):
    """shapes: mapping parameter name -> shape (e.g. from `state_shapes(cfg)` or a module's state_dict)."""
    return {k: procedural_tensor(k, tuple(v), seed, style) for k, v in shapes.items()}


def synthetic_story(stories=1, frames=5, latent_hw=(64, 64), ctx_len=85, ctx_dim=768, cfg=True, seed=42, structure="dense"):
    """CPU-seeded inputs of the denoising loop for `stories` stories.
    Returns dict(latents (S,4,f,h,w), mask (R*S,1,f,h,w), masked_latents (R*S,4,f,h,w), ctx (R*S*f, L, D)),
    R = 2 with CFG.  mask = [1,0,0,0,0] per story (first frame seen), masked latents of unseen frames are a
    constant (the stand-in for the VAE latent of a black frame, RCDMs_pipeline.py:427-432).
    structure: "dense" (the headline workload, SURVEY §8d: every context row N(0,1)) or "reference" — the row structure the
    reference's context builders really produce for this mask (RCDMs_pipeline.py:444-450): the seen frames'
    rows first ([u0, c0] per story: fine_stack output, dense), then the unseen frames' ([u1..u4, c1..c4]: semantic_stack
    has ONE key / value token, stage2_batchtest_rcdms_model.py:117-132, so all L rows of such an image are one vector)."""
    g = torch.Generator().manual_seed(seed)
    h, w = latent_hw
    reps = 2 if cfg else 1
    lat = torch.randn(stories, 4, frames, h, w, generator=g)
    ml = 0.18215 * torch.randn(stories, 4, frames, h, w, generator=g)
    ml[:, :, 1:] = 0.18215 * torch.tensor([0.9, -0.6, 0.3, -1.2]).view(1, 4, 1, 1, 1)
    mask = torch.zeros(stories, 1, frames, h, w)
    mask[:, :, 0] = 1.0
    ctx = torch.randn(reps * stories * frames, ctx_len, ctx_dim, generator=g)
    if structure == "reference":
        n_seen = reps * stories * 1                      # mask = [1, 0, 0, 0, 0]: one seen frame per (CFG half, story)
        ctx[n_seen:] = ctx[n_seen:, :1, :].expand(-1, ctx_len, -1).clone()
    elif structure != "dense":
        raise ValueError(f"structure {structure!r}")
    return dict(latents=lat, mask=torch.cat([mask] * reps), masked_latents=torch.cat([ml] * reps), ctx=ctx)


# ---- a tiny SD-1.5-style 2-D UNet checkpoint directory (config.json + diffusion_pytorch_model.bin) -----------------
TINY_2D_CONFIG = {   # the fields of runwayml/stable-diffusion-v1-5 unet/config.json, narrowed to width 32
    "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.6.0", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": [32, 64, 64, 64], "center_input_sample": False, "cross_attention_dim": 64,
    "down_block_types": ["CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"],
    "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "layers_per_block": 2,
    "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32, "out_channels": 4, "sample_size": 8,
    "up_block_types": ["UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"]}


def write_2d_checkpoint(dirpath, shapes_3d, seed=11, extra_key="unexpected_2d_only.weight"):
    """Write <dirpath>/config.json and <dirpath>/diffusion_pytorch_model.bin the way an SD-1.5 `unet/` folder looks to
    UNet3DConditionModel.from_pretrained_2d (reference unet.py:465-509): the 2-D model has no motion-module tensors, a
    4-channel conv_in, and here one extra tensor the 3-D model does not know.  shapes_3d: name -> shape of the inflated
    3-D model's state dict (either side's: the layouts are digest-checked identical).  Values are name-seeded."""
    import json
    import os
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "config.json"), "w") as f:
        json.dump(TINY_2D_CONFIG, f)
    sd = {}
    for k, shp in shapes_3d.items():
        if "motion_modules" in k:
            continue
        shp = tuple(shp)
        if k == "conv_in.weight":
            shp = (shp[0], 4) + shp[2:]
        sd[k] = procedural_tensor(k, shp, seed)
    sd[extra_key] = procedural_tensor(extra_key, (3, 5), seed)
    torch.save(sd, os.path.join(dirpath, "diffusion_pytorch_model.bin"))
    return sd
