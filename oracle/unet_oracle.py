"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A CPU fp32 restatement, in plain functional torch over a flat state dict, of the reference's stage-2
denoising path (muzishen/RCDMs): `UNet3DConditionModel.forward` and the CFG + DDIM loop that calls it.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the product (`rcdms_amd/`, `src/`) never does and has no CPU fallback.

Pinning status:
  * UNet forward and every block below it: PINNED — checked tensor-for-tensor against the reference's
    own classes (imported in the build container by oracle/make_golden.py) through the golden vectors
    in tests/golden/ (tests/test_oracle_golden.py).
  * Timesteps / TimestepEmbedding / FeedForward(GEGLU) / DDIMScheduler: third-party arithmetic from
    diffusers==0.24.0 (reference requirements.txt:12), absent from /root/reference and not installed:
    restated from the published definitions, **parity unpinned** by any reference test; pinned only by
    closed-form known-answer tests (tests/test_oracle_known_answers.py).

Every function cites the reference file:line (relative to the reference repo root) it follows.
"""
import math

import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# configuration (what UNet3DConditionModel.__init__ registers, src/models/unet.py:41-90, merged with
# configs/testing.yaml:1-15 at stage2_batchtest_rcdms_model.py:220-223)

SD15_STAGE2_CONFIG = dict(
    in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5,
    down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
    up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
    use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=False,
    motion_num_attention_heads=8, motion_num_transformer_block=1, motion_attention_blocks=2,
    motion_pe_max_len=5, use_inflated_groupnorm=False, flip_sin_to_cos=True, freq_shift=0,
)


def tiny_config(width=64, cross_dim=64, layers_per_block=1, heads=8):
    """A narrow UNet with the full stage-2 topology, for fixtures that must run in seconds."""
    cfg = dict(SD15_STAGE2_CONFIG)
    cfg.update(block_out_channels=(width, 2 * width, 4 * width, 4 * width), cross_attention_dim=cross_dim,
               layers_per_block=layers_per_block, attention_head_dim=heads, motion_num_attention_heads=heads)
    return cfg


# ------------------------------------------------------------------------------------------------
# leaf ops

def conv_frames(x, w, b, stride=1, padding=1):
    """InflatedConv3d.forward, src/models/resnet.py:10-18: a 2-D conv applied to every frame."""
    bsz, c, f, h, wd = x.shape
    y = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, wd), w, b, stride=stride, padding=padding)
    return y.reshape(bsz, f, y.shape[1], y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)


def group_norm_cross_frame(x, w, b, groups, eps):
    """torch.nn.GroupNorm on the 5-D (b, C, f, H, W) tensor: statistics over (C/groups, f, H, W), i.e.
    ACROSS frames — resnet.py:142-145,161-164,185,196 and unet.py:247-249,455 with
    use_inflated_groupnorm=False (unet.py:79; configs/testing.yaml does not set it)."""
    return F.group_norm(x, groups, w, b, eps)


def group_norm_per_frame(x4, w, b, groups, eps):
    """GroupNorm on the (b f, C, H, W) view: attention.py:284,328 / motion_module.py:119,162 (eps 1e-6)."""
    return F.group_norm(x4, groups, w, b, eps)


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0):
    """diffusers 0.24.0 models/embeddings.py get_timestep_embedding as instantiated by
    Timesteps(320, True, 0) at unet.py:100 and called at :383.  [third-party, parity unpinned]"""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    ang = t.reshape(-1, 1).to(torch.float32) * torch.exp(exponent)[None, :]
    sin, cos = torch.sin(ang), torch.cos(ang)
    return torch.cat([cos, sin], dim=-1) if flip_sin_to_cos else torch.cat([sin, cos], dim=-1)


def time_embedding_mlp(sd, t_emb):
    """diffusers TimestepEmbedding(320, 1280): linear_1 -> SiLU -> linear_2 (unet.py:103,389).
    [third-party, parity unpinned]"""
    h = F.linear(t_emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
    return F.linear(F.silu(h), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])


def attention_core(q, k, v, heads, mask=None):
    """CrossAttention.reshape_heads_to_batch_dim + _attention + reshape_batch_dim_to_heads,
    attention.py:93-105,170-199: softmax(q k^T * d^-0.5 [+ mask]) v per head, no upcast.  `mask`: additive
    (bsz, lq, lk), the same for every head (attention.py:187-188; the stage-1 prior passes 0 / -10000 entries)."""
    bsz, lq, c = q.shape
    d = c // heads

    def split(x):
        return x.reshape(bsz, x.shape[1], heads, d).permute(0, 2, 1, 3)

    qh, kh, vh = split(q), split(k), split(v)
    scores = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
    if mask is not None:
        scores = scores + mask[:, None]
    probs = torch.softmax(scores, dim=-1)
    return torch.matmul(probs, vh).permute(0, 2, 1, 3).reshape(bsz, lq, c)


def cross_attention(sd, p, x, ctx, heads):
    """CrossAttention.forward attention.py:113-168 (bias-free to_q/k/v, to_out.0 with bias)."""
    src = x if ctx is None else ctx
    q = F.linear(x, sd[p + "to_q.weight"])
    k = F.linear(src, sd[p + "to_k.weight"])
    v = F.linear(src, sd[p + "to_v.weight"])
    return F.linear(attention_core(q, k, v, heads), sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def feed_forward_geglu(sd, p, x):
    """diffusers 0.24.0 FeedForward(dim, activation_fn="geglu") (attention.py:434, motion_module.py:231):
    net.0 = GEGLU(dim, 4 dim): proj -> chunk(2) -> hidden * gelu(gate) (exact erf GELU); net.2 = Linear.
    [third-party, parity unpinned]"""
    hg = F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"])
    hidden, gate = hg.chunk(2, dim=-1)
    return F.linear(hidden * F.gelu(gate), sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def layer_norm(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


# ------------------------------------------------------------------------------------------------
# blocks

def resnet_block(sd, p, x, temb, groups, eps):
    """ResnetBlock3D.forward resnet.py:182-212 (time_embedding_norm "default", output_scale_factor 1)."""
    h = F.silu(group_norm_cross_frame(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], groups, eps))
    h = conv_frames(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"])
    t = F.linear(F.silu(temb), sd[p + "time_emb_proj.weight"], sd[p + "time_emb_proj.bias"])
    h = h + t[:, :, None, None, None]
    h = F.silu(group_norm_cross_frame(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"], groups, eps))
    h = conv_frames(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"])
    if p + "conv_shortcut.weight" in sd:
        x = conv_frames(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"], padding=0)
    return x + h


def spatial_transformer(sd, p, x, ctx, heads, groups):
    """Transformer3DModel.forward attention.py:318-365 + BasicTransformerBlock.forward :479-526
    (use_linear_projection False, no SC-attn, no attn_temp: configs/testing.yaml:4-5)."""
    bsz, c, f, h, w = x.shape
    x4 = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, w)
    y = group_norm_per_frame(x4, sd[p + "norm.weight"], sd[p + "norm.bias"], groups, 1e-6)
    y = F.conv2d(y, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    tok = y.permute(0, 2, 3, 1).reshape(bsz * f, h * w, c)
    b = p + "transformer_blocks.0."
    tok = cross_attention(sd, b + "attn1.", layer_norm(sd, b + "norm1.", tok), None, heads) + tok
    tok = cross_attention(sd, b + "attn2.", layer_norm(sd, b + "norm2.", tok), ctx, heads) + tok
    tok = feed_forward_geglu(sd, b + "ff.", layer_norm(sd, b + "norm3.", tok)) + tok
    y = tok.reshape(bsz * f, h, w, c).permute(0, 3, 1, 2)
    y = F.conv2d(y, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + x4
    return y.reshape(bsz, f, c, h, w).permute(0, 2, 1, 3, 4)


def sinusoid_table(d_model, max_len):
    """PositionalEncoding.__init__ motion_module.py:258-263."""
    pos = torch.arange(max_len, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def temporal_self_attention(sd, p, x_norm, frames, heads):
    """VersatileAttention.forward motion_module.py:294-354 in Temporal/self mode: regroup tokens so the
    sequence axis is the frame axis, add pe[:, :f] to the (already layer-normed) input, attend, regroup back."""
    bf, n, c = x_norm.shape
    bsz = bf // frames
    seq = x_norm.reshape(bsz, frames, n, c).permute(0, 2, 1, 3).reshape(bsz * n, frames, c)
    pe = sd[p + "pos_encoder.pe"] if (p + "pos_encoder.pe") in sd else None
    if pe is not None:
        seq = seq + pe[:, :frames]
    out = cross_attention(sd, p, seq, None, heads)
    return out.reshape(bsz, n, frames, c).permute(0, 2, 1, 3).reshape(bf, n, c)


def motion_module(sd, p, x, heads, groups, n_attn):
    """VanillaTemporalModule.forward motion_module.py:87-93 -> TemporalTransformer3DModel.forward :147-182
    (prior_state False) -> TemporalTransformerBlock.forward :234-246."""
    p = p + "temporal_transformer."
    bsz, c, f, h, w = x.shape
    x4 = x.permute(0, 2, 1, 3, 4).reshape(bsz * f, c, h, w)
    y = group_norm_per_frame(x4, sd[p + "norm.weight"], sd[p + "norm.bias"], groups, 1e-6)
    tok = y.permute(0, 2, 3, 1).reshape(bsz * f, h * w, c)
    tok = F.linear(tok, sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    b = p + "transformer_blocks.0."
    for i in range(n_attn):
        normed = layer_norm(sd, b + f"norms.{i}.", tok)
        tok = temporal_self_attention(sd, b + f"attention_blocks.{i}.", normed, f, heads) + tok
    tok = feed_forward_geglu(sd, b + "ff.", layer_norm(sd, b + "ff_norm.", tok)) + tok
    tok = F.linear(tok, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    y = tok.reshape(bsz * f, h, w, c).permute(0, 3, 1, 2) + x4
    return y.reshape(bsz, f, c, h, w).permute(0, 2, 1, 3, 4)


def downsample(sd, p, x):
    """Downsample3D.forward resnet.py:98-106: conv3x3 stride 2 padding 1."""
    return conv_frames(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd, p, x):
    """Upsample3D.forward resnet.py:47-80: nearest x(1,2,2) then conv3x3."""
    x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
    return conv_frames(x, sd[p + "conv.weight"], sd[p + "conv.bias"])


# ------------------------------------------------------------------------------------------------
# the UNet

def unet_forward(sd, cfg, sample, timestep, ctx):
    """UNet3DConditionModel.forward src/models/unet.py:322-463 with the block wiring of
    src/models/unet_blocks.py:384-427 (CrossAttnDownBlock3D), :499-531 (DownBlock3D), :272-280 (mid),
    :748-777 (UpBlock3D), :631-680 (CrossAttnUpBlock3D).
    sample (b, 9, f, H, W) fp32, timestep scalar / (b,) tensor, ctx (b*f, L, cross_dim) -> (b, 4, f, H, W)."""
    boc = cfg["block_out_channels"]
    heads, groups, eps = cfg["attention_head_dim"], cfg["norm_num_groups"], cfg["norm_eps"]
    mheads, n_attn = cfg["motion_num_attention_heads"], cfg["motion_attention_blocks"]
    lpb = cfg["layers_per_block"]
    bsz = sample.shape[0]

    t = torch.as_tensor(timestep)
    if t.dim() == 0:
        t = t[None]
    t = t.expand(bsz)
    emb = time_embedding_mlp(sd, timestep_embedding(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]))

    def has_motion(res):
        return cfg["use_motion_module"] and res in cfg["motion_module_resolutions"]

    h = conv_frames(sample, sd["conv_in.weight"], sd["conv_in.bias"])
    skips = [h]
    for i, kind in enumerate(cfg["down_block_types"]):
        pb = f"down_blocks.{i}."
        for j in range(lpb):
            h = resnet_block(sd, pb + f"resnets.{j}.", h, emb, groups, eps)
            if kind == "CrossAttnDownBlock3D":
                h = spatial_transformer(sd, pb + f"attentions.{j}.", h, ctx, heads, groups)
            if has_motion(2 ** i):
                h = motion_module(sd, pb + f"motion_modules.{j}.", h, mheads, groups, n_attn)
            skips.append(h)
        if i != len(boc) - 1:
            h = downsample(sd, pb + "downsamplers.0.", h)
            skips.append(h)

    h = resnet_block(sd, "mid_block.resnets.0.", h, emb, groups, eps)
    h = spatial_transformer(sd, "mid_block.attentions.0.", h, ctx, heads, groups)
    if cfg["use_motion_module"] and cfg["motion_module_mid_block"]:
        h = motion_module(sd, "mid_block.motion_modules.0.", h, mheads, groups, n_attn)
    h = resnet_block(sd, "mid_block.resnets.1.", h, emb, groups, eps)

    for i, kind in enumerate(cfg["up_block_types"]):
        pb = f"up_blocks.{i}."
        for j in range(lpb + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, pb + f"resnets.{j}.", h, emb, groups, eps)
            if kind == "CrossAttnUpBlock3D":
                h = spatial_transformer(sd, pb + f"attentions.{j}.", h, ctx, heads, groups)
            if has_motion(2 ** (3 - i)):
                h = motion_module(sd, pb + f"motion_modules.{j}.", h, mheads, groups, n_attn)
        if i != len(boc) - 1:
            h = upsample(sd, pb + "upsamplers.0.", h)

    h = F.silu(group_norm_cross_frame(h, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], groups, eps))
    return conv_frames(h, sd["conv_out.weight"], sd["conv_out.bias"])


# ------------------------------------------------------------------------------------------------
# DDIM + the CFG sampling loop   [scheduler = third-party arithmetic, parity unpinned]

class DDIMOracle:
    """diffusers 0.24.0 DDIMScheduler as the reference configures it: DDIMScheduler(beta_start=0.00085,
    beta_end=0.012, beta_schedule="linear") (stage2_batchtest_rcdms_model.py:247 + configs/testing.yaml:18-21),
    then steps_offset forced to 1 and clip_sample to False by the pipeline ctor (RCDMs_pipeline.py:84-109);
    defaults otherwise: num_train_timesteps 1000, set_alpha_to_one True, prediction_type "epsilon",
    timestep_spacing "leading", eta 0."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", num_train_timesteps=1000,
                 steps_offset=1, set_alpha_to_one=True):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise ValueError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n):
        """'leading' spacing: (arange(n) * (T // n)).round()[::-1] + steps_offset."""
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = (torch.arange(n) * ratio).flip(0).to(torch.int64) + self.steps_offset

    def step(self, eps, t, x):
        """DDIM eq. (12) with sigma = 0 and epsilon prediction; no clipping / thresholding."""
        prev_t = int(t) - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


class PNDMOracle:
    """diffusers 0.24.0 PNDMScheduler with skip_prk_steps=True (PLMS) — the other scheduler type the pipeline constructor
    accepts (RCDMs_pipeline.py:72-79), with steps_offset forced to 1 (:84-97); defaults otherwise: set_alpha_to_one False,
    epsilon prediction, "leading" spacing.  Third-party arithmetic, restated from the published class: parity unpinned.
    Written as the textbook recurrence (one list of kept predictions), independently of rcdms_amd/scheduler.py."""

    def __init__(self, beta_start=0.00085, beta_end=0.012, beta_schedule="linear", num_train_timesteps=1000, steps_offset=1):
        if beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise ValueError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.N, self.offset = num_train_timesteps, steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        self.n = n
        base = [i * (self.N // n) + self.offset for i in range(n)]
        self.timesteps = torch.tensor(list(reversed(base[:-1] + [base[-2], base[-1]])), dtype=torch.int64)
        self.kept, self.calls, self.first = [], 0, None

    def _prev(self, x, t, t_prev, e):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[t_prev] if t_prev >= 0 else self.alphas_cumprod[0]
        den = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        return (a_p / a_t) ** 0.5 * x - (a_p - a_t) * e / den

    def step(self, eps, t, x):
        t, dt = int(t), self.N // self.n
        k = self.calls
        self.calls += 1
        if k == 1:   # the repeated timestep: average with the first prediction, restart from the first sample
            return self._prev(self.first, t + dt, t, (eps + self.kept[-1]) / 2)
        self.kept = (self.kept + [eps])[-4:]
        e = self.kept
        if k == 0:
            self.first = x
            comb = e[-1]
        elif len(e) == 2:
            comb = (3 * e[-1] - e[-2]) / 2
        elif len(e) == 3:
            comb = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12
        else:
            comb = (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4]) / 24
        return self._prev(x, t, t - dt, comb)


def denoise_loop(sd, cfg, latents, mask, masked_latents, ctx, num_steps, guidance_scale, unet=None, callback=None, sched=None):
    """The hot loop of RCDMsPipeline.__call__, src/pipelines/RCDMs_pipeline.py:455-503, generalised to
    S stories (the reference hard-codes batch 1 :408 and 64x64 :476):
      latents (S,4,f,H,W); mask (reps*S,1,f,H,W); masked_latents (reps*S,4,f,H,W); ctx (reps*S*f, L, D)
      with reps = 2 when guidance_scale > 1 (uncond block first)."""
    unet = unet or (lambda x, t, c: unet_forward(sd, cfg, x, t, c))
    sched = sched or DDIMOracle()   # (sched: another oracle scheduler, e.g. PNDMOracle())
    sched.set_timesteps(num_steps)
    cfg_on = guidance_scale > 1.0
    x = latents * sched.init_noise_sigma
    for i, t in enumerate(sched.timesteps):
        model_in = torch.cat([x] * 2) if cfg_on else x
        model_in = torch.cat([model_in, mask, masked_latents], dim=1)
        eps = unet(model_in, t, ctx)
        if cfg_on:
            e_u, e_c = eps.chunk(2)
            eps = e_u + guidance_scale * (e_c - e_u)
        x = sched.step(eps, t, x)
        if callback is not None:
            callback(i, int(t), x)
    return x
