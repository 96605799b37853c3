"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU fp32 functional restatement of the reference's stage-1 frame-prior
transformer, `MyPriorTransformer.forward` (src/models/myprior_transformer.py:275-411): flat state dict
-> tensor.  Pinned by tests/golden/prior_*.npz, minted from the reference class itself (oracle/make_golden.py --only
prior).  Third-party arithmetic restated, not pinned by the reference (diffusers 0.24.0, requirements.txt:12):
Timesteps / TimestepEmbedding (shared with the UNet oracle) and FeedForward(activation_fn="gelu") = Linear -> exact
GELU -> Linear.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this file."""
import torch
import torch.nn.functional as F

from . import unet_oracle as O


def feed_forward_gelu(sd, p, x):
    """diffusers FeedForward(dim, activation_fn="gelu") (attention.py:434 via myprior_transformer.py:155): net.0 =
    GELU(dim, 4 dim) = proj + exact gelu, net.2 = Linear.  [third-party, parity unpinned]"""
    h = F.gelu(F.linear(x, sd[p + "net.0.proj.weight"], sd[p + "net.0.proj.bias"]))
    return F.linear(h, sd[p + "net.2.weight"], sd[p + "net.2.bias"])


def self_attention_biased(sd, p, x, heads, mask):
    """CrossAttention.forward attention.py:113-168 with attention_bias=True (to_q/k/v carry biases) and the additive
    mask of :187-188."""
    q = F.linear(x, sd[p + "to_q.weight"], sd[p + "to_q.bias"])
    k = F.linear(x, sd[p + "to_k.weight"], sd[p + "to_k.bias"])
    v = F.linear(x, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    return F.linear(O.attention_core(q, k, v, heads, mask), sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])


def prior_block(sd, p, x, heads, mask):
    """BasicTransformerBlock.forward attention.py:479-526 as the prior builds it (myprior_transformer.py:149-159):
    no cross-attention (attn2 is None), gelu feed-forward, attention biases."""
    x = self_attention_biased(sd, p + "attn1.", O.layer_norm(sd, p + "norm1.", x), heads, mask) + x
    return feed_forward_gelu(sd, p + "ff.", O.layer_norm(sd, p + "norm3.", x)) + x


def prior_motion_module(sd, p, x, heads, n_attn, frames=5):
    """VanillaTemporalModule.forward motion_module.py:87-93 -> TemporalTransformer3DModel.forward :147-182 with
    prior_state=True: LayerNorm (`prior_norm`) instead of the per-frame GroupNorm, tokens stay (b f, n, c),
    video_length is the literal 5 (:150), residual added after proj_out (:172-174)."""
    p = p + "temporal_transformer."
    tok = F.linear(O.layer_norm(sd, p + "prior_norm.", x), sd[p + "proj_in.weight"], sd[p + "proj_in.bias"])
    b = p + "transformer_blocks.0."
    for i in range(n_attn):
        normed = O.layer_norm(sd, b + f"norms.{i}.", tok)
        tok = O.temporal_self_attention(sd, b + f"attention_blocks.{i}.", normed, frames, heads) + tok
    tok = O.feed_forward_geglu(sd, b + "ff.", O.layer_norm(sd, b + "ff_norm.", tok)) + tok
    return F.linear(tok, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"]) + x


def prior_forward(sd, cfg, hidden_states, timestep, proj_embedding, encoder_hidden_states, proj_embedding1, mask_label,
                  attention_mask=None):
    """cfg: dict(num_attention_heads, attention_head_dim, num_layers, motion_heads, motion_attn).  Shapes as
    myprior_transformer.py:275-300: hidden_states / proj_embedding / proj_embedding1 / mask_label (B, E),
    encoder_hidden_states (B, T, E), attention_mask (B, T) of 0/1 -> (B, clip_embed_dim)."""
    f32 = lambda t: t.to(torch.float32)
    sd = {k: f32(v) for k, v in sd.items()}
    heads = cfg["num_attention_heads"]
    inner = heads * cfg["attention_head_dim"]
    B = hidden_states.shape[0]
    t = torch.as_tensor(timestep).reshape(-1).to(torch.float32).expand(B)
    temb = O.timestep_embedding(t, inner)                                          # Timesteps(inner, True, 0)  :112
    temb = F.linear(F.silu(F.linear(temb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])),
                    sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
    lin = lambda name, x: F.linear(f32(x), sd[name + ".weight"], sd[name + ".bias"])
    seq = torch.cat([lin("encoder_hidden_states_proj", encoder_hidden_states),          # :337
                     lin("embedding_proj", proj_embedding)[:, None],                    # :332
                     lin("embedding_proj1", proj_embedding1)[:, None],                  # :333
                     lin("embedding_proj2", mask_label)[:, None],                       # :334
                     temb[:, None],
                     lin("proj_in", hidden_states)[:, None],                            # :344
                     sd["prd_embedding"].expand(B, -1, -1)], dim=1)                     # :379-382
    seq = seq + sd["positional_embedding"]                                              # :391
    L = seq.shape[1]
    mask = None
    if attention_mask is not None:                                                      # :393-397
        pad = (1.0 - f32(attention_mask)) * -10000.0
        pad = F.pad(pad, (0, L - pad.shape[1]), value=0.0)
        mask = pad[:, None, :] + torch.full((L, L), -10000.0).triu_(1)[None]
    for i in range(cfg["num_layers"]):                                                  # :401-402, blocks interleaved
        seq = prior_block(sd, f"transformer_blocks.{2 * i}.", seq, heads, mask)
        seq = prior_motion_module(sd, f"transformer_blocks.{2 * i + 1}.", seq, cfg["motion_heads"], cfg["motion_attn"])
    seq = O.layer_norm(sd, "norm_out.", seq)[:, -1]                                     # :404-406
    return F.linear(seq, sd["proj_to_clip_embeddings.weight"], sd["proj_to_clip_embeddings.bias"])


def prior_denoise_loop(sd, cfg, scheduler, latents, proj_embedding, encoder_hidden_states, proj_embedding1, mask_label,
                       attention_mask, num_steps, guidance_scale, noise):
    """The sampling loop of Seq_Inpaint_Prior_Pipeline.__call__ (src/pipelines/prior_pipeline.py:293-344) with the
    scheduler's per-step noise supplied (`noise` (T, n, E)) instead of drawn: duplicate the latents for CFG (:314),
    prior forward (:316-325), CFG combine (:327-333), scheduler.step with prev_timestep = the next timestep (:335-344).
    `scheduler`: rcdms_amd.scheduler.UnCLIPScheduler (diffusers arithmetic restated; parity unpinned)."""
    scheduler.set_timesteps(num_steps)
    ts = scheduler.timesteps.tolist()
    cfg_on = guidance_scale > 1.0
    lat = latents.float() * scheduler.init_noise_sigma
    for i, t in enumerate(ts):
        x = torch.cat([lat] * 2) if cfg_on else lat
        pred = prior_forward(sd, cfg, x, t, proj_embedding, encoder_hidden_states, proj_embedding1, mask_label,
                             attention_mask)
        if cfg_on:
            u, c = pred.chunk(2)
            pred = u + guidance_scale * (c - u)
        prev_t = ts[i + 1] if i + 1 < len(ts) else None
        lat = scheduler.step(pred, t, lat, prev_timestep=prev_t, noise=noise[i]).prev_sample
    return lat
