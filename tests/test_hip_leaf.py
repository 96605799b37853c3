"""GPU parity of the mirrored LEAF classes' standalone forward() — CrossAttention (attention.py:113-168),
BasicTransformerBlock (:479-526), InflatedGroupNorm (resnet.py:21-29), Mish (resnet.py:215-217) — against the oracle's
restatement of the same reference functions (oracle/unet_oracle.py, itself pinned to the reference classes by the block
fixtures), plus the caller-visible context-caching behaviour of UNet3DConditionModel.forward / RCDMsPipeline.denoise."""
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_oracle as O
from rcdms_amd import synth
from tests.test_hip_unet import DEV, build, check

pytestmark = pytest.mark.gpu


def _module(cls, seed, **kw):
    with torch.device("meta"):
        m = cls(**kw)
    sd = synth.procedural_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m = m.to_empty(device="cpu")
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


@pytest.mark.parametrize("cross", [False, True])
@pytest.mark.parametrize("heads,dim_head,Lq", [(8, 40, 256), (8, 8, 64)])
def test_cross_attention_forward(hiplib, cross, heads, dim_head, Lq):
    from src.models.attention import CrossAttention
    C = heads * dim_head
    m, sd = _module(CrossAttention, 21, query_dim=C, cross_attention_dim=64 if cross else None, heads=heads,
                    dim_head=dim_head)
    x = synth.normal_tensor("leaf.x", (3, Lq, C), 5)
    ctx = synth.normal_tensor("leaf.ctx", (3, 13, 64), 5) if cross else None
    with torch.no_grad():
        y = m(x.to(DEV), encoder_hidden_states=ctx.to(DEV) if cross else None)
    ref = O.cross_attention(sd, "", x, ctx, heads)
    check(y, ref, 2e-3, 8e-3, f"CrossAttention.forward cross={cross} d={dim_head}")


@pytest.mark.parametrize("cross_dim", [64, None])
def test_basic_transformer_block_forward(hiplib, cross_dim):
    from src.models.attention import BasicTransformerBlock
    heads, dh, Lq = 8, 8, 64
    C = heads * dh
    m, sd = _module(BasicTransformerBlock, 22, dim=C, num_attention_heads=heads, attention_head_dim=dh,
                    cross_attention_dim=cross_dim, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False)
    x = synth.normal_tensor("leafb.x", (10, Lq, C), 6)
    ctx = synth.normal_tensor("leafb.ctx", (10, 13, cross_dim), 6) if cross_dim else None
    with torch.no_grad():
        y = m(x.to(DEV), encoder_hidden_states=ctx.to(DEV) if cross_dim else None)
    tok = O.cross_attention(sd, "attn1.", O.layer_norm(sd, "norm1.", x), None, heads) + x
    if cross_dim:
        tok = O.cross_attention(sd, "attn2.", O.layer_norm(sd, "norm2.", tok), ctx, heads) + tok
    ref = O.feed_forward_geglu(sd, "ff.", O.layer_norm(sd, "norm3.", tok)) + tok
    check(y, ref, 2e-3, 8e-3, f"BasicTransformerBlock.forward cross_dim={cross_dim}")


def test_inflated_groupnorm_forward(hiplib):
    from src.models.resnet import InflatedGroupNorm
    m, sd = _module(InflatedGroupNorm, 23, num_groups=32, num_channels=64, eps=1e-6)
    x = synth.normal_tensor("leafg.x", (2, 64, 5, 8, 8), 7) * 3.0 + 0.5
    with torch.no_grad():
        y = m(x.to(DEV))
    x4 = x.permute(0, 2, 1, 3, 4).reshape(10, 64, 8, 8)
    ref = O.group_norm_per_frame(x4, sd["weight"], sd["bias"], 32, 1e-6).reshape(2, 5, 64, 8, 8).permute(0, 2, 1, 3, 4)
    check(y, ref, 1e-3, 4e-3, "InflatedGroupNorm.forward")


def test_mish_forward(hiplib):
    from src.models.resnet import Mish
    x = torch.linspace(-30, 30, 4099)
    y = Mish()(x.to(DEV)).cpu()
    ref = x * torch.tanh(F.softplus(x))
    assert (y - ref).abs().max().item() <= 2e-6 * 30


def test_geglu_half_order_known_answer(hiplib):
    """diffusers GEGLU: `hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)` — the FIRST half of the projection is
    the linear branch.  Known answer independent of any restatement: hidden rows = identity, gate rows = 0 with bias 10
    (gelu(10) = 10 to fp32) -> out = 10 x; with the halves swapped it would be gelu(x) * 10-independent garbage."""
    from rcdms_amd import hip
    C, M = 64, 128
    w = torch.zeros(2 * C, C)
    w[:C] = torch.eye(C)
    b = torch.zeros(2 * C)
    b[C:] = 10.0
    wd = torch.empty(2 * C, C, dtype=torch.float16, device=DEV)
    bd = torch.empty(2 * C, dtype=torch.float32, device=DEV)
    w_dev, b_dev = w.to(DEV), b.to(DEV)            # keep the device copies alive across the asynchronous pack
    hip.pack_geglu_rows(w_dev.data_ptr(), b_dev.data_ptr(), 2 * C, C, wd.data_ptr(), bd.data_ptr())
    x = synth.normal_tensor("geglu.x", (M, C), 8).to(DEV).half()
    out = torch.empty(M, C, dtype=torch.float16, device=DEV)
    d = hip.GemmDesc(M, 2 * C, C, C, C, 0, hip.EPI_BIAS | hip.EPI_GEGLU, 1, 0, 1.0, 1)
    hip.gemm(d, x.data_ptr(), wd.data_ptr(), bd.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    assert torch.allclose(out.float(), 10.0 * x.float(), rtol=2e-3, atol=2e-3)


def test_context_is_not_cached_by_address(hiplib):
    """two different contexts of the same shape allocated one after the other (the caching allocator
    hands the second the first one's address, _version 0 again) must give different outputs — the context cache may
    only key on a tensor object the program itself keeps alive."""
    m = build("unet_tiny")
    s = synth.synthetic_story(stories=1, latent_hw=(16, 16), ctx_len=13, ctx_dim=64, seed=3)
    x = torch.cat([torch.cat([s["latents"]] * 2), s["mask"], s["masked_latents"]], dim=1).to(DEV)

    def run(seed):
        ctx = synth.normal_tensor("ctxcache", (10, 13, 64), seed).to(DEV)   # a fresh, transient tensor every call
        with torch.no_grad():
            return m(x, 981, ctx).clone(), ctx.data_ptr()
    y1, p1 = run(1)
    y2, p2 = run(2)
    y1b, _ = run(1)
    assert not torch.equal(y1, y2), f"second context ignored (addresses {p1:#x} / {p2:#x})"
    assert torch.equal(y1, y1b)
    # in-place edit of a context the caller keeps (version counter moves) is seen too
    ctx = synth.normal_tensor("ctxcache", (10, 13, 64), 1).to(DEV)
    with torch.no_grad():
        ya = m(x, 981, ctx).clone()
        ctx.mul_(0.5)
        yb = m(x, 981, ctx).clone()
    assert torch.equal(ya, y1) and not torch.equal(ya, yb)


def test_denoise_loop_reloads_context_every_story(hiplib):
    """RCDMsPipeline.denoise path: one DenoiseLoop object, two stories with different (transient) contexts."""
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    m = build("unet_tiny")
    s = synth.synthetic_story(stories=1, latent_hw=(16, 16), ctx_len=13, ctx_dim=64, seed=3)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1, clip_sample=False)
    loop = DenoiseLoop(m, 1, 5, 16, 16, 13, 2.0, sched, 2)
    outs = []
    for seed in (1, 2, 1):
        ctx = synth.normal_tensor("ctxcache", (10, 13, 64), seed).to(DEV)
        loop.load(s["latents"], s["mask"], s["masked_latents"], ctx)
        outs.append(loop.run().clone())
        del ctx
    assert not torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_unet_forward_validates_shapes(hiplib):
    m = build("unet_tiny")
    ctx = torch.zeros(10, 13, 64, device=DEV)
    with pytest.raises(ValueError, match="channels"):
        m(torch.zeros(2, 4, 5, 16, 16, device=DEV), 981, ctx)
    with pytest.raises(ValueError, match="frames"):
        m(torch.zeros(2, 9, 6, 16, 16, device=DEV), 981, torch.zeros(12, 13, 64, device=DEV))
    with pytest.raises(ValueError, match="encoder_hidden_states"):
        m(torch.zeros(2, 9, 5, 16, 16, device=DEV), 981, torch.zeros(9, 13, 64, device=DEV))


def test_fused_loop_refuses_clip_sample(hiplib):
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler
    m = build("unet_tiny")
    with pytest.raises(NotImplementedError, match="clip_sample"):
        DenoiseLoop(m, 1, 5, 16, 16, 13, 2.0, DDIMScheduler(), 2)     # DDIMScheduler() defaults to clip_sample=True
