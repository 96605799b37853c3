"""GPU parity of the five mirrored BLOCK classes' standalone forward() (src/models/unet_blocks.py: UNetMidBlock3DCrossAttn
:272-280, CrossAttnDownBlock3D :384-427, DownBlock3D :499-531, CrossAttnUpBlock3D :631-680, UpBlock3D :748-777) against
the same wiring expressed with the oracle's functions (oracle/unet_oracle.py: resnet_block, spatial_transformer,
motion_module, downsample, upsample — each pinned to the reference classes by the block fixtures).  Width 64, 8x8
latents: the blocks run their children's HIP forwards one after the other."""
import pytest
import torch

from oracle import unet_oracle as O
from rcdms_amd import synth
from tests.test_hip_leaf import _module
from tests.test_hip_unet import DEV, check

pytestmark = pytest.mark.gpu

HEADS, GROUPS, EPS, TED, CTX = 8, 32, 1e-5, 128, 64
MOTION = dict(use_motion_module=True, motion_module_type="Vanilla", motion_module_kwargs=dict(
    num_attention_heads=8, num_transformer_block=1, attention_block_types=["Temporal_Self", "Temporal_Self"],
    temporal_position_encoding=True, temporal_position_encoding_max_len=5, temporal_attention_dim_div=1))
COMMON = dict(temb_channels=TED, resnet_eps=EPS, resnet_act_fn="silu", resnet_groups=GROUPS, use_inflated_groupnorm=False)
XATTN = dict(cross_attention_dim=CTX, attn_num_head_channels=HEADS, unet_use_cross_frame_attention=False,
             unet_use_temporal_attention=False)


def _inputs(b, c, hw, seed):
    x = synth.normal_tensor("blk.x", (b, c, 5, hw, hw), seed)
    temb = synth.normal_tensor("blk.temb", (b, TED), seed)
    ctx = synth.normal_tensor("blk.ctx", (b * 5, 13, CTX), seed)
    return x, temb, ctx


def _layer(sd, p, j, h, temb, ctx, attn):
    h = O.resnet_block(sd, f"{p}resnets.{j}.", h, temb, GROUPS, EPS)
    if attn:
        h = O.spatial_transformer(sd, f"{p}attentions.{j}.", h, ctx, HEADS, GROUPS)
    return O.motion_module(sd, f"{p}motion_modules.{j}.", h, HEADS, GROUPS, 2)


@pytest.mark.parametrize("attn", [True, False])
def test_down_block_forward(hiplib, attn):
    from src.models.unet_blocks import CrossAttnDownBlock3D, DownBlock3D
    kw = dict(in_channels=64, out_channels=128, num_layers=2, add_downsample=True, downsample_padding=1, **COMMON, **MOTION)
    m, sd = _module(CrossAttnDownBlock3D if attn else DownBlock3D, 31, **kw, **(XATTN if attn else {}))
    x, temb, ctx = _inputs(2, 64, 8, 9)
    with torch.no_grad():
        y, skips = m(x.to(DEV), temb.to(DEV), encoder_hidden_states=ctx.to(DEV))
    h, ref_skips = x, []
    for j in range(2):
        h = _layer(sd, "", j, h, temb, ctx, attn)
        ref_skips.append(h)
    h = O.downsample(sd, "downsamplers.0.", h)
    ref_skips.append(h)
    assert len(skips) == 3 and y.shape == (2, 128, 5, 4, 4)
    for k, (a, r) in enumerate(zip(skips, ref_skips)):
        check(a, r, 2.5e-3, 2.5e-3, f"{'CrossAttn' if attn else ''}DownBlock3D skip {k}")
    assert torch.equal(y, skips[-1])


@pytest.mark.parametrize("attn", [True, False])
def test_up_block_forward(hiplib, attn):
    from src.models.unet_blocks import CrossAttnUpBlock3D, UpBlock3D
    kw = dict(in_channels=64, out_channels=128, prev_output_channel=128, num_layers=3, add_upsample=True, **COMMON, **MOTION)
    m, sd = _module(CrossAttnUpBlock3D if attn else UpBlock3D, 32, **kw, **(XATTN if attn else {}))
    x, temb, ctx = _inputs(2, 128, 8, 10)
    skips = tuple(synth.normal_tensor(f"blk.skip{k}", (2, c, 5, 8, 8), 10) for k, c in enumerate((64, 128, 128)))
    with torch.no_grad():
        y = m(x.to(DEV), tuple(s.to(DEV) for s in skips), temb.to(DEV), encoder_hidden_states=ctx.to(DEV))
    h = x
    for j in range(3):
        h = _layer(sd, "", j, torch.cat([h, skips[2 - j]], dim=1), temb, ctx, attn)
    ref = O.upsample(sd, "upsamplers.0.", h)
    assert y.shape == (2, 128, 5, 16, 16)
    check(y, ref, 3e-3, 3.5e-3, f"{'CrossAttn' if attn else ''}UpBlock3D")


def test_mid_block_forward(hiplib):
    from src.models.unet_blocks import UNetMidBlock3DCrossAttn
    m, sd = _module(UNetMidBlock3DCrossAttn, 33, in_channels=64, num_layers=1, use_motion_module=False, **COMMON, **XATTN)
    x, temb, ctx = _inputs(2, 64, 8, 11)
    with torch.no_grad():
        y = m(x.to(DEV), temb.to(DEV), encoder_hidden_states=ctx.to(DEV))
    h = O.resnet_block(sd, "resnets.0.", x, temb, GROUPS, EPS)
    h = O.spatial_transformer(sd, "attentions.0.", h, ctx, HEADS, GROUPS)
    ref = O.resnet_block(sd, "resnets.1.", h, temb, GROUPS, EPS)
    check(y, ref, 1.8e-3, 2.1e-3, "UNetMidBlock3DCrossAttn")
