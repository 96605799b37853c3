"""Mirror of the reference's src/models/unet_blocks.py interface: the five 3-D block classes
(UNetMidBlock3DCrossAttn :172-280, CrossAttnDownBlock3D :283-427, DownBlock3D :430-531,
CrossAttnUpBlock3D :534-680, UpBlock3D :683-777) and the two factories (:13-169).

Here a block is a parameter container with the reference's child names (resnets / attentions /
motion_modules / downsamplers / upsamplers).  Inside UNet3DConditionModel.forward the layer order resnet ->
transformer -> motion module and the skip push/pop are executed by the launch planner
(rcdms_amd.engine.UNetProgram), where the skip concat is a buffer layout rather than a copy.  Called on its own, a
block's forward() has the reference's signature and return convention and runs its children's HIP forwards one after
the other (ResnetBlock3D / Transformer3DModel / VanillaTemporalModule / Downsample3D / Upsample3D): correct, not fast —
the fast path is the UNet's plan."""
import torch
from torch import nn

from .attention import Transformer3DModel
from .motion_module import get_motion_module
from .resnet import Downsample3D, ResnetBlock3D, Upsample3D

class _Block3D(nn.Module):
    """Shared construction: n layers of [resnet, optional transformer, optional motion module] + sampler."""
    has_cross_attention = False

    def _build(self, resnet_io, temb_channels, *, attn, heads, cross_attention_dim, eps, groups, act, scale_shift,
               output_scale_factor, pre_norm, dropout, inflated_gn, use_motion_module, motion_module_type,
               motion_module_kwargs, xf_kwargs, n_attn=None, n_motion=None):
        self.resnets = nn.ModuleList([ResnetBlock3D(
            in_channels=ci, out_channels=co, temb_channels=temb_channels, eps=eps, groups=groups, dropout=dropout,
            time_embedding_norm=scale_shift, non_linearity=act, output_scale_factor=output_scale_factor,
            pre_norm=pre_norm, use_inflated_groupnorm=inflated_gn) for ci, co in resnet_io])
        co = resnet_io[-1][1]
        n_attn = len(resnet_io) if n_attn is None else n_attn
        n_motion = len(resnet_io) if n_motion is None else n_motion
        if attn:
            self.attentions = nn.ModuleList([Transformer3DModel(
                heads, co // heads, in_channels=co, num_layers=1, cross_attention_dim=cross_attention_dim,
                norm_num_groups=groups, **xf_kwargs) for _ in range(n_attn)])
        self.motion_modules = nn.ModuleList([get_motion_module(
            in_channels=co, motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs)
            if use_motion_module else None for _ in range(n_motion)])
        self.gradient_checkpointing = False

    def _layer(self, j, h, temb, encoder_hidden_states, resnet=None):
        """One [resnet -> transformer -> motion module] layer (reference :412-417, :522-525, :665-670, :768-772)."""
        h = (resnet if resnet is not None else self.resnets[j])(h, temb)
        if self.has_cross_attention:
            h = self.attentions[j](h, encoder_hidden_states=encoder_hidden_states).sample
        motion = self.motion_modules[j] if j < len(self.motion_modules) else None
        if motion is not None:
            h = motion(h, temb, encoder_hidden_states=encoder_hidden_states)
        return h

    @staticmethod
    def _no_mask(attention_mask):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed by the stage-2 pipeline and has no HIP path")


def _xf(use_linear_projection, upcast_attention, cfa, ta, only_cross_attention=None):
    kw = dict(use_linear_projection=use_linear_projection, upcast_attention=upcast_attention,
              unet_use_cross_frame_attention=cfa, unet_use_temporal_attention=ta)
    if only_cross_attention is not None:
        kw["only_cross_attention"] = only_cross_attention
    return kw


class UNetMidBlock3DCrossAttn(_Block3D):
    has_cross_attention = True

    def __init__(self, in_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish",
                 resnet_groups: int = 32, resnet_pre_norm: bool = True, attn_num_head_channels=1,
                 output_scale_factor=1.0, cross_attention_dim=1280, dual_cross_attention=False,
                 use_linear_projection=False, upcast_attention=False, unet_use_cross_frame_attention=None,
                 unet_use_temporal_attention=None, use_inflated_groupnorm=None, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.attn_num_head_channels = attn_num_head_channels
        resnet_groups = resnet_groups if resnet_groups is not None else min(in_channels // 4, 32)
        self._build([(in_channels, in_channels)] * (num_layers + 1), temb_channels, attn=True,
                    heads=attn_num_head_channels, cross_attention_dim=cross_attention_dim, eps=resnet_eps,
                    groups=resnet_groups, act=resnet_act_fn, scale_shift=resnet_time_scale_shift,
                    output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm, dropout=dropout,
                    inflated_gn=use_inflated_groupnorm, use_motion_module=use_motion_module,
                    motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs,
                    xf_kwargs=_xf(use_linear_projection, upcast_attention, unet_use_cross_frame_attention,
                                  unet_use_temporal_attention), n_attn=num_layers, n_motion=num_layers)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None):
        """reference :272-280: resnets[0], then per layer transformer -> motion module -> resnet."""
        self._no_mask(attention_mask)
        h = self.resnets[0](hidden_states, temb)
        for j, attn in enumerate(self.attentions):
            h = attn(h, encoder_hidden_states=encoder_hidden_states).sample
            if self.motion_modules[j] is not None:
                h = self.motion_modules[j](h, temb, encoder_hidden_states=encoder_hidden_states)
            h = self.resnets[j + 1](h, temb)
        return h


class _DownBase(_Block3D):
    def _down(self, in_channels, out_channels, temb_channels, num_layers, add_downsample, downsample_padding, **kw):
        io = [(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)]
        self._build(io, temb_channels, **kw)
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, use_conv=True, out_channels=out_channels,
                                                        padding=downsample_padding, name="op")]) if add_downsample else None

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None, attention_mask=None):
        """reference :384-427 / :499-531: returns (hidden_states, output_states) — one skip tensor per layer and one
        after the downsampler."""
        self._no_mask(attention_mask)
        h, skips = hidden_states, ()
        for j in range(len(self.resnets)):
            h = self._layer(j, h, temb, encoder_hidden_states)
            skips += (h,)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                h = d(h)
            skips += (h,)
        return h, skips


class CrossAttnDownBlock3D(_DownBase):
    has_cross_attention = True

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default",
                 resnet_act_fn: str = "swish", resnet_groups: int = 32, resnet_pre_norm: bool = True,
                 attn_num_head_channels=1, cross_attention_dim=1280, output_scale_factor=1.0, downsample_padding=1,
                 add_downsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, unet_use_cross_frame_attention=None,
                 unet_use_temporal_attention=None, use_inflated_groupnorm=None, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.attn_num_head_channels = attn_num_head_channels
        self._down(in_channels, out_channels, temb_channels, num_layers, add_downsample, downsample_padding,
                   attn=True, heads=attn_num_head_channels, cross_attention_dim=cross_attention_dim, eps=resnet_eps,
                   groups=resnet_groups, act=resnet_act_fn, scale_shift=resnet_time_scale_shift,
                   output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm, dropout=dropout,
                   inflated_gn=use_inflated_groupnorm, use_motion_module=use_motion_module,
                   motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs,
                   xf_kwargs=_xf(use_linear_projection, upcast_attention, unet_use_cross_frame_attention,
                                 unet_use_temporal_attention, only_cross_attention))


class DownBlock3D(_DownBase):
    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, resnet_eps: float = 1e-6, resnet_time_scale_shift: str = "default",
                 resnet_act_fn: str = "swish", resnet_groups: int = 32, resnet_pre_norm: bool = True,
                 output_scale_factor=1.0, add_downsample=True, downsample_padding=1, use_inflated_groupnorm=None,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        self._down(in_channels, out_channels, temb_channels, num_layers, add_downsample, downsample_padding,
                   attn=False, heads=None, cross_attention_dim=None, eps=resnet_eps, groups=resnet_groups,
                   act=resnet_act_fn, scale_shift=resnet_time_scale_shift, output_scale_factor=output_scale_factor,
                   pre_norm=resnet_pre_norm, dropout=dropout, inflated_gn=use_inflated_groupnorm,
                   use_motion_module=use_motion_module, motion_module_type=motion_module_type,
                   motion_module_kwargs=motion_module_kwargs, xf_kwargs={})


class _UpBase(_Block3D):
    def _up(self, in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample, **kw):
        io = []
        for i in range(num_layers):
            skip = in_channels if i == num_layers - 1 else out_channels
            h = prev_output_channel if i == 0 else out_channels
            io.append((h + skip, out_channels))
        self._build(io, temb_channels, **kw)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, use_conv=True, out_channels=out_channels)]) \
            if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None, upsample_size=None,
                attention_mask=None):
        """reference :631-680 / :748-777: every layer pops the LAST skip tensor and concatenates it behind the hidden
        states on the channel axis, then resnet -> transformer -> motion module; the upsampler closes the block."""
        self._no_mask(attention_mask)
        h, skips = hidden_states, tuple(res_hidden_states_tuple)
        for j in range(len(self.resnets)):
            h = self._layer(j, torch.cat([h, skips[-1]], dim=1), temb, encoder_hidden_states)
            skips = skips[:-1]
        if self.upsamplers is not None:
            for u in self.upsamplers:
                h = u(h, upsample_size)
        return h


class CrossAttnUpBlock3D(_UpBase):
    has_cross_attention = True

    def __init__(self, in_channels: int, out_channels: int, prev_output_channel: int, temb_channels: int,
                 dropout: float = 0.0, num_layers: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, attn_num_head_channels=1, cross_attention_dim=1280,
                 output_scale_factor=1.0, add_upsample=True, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, unet_use_cross_frame_attention=None,
                 unet_use_temporal_attention=None, use_inflated_groupnorm=None, use_motion_module=None,
                 motion_module_type=None, motion_module_kwargs=None):
        super().__init__()
        if dual_cross_attention:
            raise NotImplementedError
        self.attn_num_head_channels = attn_num_head_channels
        self._up(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample,
                 attn=True, heads=attn_num_head_channels, cross_attention_dim=cross_attention_dim, eps=resnet_eps,
                 groups=resnet_groups, act=resnet_act_fn, scale_shift=resnet_time_scale_shift,
                 output_scale_factor=output_scale_factor, pre_norm=resnet_pre_norm, dropout=dropout,
                 inflated_gn=use_inflated_groupnorm, use_motion_module=use_motion_module,
                 motion_module_type=motion_module_type, motion_module_kwargs=motion_module_kwargs,
                 xf_kwargs=_xf(use_linear_projection, upcast_attention, unet_use_cross_frame_attention,
                               unet_use_temporal_attention, only_cross_attention))


class UpBlock3D(_UpBase):
    def __init__(self, in_channels: int, prev_output_channel: int, out_channels: int, temb_channels: int,
                 dropout: float = 0.0, num_layers: int = 1, resnet_eps: float = 1e-6,
                 resnet_time_scale_shift: str = "default", resnet_act_fn: str = "swish", resnet_groups: int = 32,
                 resnet_pre_norm: bool = True, output_scale_factor=1.0, add_upsample=True,
                 use_inflated_groupnorm=None, use_motion_module=None, motion_module_type=None,
                 motion_module_kwargs=None):
        super().__init__()
        self._up(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample,
                 attn=False, heads=None, cross_attention_dim=None, eps=resnet_eps, groups=resnet_groups,
                 act=resnet_act_fn, scale_shift=resnet_time_scale_shift, output_scale_factor=output_scale_factor,
                 pre_norm=resnet_pre_norm, dropout=dropout, inflated_gn=use_inflated_groupnorm,
                 use_motion_module=use_motion_module, motion_module_type=motion_module_type,
                 motion_module_kwargs=motion_module_kwargs, xf_kwargs={})


_DOWN = {"DownBlock3D": DownBlock3D, "CrossAttnDownBlock3D": CrossAttnDownBlock3D}
_UP = {"UpBlock3D": UpBlock3D, "CrossAttnUpBlock3D": CrossAttnUpBlock3D}


def _strip(name):
    return name[7:] if name.startswith("UNetRes") else name


def get_down_block(down_block_type, num_layers, in_channels, out_channels, temb_channels, add_downsample, resnet_eps,
                   resnet_act_fn, attn_num_head_channels, resnet_groups=None, cross_attention_dim=None,
                   downsample_padding=None, dual_cross_attention=False, use_linear_projection=False,
                   only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                   unet_use_cross_frame_attention=None, unet_use_temporal_attention=None, use_inflated_groupnorm=None,
                   use_motion_module=None, motion_module_type=None, motion_module_kwargs=None):
    kind = _strip(down_block_type)
    if kind not in _DOWN:
        raise ValueError(f"{down_block_type} does not exist.")
    kw = dict(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels, temb_channels=temb_channels,
              add_downsample=add_downsample, resnet_eps=resnet_eps, resnet_act_fn=resnet_act_fn,
              resnet_groups=resnet_groups, downsample_padding=downsample_padding,
              resnet_time_scale_shift=resnet_time_scale_shift, use_inflated_groupnorm=use_inflated_groupnorm,
              use_motion_module=use_motion_module, motion_module_type=motion_module_type,
              motion_module_kwargs=motion_module_kwargs)
    if kind == "CrossAttnDownBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnDownBlock3D")
        kw.update(cross_attention_dim=cross_attention_dim, attn_num_head_channels=attn_num_head_channels,
                  dual_cross_attention=dual_cross_attention, use_linear_projection=use_linear_projection,
                  only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
                  unet_use_cross_frame_attention=unet_use_cross_frame_attention,
                  unet_use_temporal_attention=unet_use_temporal_attention)
    return _DOWN[kind](**kw)


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, temb_channels,
                 add_upsample, resnet_eps, resnet_act_fn, attn_num_head_channels, resnet_groups=None,
                 cross_attention_dim=None, dual_cross_attention=False, use_linear_projection=False,
                 only_cross_attention=False, upcast_attention=False, resnet_time_scale_shift="default",
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None, use_inflated_groupnorm=None,
                 use_motion_module=None, motion_module_type=None, motion_module_kwargs=None):
    kind = _strip(up_block_type)
    if kind not in _UP:
        raise ValueError(f"{up_block_type} does not exist.")
    kw = dict(num_layers=num_layers, in_channels=in_channels, out_channels=out_channels,
              prev_output_channel=prev_output_channel, temb_channels=temb_channels, add_upsample=add_upsample,
              resnet_eps=resnet_eps, resnet_act_fn=resnet_act_fn, resnet_groups=resnet_groups,
              resnet_time_scale_shift=resnet_time_scale_shift, use_inflated_groupnorm=use_inflated_groupnorm,
              use_motion_module=use_motion_module, motion_module_type=motion_module_type,
              motion_module_kwargs=motion_module_kwargs)
    if kind == "CrossAttnUpBlock3D":
        if cross_attention_dim is None:
            raise ValueError("cross_attention_dim must be specified for CrossAttnUpBlock3D")
        kw.update(cross_attention_dim=cross_attention_dim, attn_num_head_channels=attn_num_head_channels,
                  dual_cross_attention=dual_cross_attention, use_linear_projection=use_linear_projection,
                  only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
                  unet_use_cross_frame_attention=unet_use_cross_frame_attention,
                  unet_use_temporal_attention=unet_use_temporal_attention)
    return _UP[kind](**kw)
