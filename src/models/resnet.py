"""Mirror of the reference's src/models/resnet.py interface (class names, ctor kwargs, parameter names)
with forward() on the MI355X HIP path (rcdms_amd.engine).  No torch math here: these modules only own
parameters in the reference's layout; calling them on a CPU tensor raises (no CPU fallback).

Reference: InflatedConv3d resnet.py:10-18, InflatedGroupNorm :21-29, Upsample3D :32-80,
Downsample3D :83-106, ResnetBlock3D :109-212."""
import torch
import torch.nn as nn

from rcdms_amd import engine


def _sd(module):
    return {k: v for k, v in module.state_dict().items()}


class InflatedConv3d(nn.Conv2d):
    """2-D conv applied to every frame of a (b, c, f, h, w) tensor; 3x3 (stride 1|2, pad 1) and 1x1 supported."""

    def forward(self, x):
        k, s, p = self.kernel_size[0], self.stride[0], self.padding[0]
        if (k, p) not in ((3, 1), (1, 0)) or s not in (1, 2) or (k == 1 and s != 1):
            raise NotImplementedError(f"InflatedConv3d k={k} s={s} p={p} has no HIP kernel")
        return engine.run_block("conv", _sd(self), x, stride=s)


class InflatedGroupNorm(nn.GroupNorm):
    """Per-frame GroupNorm (reference resnet.py:21-29: "b c f h w -> (b f) c h w", nn.GroupNorm, back)."""

    def forward(self, x):
        if x.dim() != 5:
            raise ValueError(f"InflatedGroupNorm expects (b, c, f, h, w), got {tuple(x.shape)}")
        if not self.affine:
            raise NotImplementedError("InflatedGroupNorm without affine parameters has no HIP path")
        return engine.run_block("groupnorm", _sd(self), x, eps=self.eps, groups=self.num_groups).to(x.dtype)


class Upsample3D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv"):
        super().__init__()
        if use_conv_transpose or not use_conv:
            raise NotImplementedError("Upsample3D: only use_conv=True (as in the reference)")
        self.channels, self.out_channels, self.name = channels, out_channels or channels, name
        self.use_conv, self.use_conv_transpose = use_conv, use_conv_transpose
        self.conv = InflatedConv3d(channels, self.out_channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        if output_size is not None:
            raise NotImplementedError("forced upsample size is not supported on the HIP path (sizes must be /8)")
        assert hidden_states.shape[1] == self.channels
        return engine.run_block("up", _sd(self), hidden_states)


class Downsample3D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        if not use_conv or padding != 1:
            raise NotImplementedError("Downsample3D: only use_conv=True, padding=1 (as in the reference)")
        self.channels, self.out_channels, self.padding, self.name = channels, out_channels or channels, padding, name
        self.use_conv = use_conv
        self.conv = InflatedConv3d(channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        assert hidden_states.shape[1] == self.channels
        return engine.run_block("down", _sd(self), hidden_states)


class ResnetBlock3D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish",
                 time_embedding_norm="default", output_scale_factor=1.0, use_in_shortcut=None,
                 use_inflated_groupnorm=None):
        super().__init__()
        assert use_inflated_groupnorm is not None
        if time_embedding_norm != "default" or temb_channels is None:
            raise NotImplementedError("ResnetBlock3D: only time_embedding_norm='default' with a time embedding")
        if non_linearity not in ("swish", "silu"):
            raise NotImplementedError("ResnetBlock3D: only SiLU")
        out_channels = out_channels or in_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.groups, self.eps, self.output_scale_factor = groups, eps, output_scale_factor
        self.use_inflated_groupnorm = bool(use_inflated_groupnorm)
        self.time_embedding_norm, self.pre_norm, self.use_conv_shortcut = time_embedding_norm, True, conv_shortcut
        norm = InflatedGroupNorm if use_inflated_groupnorm else nn.GroupNorm
        self.norm1 = norm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = norm(num_groups=groups_out or groups, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = InflatedConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.use_in_shortcut = in_channels != out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = InflatedConv3d(in_channels, out_channels, kernel_size=1) if self.use_in_shortcut else None

    def forward(self, input_tensor, temb):
        if self.use_inflated_groupnorm:
            raise NotImplementedError("use_inflated_groupnorm=True is not built on the HIP path (reference default False)")
        return engine.run_block("resnet", _sd(self), input_tensor, temb=temb, eps=self.eps, groups=self.groups,
                                output_scale_factor=self.output_scale_factor)


class Mish(nn.Module):
    """resnet.py:215-217: x * tanh(softplus(x)).  Never instantiated by configs/testing.yaml (non_linearity "silu");
    kept callable for drop-in completeness through the elementwise HIP kernel rcdm_mish."""

    def forward(self, hidden_states):
        from rcdms_amd import hip
        if not hidden_states.is_cuda:
            raise hip.RcdmError("rcdms_amd runs on MI355X only: input tensor is not on a CUDA/HIP device (no CPU fallback)")
        x = hidden_states.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        hip.mish(x.data_ptr(), out.data_ptr(), x.numel())
        torch.cuda.current_stream(x.device).synchronize()
        return out.to(hidden_states.dtype)
