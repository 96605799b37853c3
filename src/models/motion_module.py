"""Mirror of the reference's src/models/motion_module.py interface: VanillaTemporalModule (:53-93),
TemporalTransformer3DModel (:96-182), TemporalTransformerBlock (:185-246), PositionalEncoding (:249-267),
VersatileAttention (:270-354).  Parameter holders; forward() of the module runs on the HIP path."""
import math
from dataclasses import dataclass

import torch
from torch import nn

from rcdms_amd import engine
from .attention import CrossAttention, FeedForward


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


@dataclass
class TemporalTransformer3DModelOutput:
    sample: torch.Tensor


def get_motion_module(in_channels, motion_module_type: str, motion_module_kwargs: dict, prior_state=False):
    if motion_module_type != "Vanilla":
        raise ValueError
    return VanillaTemporalModule(in_channels=in_channels, prior_state=prior_state, **motion_module_kwargs)


class PositionalEncoding(nn.Module):
    """Fixed sinusoid table registered as the (persistent) buffer `pe`, shape (1, max_len, d_model)."""

    def __init__(self, d_model, dropout=0.0, max_len=24):
        super().__init__()
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        freq = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * (-math.log(10000.0) / d_model))
        table = torch.zeros(1, max_len, d_model)
        table[0, :, 0::2] = torch.sin(pos * freq)
        table[0, :, 1::2] = torch.cos(pos * freq)
        self.register_buffer("pe", table)


class VersatileAttention(CrossAttention):
    def __init__(self, attention_mode=None, cross_frame_attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=24, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert attention_mode == "Temporal"
        self.attention_mode = attention_mode
        self.is_cross_attention = kwargs.get("cross_attention_dim") is not None
        if self.is_cross_attention:
            raise NotImplementedError("Temporal_Cross attention blocks are not used by configs/testing.yaml")
        self.pos_encoder = PositionalEncoding(kwargs["query_dim"], max_len=temporal_position_encoding_max_len) \
            if temporal_position_encoding else None

    def extra_repr(self):
        return f"(Module Info) Attention_Mode: {self.attention_mode}, Is_Cross_Attention: {self.is_cross_attention}"


class TemporalTransformerBlock(nn.Module):
    def __init__(self, dim, num_attention_heads, attention_head_dim, attention_block_types=("Temporal_Self", "Temporal_Self"),
                 dropout=0.0, norm_num_groups=32, cross_attention_dim=768, activation_fn="geglu", attention_bias=False,
                 upcast_attention=False, cross_frame_attention_mode=None, temporal_position_encoding=False,
                 temporal_position_encoding_max_len=24):
        super().__init__()
        self.attention_blocks = nn.ModuleList([VersatileAttention(
            attention_mode=name.split("_")[0], cross_attention_dim=cross_attention_dim if name.endswith("_Cross") else None,
            query_dim=dim, heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
            upcast_attention=upcast_attention, cross_frame_attention_mode=cross_frame_attention_mode,
            temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len) for name in attention_block_types])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in attention_block_types])
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.ff_norm = nn.LayerNorm(dim)


class TemporalTransformer3DModel(nn.Module):
    def __init__(self, in_channels, num_attention_heads, attention_head_dim, num_layers,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), dropout=0.0, norm_num_groups=32,
                 cross_attention_dim=768, activation_fn="geglu", attention_bias=False, upcast_attention=False,
                 cross_frame_attention_mode=None, temporal_position_encoding=False, temporal_position_encoding_max_len=24):
        super().__init__()
        inner = num_attention_heads * attention_head_dim
        if num_layers != 1 or inner != in_channels:
            raise NotImplementedError("TemporalTransformer3DModel: 1 transformer block, inner_dim == in_channels")
        self.num_attention_heads, self.n_attn, self.norm_num_groups = num_attention_heads, len(attention_block_types), norm_num_groups
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.prior_norm = nn.LayerNorm(in_channels)  # stage-1 only; kept so the state-dict keys match
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([TemporalTransformerBlock(
            dim=inner, num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
            attention_block_types=attention_block_types, dropout=dropout, norm_num_groups=norm_num_groups,
            cross_attention_dim=cross_attention_dim, activation_fn=activation_fn, attention_bias=attention_bias,
            upcast_attention=upcast_attention, cross_frame_attention_mode=cross_frame_attention_mode,
            temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len)])
        self.proj_out = nn.Linear(inner, in_channels)


class VanillaTemporalModule(nn.Module):
    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=2,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), cross_frame_attention_mode=None,
                 temporal_position_encoding=False, temporal_position_encoding_max_len=24,
                 temporal_attention_dim_div=1, zero_initialize=True, prior_state=False):
        super().__init__()
        if temporal_attention_dim_div != 1:
            raise NotImplementedError("VanillaTemporalModule: temporal_attention_dim_div 1 only (configs/testing.yaml)")
        self.prior_state = prior_state
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels=in_channels, num_attention_heads=num_attention_heads,
            attention_head_dim=in_channels // num_attention_heads // temporal_attention_dim_div,
            num_layers=num_transformer_block, attention_block_types=attention_block_types,
            cross_frame_attention_mode=cross_frame_attention_mode, temporal_position_encoding=temporal_position_encoding,
            temporal_position_encoding_max_len=temporal_position_encoding_max_len)
        if zero_initialize:
            zero_module(self.temporal_transformer.proj_out)

    def forward(self, input_tensor, temb=None, encoder_hidden_states=None, attention_mask=None, anchor_frame_idx=None):
        if self.prior_state:
            raise NotImplementedError("prior_state motion modules run fused inside MyPriorTransformer on the HIP path")
        tt = self.temporal_transformer
        return engine.run_block("motion", self.state_dict(), input_tensor, heads=tt.num_attention_heads,
                                n_attn=tt.n_attn, groups=tt.norm_num_groups)
