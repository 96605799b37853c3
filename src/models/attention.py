"""Mirror of the reference's src/models/attention.py interface: CrossAttention (attention.py:31-251),
Transformer3DModel (:254-365), BasicTransformerBlock (:368-526), plus the two diffusers 0.24.0 classes the
reference imports (FeedForward / GEGLU) restated as parameter holders.  forward() runs on the HIP path."""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn

from rcdms_amd import engine


@dataclass
class Transformer3DModelOutput:
    sample: torch.Tensor


class GEGLU(nn.Module):
    """diffusers GEGLU: proj -> chunk(2) -> hidden * gelu(gate).  Parameter holder (fused into the FF GEMM)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class GELU(nn.Module):
    """diffusers GELU(dim_in, dim_out): proj -> exact gelu.  Parameter holder (the gelu is a GEMM epilogue)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)


class FeedForward(nn.Module):
    """diffusers FeedForward(dim, activation_fn): net = [GEGLU | GELU (dim, 4 dim), Dropout, Linear(4 dim, dim)];
    "geglu" in the stage-2 UNet and every motion module, "gelu" in the stage-1 prior's transformer blocks."""

    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
        super().__init__()
        if activation_fn not in ("geglu", "gelu"):
            raise NotImplementedError("FeedForward: geglu (stage-2 UNet, motion modules) or gelu (stage-1 prior)")
        inner = int(dim * mult)
        act = GEGLU(dim, inner) if activation_fn == "geglu" else GELU(dim, inner)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])


class CrossAttention(nn.Module):
    """Parameter holder with the reference's names (to_q/to_k/to_v bias-free, to_out.0 with bias)."""

    def __init__(self, query_dim: int, cross_attention_dim: Optional[int] = None, heads: int = 8, dim_head: int = 64,
                 dropout: float = 0.0, bias=False, upcast_attention: bool = False, upcast_softmax: bool = False,
                 added_kv_proj_dim: Optional[int] = None, norm_num_groups: Optional[int] = None):
        super().__init__()
        if added_kv_proj_dim is not None or norm_num_groups is not None:
            raise NotImplementedError("CrossAttention: added_kv / group_norm are unused by the reference's configs")
        inner = dim_head * heads
        self.heads, self.scale = heads, dim_head ** -0.5
        self.sliceable_head_dim, self._slice_size = heads, None
        self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
        self._use_memory_efficient_attention_xformers = False
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)   # bias only in the stage-1 prior (attention_bias=True)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def set_attention_slice(self, slice_size):
        # the flash kernel never materialises the score tensor; slicing is accepted and ignored
        if slice_size is not None and slice_size > self.sliceable_head_dim:
            raise ValueError(f"slice_size {slice_size} has to be smaller or equal to {self.sliceable_head_dim}.")
        self._slice_size = slice_size

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
        """attention.py:113-168: to_q / to_k / to_v -> softmax(scale QK^T) V -> to_out, on (batch, tokens, channels).
        Standalone HIP dispatch (three GEMMs around the flash kernel); inside Transformer3DModel the same kernels run
        from the block's fused plan."""
        if attention_mask is not None:
            raise NotImplementedError("CrossAttention.forward: attention_mask is never passed by the stage-2 pipeline; the "
                                      "stage-1 prior's causal / padding masks run through rcdm_flash_attn_masked in its plan")
        return engine.run_tokens("attention", self.state_dict(), hidden_states, ctx=encoder_hidden_states,
                                 heads=self.heads).to(hidden_states.dtype)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim: int, num_attention_heads: int, attention_head_dim: int, dropout=0.0,
                 cross_attention_dim: Optional[int] = None, activation_fn: str = "geglu",
                 num_embeds_ada_norm: Optional[int] = None, attention_bias: bool = False,
                 only_cross_attention: bool = False, upcast_attention: bool = False,
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None):
        super().__init__()
        assert unet_use_cross_frame_attention is not None and unet_use_temporal_attention is not None
        if unet_use_cross_frame_attention or unet_use_temporal_attention or num_embeds_ada_norm is not None \
                or only_cross_attention:
            raise NotImplementedError("BasicTransformerBlock: configuration not used by configs/testing.yaml")
        mk = dict(heads=num_attention_heads, dim_head=attention_head_dim, dropout=dropout, bias=attention_bias,
                  upcast_attention=upcast_attention)
        self.attn1 = CrossAttention(query_dim=dim, **mk)
        self.norm1 = nn.LayerNorm(dim)
        # no cross-attention in the stage-1 prior's blocks (attention.py:417-433: attn2 / norm2 are None)
        self.attn2 = CrossAttention(query_dim=dim, cross_attention_dim=cross_attention_dim, **mk) \
            if cross_attention_dim is not None else None
        self.norm2 = nn.LayerNorm(dim) if cross_attention_dim is not None else None
        self.ff = FeedForward(dim, dropout=dropout, activation_fn=activation_fn)
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, attention_mask=None, video_length=None):
        """attention.py:479-526: h += attn1(norm1(h)); h += attn2(norm2(h), ctx); h += ff(norm3(h)) on
        (batch*frames, tokens, channels).  Standalone HIP dispatch of the block's launch list."""
        if attention_mask is not None:
            raise NotImplementedError("BasicTransformerBlock.forward: attention_mask is never passed by the stage-2 pipeline")
        return engine.run_tokens("block", self.state_dict(), hidden_states, ctx=encoder_hidden_states,
                                 heads=self.attn1.heads).to(hidden_states.dtype)


class Transformer3DModel(nn.Module):
    def __init__(self, num_attention_heads: int = 16, attention_head_dim: int = 88, in_channels: Optional[int] = None,
                 num_layers: int = 1, dropout: float = 0.0, norm_num_groups: int = 32,
                 cross_attention_dim: Optional[int] = None, attention_bias: bool = False, activation_fn: str = "geglu",
                 num_embeds_ada_norm: Optional[int] = None, use_linear_projection: bool = False,
                 only_cross_attention: bool = False, upcast_attention: bool = False,
                 unet_use_cross_frame_attention=None, unet_use_temporal_attention=None):
        super().__init__()
        if use_linear_projection or num_layers != 1:
            raise NotImplementedError("Transformer3DModel: SD-1.5 layout only (1 layer, conv projections)")
        inner = num_attention_heads * attention_head_dim
        if inner != in_channels:
            raise NotImplementedError("Transformer3DModel: inner_dim must equal in_channels")
        self.num_attention_heads, self.attention_head_dim, self.in_channels = num_attention_heads, attention_head_dim, in_channels
        self.use_linear_projection, self.norm_num_groups = use_linear_projection, norm_num_groups
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(
            inner, num_attention_heads, attention_head_dim, dropout=dropout, cross_attention_dim=cross_attention_dim,
            activation_fn=activation_fn, num_embeds_ada_norm=num_embeds_ada_norm, attention_bias=attention_bias,
            only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
            unet_use_cross_frame_attention=unet_use_cross_frame_attention,
            unet_use_temporal_attention=unet_use_temporal_attention)])
        self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1)

    def forward(self, hidden_states, encoder_hidden_states=None, timestep=None, return_dict: bool = True):
        assert hidden_states.dim() == 5, f"Expected hidden_states to have ndim=5, but got ndim={hidden_states.dim()}."
        out = engine.run_block("transformer", self.state_dict(), hidden_states, ctx=encoder_hidden_states,
                               heads=self.num_attention_heads, groups=self.norm_num_groups)
        return Transformer3DModelOutput(sample=out) if return_dict else (out,)
