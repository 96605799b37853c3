"""Mirror of the reference's src/models/unet.py interface: `UNet3DConditionModel` (unet.py:37-509) with the
same constructor kwargs (:41-90), `from_pretrained_2d` (:465-509), `from_config`, `forward` signature and
return convention (:322-330, raw tensor when return_dict=True :462, 1-tuple otherwise :459-460), `.config`,
`set_attention_slice`, and the reference's 1286-key state-dict layout.

forward() executes on MI355X through rcdms_amd.engine.UNetProgram: a static plan of hand-written HIP kernel
launches (captured into a hipGraph after the first call).  There is no torch fallback: a CPU tensor raises."""
import inspect
import json
import os
from dataclasses import dataclass
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from rcdms_amd import engine, hip
from .resnet import InflatedConv3d, InflatedGroupNorm
from .unet_blocks import (CrossAttnDownBlock3D, CrossAttnUpBlock3D, DownBlock3D, UNetMidBlock3DCrossAttn, UpBlock3D,
                          get_down_block, get_up_block)

WEIGHTS_NAME = "diffusion_pytorch_model.bin"  # diffusers.utils.WEIGHTS_NAME


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor


class _Config(dict):
    """Attribute + mapping access, like diffusers' FrozenDict (`unet.config.sample_size`, `dict(unet.config)`)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class Timesteps(nn.Module):
    """diffusers Timesteps(num_channels, flip_sin_to_cos, downscale_freq_shift): parameter-free holder; the
    sinusoid is computed by rcdm_timestep_embed."""

    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding(in, time_embed_dim): linear_1 -> SiLU -> linear_2 (parameter holder)."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class UNet3DConditionModel(nn.Module):
    _supports_gradient_checkpointing = True
    config_name = "config.json"

    def __init__(
            self,
            sample_size: Optional[int] = None,
            in_channels: int = 4,
            out_channels: int = 4,
            center_input_sample: bool = False,
            flip_sin_to_cos: bool = True,
            freq_shift: int = 0,
            down_block_types: Tuple[str] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D",
                                            "DownBlock3D"),
            mid_block_type: str = "UNetMidBlock3DCrossAttn",
            up_block_types: Tuple[str] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D",
                                          "CrossAttnUpBlock3D"),
            only_cross_attention: Union[bool, Tuple[bool]] = False,
            block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
            layers_per_block: int = 2,
            downsample_padding: int = 1,
            mid_block_scale_factor: float = 1,
            act_fn: str = "silu",
            norm_num_groups: int = 32,
            norm_eps: float = 1e-5,
            cross_attention_dim: int = 1280,
            attention_head_dim: Union[int, Tuple[int]] = 8,
            dual_cross_attention: bool = False,
            use_linear_projection: bool = False,
            class_embed_type: Optional[str] = None,
            num_class_embeds: Optional[int] = None,
            upcast_attention: bool = False,
            resnet_time_scale_shift: str = "default",
            use_inflated_groupnorm=False,
            use_motion_module=False,
            motion_module_resolutions=(1, 2, 4, 8),
            motion_module_mid_block=False,
            motion_module_decoder_only=False,
            motion_module_type=None,
            motion_module_kwargs={},
            unet_use_cross_frame_attention=None,
            unet_use_temporal_attention=None,
    ):
        super().__init__()
        # register_to_config equivalent: every ctor argument, by name
        frame = inspect.currentframe()
        names = [n for n in inspect.signature(UNet3DConditionModel.__init__).parameters if n != "self"]
        self.config = _Config({n: frame.f_locals[n] for n in names})
        self._internal_dict = self.config
        self._programs = {}
        self._weights_gen = 0   # bumped whenever parameters may have changed: cached launch plans / loops key on it

        if class_embed_type is not None or num_class_embeds is not None:
            raise NotImplementedError("class embeddings are not used by the stage-2 UNet")
        if mid_block_type != "UNetMidBlock3DCrossAttn":
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")
        if isinstance(attention_head_dim, (tuple, list)):
            if len(set(attention_head_dim)) != 1:
                raise NotImplementedError("per-block attention_head_dim is not supported on the HIP path")
            attention_head_dim = attention_head_dim[0]
        if not isinstance(only_cross_attention, bool):
            only_cross_attention = any(only_cross_attention)
        self.sample_size = sample_size
        boc = tuple(block_out_channels)
        ted = boc[0] * 4
        n = len(boc)

        self.conv_in = InflatedConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.class_embedding = None

        common = dict(temb_channels=ted, resnet_eps=norm_eps, resnet_act_fn=act_fn, resnet_groups=norm_num_groups,
                      cross_attention_dim=cross_attention_dim, attn_num_head_channels=attention_head_dim,
                      dual_cross_attention=dual_cross_attention, use_linear_projection=use_linear_projection,
                      only_cross_attention=only_cross_attention, upcast_attention=upcast_attention,
                      resnet_time_scale_shift=resnet_time_scale_shift,
                      unet_use_cross_frame_attention=unet_use_cross_frame_attention,
                      unet_use_temporal_attention=unet_use_temporal_attention,
                      use_inflated_groupnorm=use_inflated_groupnorm, motion_module_type=motion_module_type,
                      motion_module_kwargs=motion_module_kwargs)

        self.down_blocks = nn.ModuleList()
        ch = boc[0]
        for i, kind in enumerate(down_block_types):
            mm = use_motion_module and (2 ** i in motion_module_resolutions) and not motion_module_decoder_only
            self.down_blocks.append(get_down_block(kind, num_layers=layers_per_block, in_channels=ch,
                                                   out_channels=boc[i], add_downsample=i != n - 1,
                                                   downsample_padding=downsample_padding, use_motion_module=mm,
                                                   **common))
            ch = boc[i]

        mid_kw = {k: v for k, v in common.items() if k not in ("resnet_groups", "only_cross_attention")}
        self.mid_block = UNetMidBlock3DCrossAttn(in_channels=boc[-1], output_scale_factor=mid_block_scale_factor,
                                                 resnet_groups=norm_num_groups,
                                                 use_motion_module=use_motion_module and motion_module_mid_block,
                                                 **mid_kw)

        self.num_upsamplers = 0
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        prev = rev[0]
        for i, kind in enumerate(up_block_types):
            last = i == n - 1
            self.num_upsamplers += 0 if last else 1
            mm = use_motion_module and (2 ** (3 - i) in motion_module_resolutions)
            self.up_blocks.append(get_up_block(kind, num_layers=layers_per_block + 1, in_channels=rev[min(i + 1, n - 1)],
                                               out_channels=rev[i], prev_output_channel=prev, add_upsample=not last,
                                               use_motion_module=mm, **common))
            prev = rev[i]

        norm = InflatedGroupNorm if use_inflated_groupnorm else nn.GroupNorm
        self.conv_norm_out = norm(num_channels=boc[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = InflatedConv3d(boc[0], out_channels, kernel_size=3, padding=1)

    # ---- diffusers-style conveniences the reference's callers rely on ----------------------------------
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_config(cls, config, **kwargs):
        """ConfigMixin.from_config as used at unet.py:492: known ctor keys from `config`, overridden by kwargs."""
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        merged = {k: v for k, v in dict(config).items() if k in accepted}
        merged.update({k: v for k, v in kwargs.items() if k in accepted})
        return cls(**merged)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, unet_additional_kwargs=None):
        """Inflate SD-1.5 2-D weights: read <path>/config.json + diffusion_pytorch_model.bin, force the 9-channel
        input and the 3-D block names, load everything except conv_in (reference unet.py:465-509)."""
        if subfolder is not None:
            pretrained_model_path = os.path.join(pretrained_model_path, subfolder)
        print(f"loaded temporal unet's pretrained weights from {pretrained_model_path} ...")
        config_file = os.path.join(pretrained_model_path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")
        with open(config_file, "r") as f:
            config = json.load(f)
        config["_class_name"] = cls.__name__
        config["in_channels"] = 9
        config["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
        config["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
        model = cls.from_config(config, **(unet_additional_kwargs or {}))
        model_file = os.path.join(pretrained_model_path, WEIGHTS_NAME)
        if not os.path.isfile(model_file):
            raise RuntimeError(f"{model_file} does not exist")
        state_dict = torch.load(model_file, map_location="cpu")
        kept = {k: v for k, v in state_dict.items() if not k.startswith("conv_in")}
        missing, unexpected = model.load_state_dict(kept, strict=False)
        print(f"### missing keys: {len(missing)}; \n### unexpected keys: {len(unexpected)};")
        n_temporal = sum(p.numel() for name, p in model.named_parameters() if "temporal" in name)
        print(f"### Temporal Module Parameters: {n_temporal / 1e6} M")
        return model

    def set_attention_slice(self, slice_size):
        """Accepted for API compatibility (unet.py:253-316); the flash kernels never materialise the scores."""
        heads = [m for m in self.modules() if hasattr(m, "sliceable_head_dim")]
        if isinstance(slice_size, list) and len(slice_size) != len(heads):
            raise ValueError(f"You have provided {len(slice_size)}, but {self.config} has {len(heads)} different"
                             f" attention layers. Make sure to match `len(slice_size)` to be {len(heads)}.")

    def _set_gradient_checkpointing(self, module, value=False):
        if isinstance(module, (CrossAttnDownBlock3D, DownBlock3D, CrossAttnUpBlock3D, UpBlock3D)):
            module.gradient_checkpointing = value

    # ---- weights changed -> drop cached launch plans -------------------------------------------------------
    MAX_LIVE_PLANS = 4   # plans (geometry x shared-prefix x rank-1-context variant) kept packed at once (2.55 GB of f16 weights + ~3.5 GB of buffers each)

    def _invalidate(self):
        self._programs = {}
        self._weights_gen = getattr(self, "_weights_gen", 0) + 1

    def load_state_dict(self, *args, **kwargs):
        self._invalidate()
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._invalidate()
        return super()._apply(fn, *args, **kwargs)

    def engine_config(self):
        c = self.config
        mk = dict(c.motion_module_kwargs or {})
        ahd = c.attention_head_dim[0] if isinstance(c.attention_head_dim, (tuple, list)) else c.attention_head_dim
        if c.use_inflated_groupnorm:
            raise NotImplementedError("use_inflated_groupnorm=True is not built on the HIP path (reference default False)")
        return dict(
            block_out_channels=tuple(c.block_out_channels), layers_per_block=c.layers_per_block,
            cross_attention_dim=c.cross_attention_dim, attention_head_dim=ahd, norm_num_groups=c.norm_num_groups,
            norm_eps=c.norm_eps, down_block_types=tuple(c.down_block_types), up_block_types=tuple(c.up_block_types),
            use_motion_module=bool(c.use_motion_module) and not c.motion_module_decoder_only,
            motion_module_resolutions=tuple(c.motion_module_resolutions),
            motion_module_mid_block=bool(c.motion_module_mid_block),
            motion_num_attention_heads=mk.get("num_attention_heads", 8),
            motion_attention_blocks=len(mk.get("attention_block_types", ("Temporal_Self", "Temporal_Self"))),
            mid_block_scale_factor=c.mid_block_scale_factor)

    def program(self, b, frames, H, W, L, shared_prefix=False, rank1_runs=None):
        """The cached static launch plan for one input geometry (shared_prefix, rank1_runs: see rcdms_amd.engine.UNetProgram)."""
        dev = self.device
        if dev.type != "cuda":
            raise hip.RcdmError("UNet3DConditionModel runs on MI355X only: move the model to a CUDA/HIP device "
                                "(rcdms_amd has no CPU fallback)")
        if self.config.motion_module_decoder_only and self.config.use_motion_module:
            raise NotImplementedError("motion_module_decoder_only is not supported on the HIP path")
        if rank1_runs is not None and (sum(i1 - i0 for i0, i1 in rank1_runs) >= b * frames or not engine.SW.RANK1_CTX):
            rank1_runs = None       # every image full rank (or the fast path switched off): the general plan
        key = (b, frames, H, W, L, str(dev), bool(shared_prefix), tuple(rank1_runs) if rank1_runs is not None else None)
        prog = self._programs.get(key)
        if prog is None:
            while len(self._programs) >= self.MAX_LIVE_PLANS:      # least recently used geometry goes first
                self._programs.pop(next(iter(self._programs)))
            with torch.no_grad():
                prog = engine.UNetProgram(self.engine_config(), self.state_dict(), b, frames, H, W, L, dev,
                                          shared_prefix=shared_prefix, rank1_runs=rank1_runs)
        else:
            self._programs.pop(key)
        self._programs[key] = prog                                  # (re)insert as most recently used
        return prog

    def numerics_report(self, sample, timestep, encoder_hidden_states, verbose=True):
        """Range / headroom report of the f16 HIP path for THESE weights on THIS input — run it once after
        load_state_dict() of a real checkpoint (rcdms_amd/numerics.py says what is reported): per plan
        buffer max |activation| against the f16 limit 65504, per self-attention site the weight-norm score bound and
        whether the wide-range kernel was selected, per deferred-LayerNorm site max |mean rstd S|.  Non-finite values
        anywhere raise RcdmError.  Arguments as forward() (unet.py:322-330); returns the report dict."""
        from rcdms_amd.numerics import numerics_report
        b, _, f, H, W = sample.shape
        prog = self.program(b, f, H, W, encoder_hidden_states.shape[1], rank1_runs=self._context_runs(encoder_hidden_states))
        rep = numerics_report(prog, sample, timestep, encoder_hidden_states)
        if verbose:
            print(rep["text"])
        return rep

    def _context_runs(self, ctx):
        """Which images of this context have L identical rows (the reference's unseen frames,
        RCDMs_pipeline.py:447-450) — decides the plan variant as rcdms_amd.engine.full_rank_runs does.  One device
        comparison + a host read per NEW context tensor: the reference's loop passes the same tensor object at every
        step (RCDMs_pipeline.py:488), so the answer is cached on (object, version)."""
        if not engine.SW.RANK1_CTX:
            return None
        try:
            ver = ctx._version
        except RuntimeError:
            ver = None
        c = getattr(self, "_ctx_runs_cache", None)
        if ver is not None and c is not None and c[0] is ctx and c[1] == ver:
            return c[2]
        runs = engine.full_rank_runs(ctx.detach())
        self._ctx_runs_cache = (ctx, ver, runs) if ver is not None else None
        return runs

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor, class_labels: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, return_dict: bool = True):
        if attention_mask is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / class_labels are never passed by the stage-2 pipeline")
        if sample.dim() != 5:
            raise ValueError(f"sample must be (b, c, f, h, w), got {tuple(sample.shape)}")
        if self.config.center_input_sample:
            sample = 2 * sample - 1.0
        b, c_in, f, H, W = sample.shape
        if c_in != self.config.in_channels:
            raise ValueError(f"sample has {c_in} channels, conv_in expects {self.config.in_channels} "
                             f"(latents + mask + masked latents, RCDMs_pipeline.py:482)")
        mk = dict(self.config.motion_module_kwargs or {})
        if self.config.use_motion_module and mk.get("temporal_position_encoding", False):
            max_len = mk.get("temporal_position_encoding_max_len", 24)
            if f > max_len:
                raise ValueError(f"{f} frames exceed the temporal position-encoding table (max_len {max_len})")
        if encoder_hidden_states.dim() != 3 or encoder_hidden_states.shape[0] != b * f:
            raise ValueError(f"encoder_hidden_states must be (b*f, L, D) = ({b * f}, L, {self.config.cross_attention_dim}), "
                             f"got {tuple(encoder_hidden_states.shape)}")
        t = timestep
        if torch.is_tensor(t) and t.numel() > 1:
            t = t.reshape(-1)
            if t.numel() != b:
                raise ValueError("timestep must be a scalar or have one entry per batch element")
        prog = self.program(b, f, H, W, encoder_hidden_states.shape[1], rank1_runs=self._context_runs(encoder_hidden_states))
        out = prog.forward(sample, t, encoder_hidden_states)
        out = out.to(sample.dtype) if sample.dtype != torch.float32 else out
        if not return_dict:
            return (out,)
        return out
