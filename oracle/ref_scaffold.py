"""ORACLE — TEST INFRASTRUCTURE ONLY (build container only; /root/reference does not exist on the GPU box).

Scaffolding that lets the reference's OWN model files (/root/reference/src/models/{resnet,attention,
motion_module,unet_blocks,unet}.py) be imported and run on CPU in this container, where `diffusers`,
`torchvision` and `xformers` are not installed.  It provides:
  * an empty `torchvision` (imported, never used: motion_module.py:8);
  * a minimal `diffusers` exposing the ~10 symbols those files import: ModelMixin, ConfigMixin,
    register_to_config, FrozenDict, BaseOutput, logging.get_logger, is_xformers_available -> False,
    WEIGHTS_NAME, and the three pieces of third-party ARITHMETIC on the path — Timesteps,
    TimestepEmbedding, FeedForward/GEGLU — restated from the published diffusers==0.24.0 definitions
    (reference requirements.txt:12).  Those three are therefore NOT pinned by the reference:
    the golden vectors pin everything the reference repo itself contains.
The reference package is loaded under the alias `refsrc` so it cannot collide with this repo's `src`.
Nothing here is imported by the product."""
import functools
import importlib
import importlib.util
import inspect
import math
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REF_ROOT = "/root/reference"


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def register_to_config(init):
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        init(self, *args, **kwargs)
        self._internal_dict = FrozenDict(cfg)
    return wrapped


class ConfigMixin:
    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        merged = {k: v for k, v in dict(config).items() if k in accepted}
        merged.update({k: v for k, v in kwargs.items() if k in accepted})
        return cls(**merged)


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(dict):
    """Dataclass-with-mapping base of diffusers; the reference only reads `.sample`."""

    def __post_init__(self):
        for f in getattr(self, "__dataclass_fields__", {}):
            self[f] = getattr(self, f)


# ---- diffusers==0.24.0 arithmetic used by the reference UNet (third-party, restated) ----------------

class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(self, timesteps):
        half = self.num_channels // 2
        exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - self.downscale_freq_shift)
        emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if self.flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        assert act_fn == "silu"
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        x, gate = self.proj(x).chunk(2, dim=-1)
        return x * F.gelu(gate)


class GELU(nn.Module):  # diffusers 0.24.0 models.activations.GELU(dim_in, dim_out, approximate="none")
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out)

    def forward(self, x):
        return F.gelu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False):
        super().__init__()
        inner = int(dim * mult)
        assert activation_fn in ("geglu", "gelu")
        act = GEGLU(dim, inner) if activation_fn == "geglu" else GELU(dim, inner)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner, dim_out or dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class AdaLayerNorm(nn.Module):  # imported by attention.py:14, never instantiated by stage 2
    def __init__(self, *a, **k):
        raise NotImplementedError


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Put the stub modules in sys.modules (idempotent)."""
    if "diffusers" in sys.modules and getattr(sys.modules["diffusers"], "_rcdm_stub", False):
        return
    _mod("torchvision")
    logging = _mod("diffusers.utils.logging", get_logger=lambda name=None: __import__("logging").getLogger(name or "ref"))
    import_utils = _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    utils = _mod("diffusers.utils", BaseOutput=BaseOutput, logging=logging, WEIGHTS_NAME="diffusion_pytorch_model.bin",
                 import_utils=import_utils, is_accelerate_available=lambda: False, deprecate=lambda *a, **k: None)
    cfgu = _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config,
                FrozenDict=FrozenDict)
    emb = _mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding)
    att = _mod("diffusers.models.attention", FeedForward=FeedForward, AdaLayerNorm=AdaLayerNorm)
    models = _mod("diffusers.models", embeddings=emb, attention=att)
    # extra names src/models/myprior_transformer.py imports (stage-1 prior); none carries arithmetic
    loaders = _mod("diffusers.loaders", UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}))
    ap = _mod("diffusers.models.attention_processor",
              ADDED_KV_ATTENTION_PROCESSORS=(), CROSS_ATTENTION_PROCESSORS=(), AttentionProcessor=object,
              AttnAddedKVProcessor=type("AttnAddedKVProcessor", (), {}), AttnProcessor=type("AttnProcessor", (), {}))
    mu = _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    models.attention_processor, models.modeling_utils = ap, mu
    d = _mod("diffusers", ModelMixin=ModelMixin, utils=utils, configuration_utils=cfgu, models=models, loaders=loaders)
    d._rcdm_stub = True


def load_reference_models():
    """Import the reference's src.models.* under the alias `refsrc` and return the unet module."""
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"{REF_ROOT} not present: golden vectors can only be minted in the build container")
    install()
    if "refsrc" not in sys.modules:
        pkg = types.ModuleType("refsrc")
        pkg.__path__ = [os.path.join(REF_ROOT, "src")]
        sys.modules["refsrc"] = pkg
        sub = types.ModuleType("refsrc.models")
        sub.__path__ = [os.path.join(REF_ROOT, "src", "models")]
        sys.modules["refsrc.models"] = sub
    return importlib.import_module("refsrc.models.unet")


# configs/testing.yaml:1-15 as a plain dict (omegaconf is not installed)
TESTING_YAML_UNET_KWARGS = dict(
    use_motion_module=True, motion_module_resolutions=[1, 2, 4, 8], unet_use_cross_frame_attention=False,
    unet_use_temporal_attention=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=["Temporal_Self", "Temporal_Self"],
                              temporal_position_encoding=True, temporal_position_encoding_max_len=5,
                              temporal_attention_dim_div=1, zero_initialize=True))

# runwayml/stable-diffusion-v1-5 unet/config.json fields that UNet3DConditionModel.__init__ accepts
SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    block_out_channels=[320, 640, 1280, 1280], layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8)


def build_reference_unet(width=None, cross_dim=None, layers_per_block=None):
    """The reference UNet exactly as from_pretrained_2d builds it (unet.py:476-492), optionally narrowed."""
    ref_unet = load_reference_models()
    cfg = dict(SD15_UNET_CONFIG)
    cfg["in_channels"] = 9
    cfg["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
    cfg["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
    if width is not None:
        cfg["block_out_channels"] = [width, 2 * width, 4 * width, 4 * width]
    if cross_dim is not None:
        cfg["cross_attention_dim"] = cross_dim
    if layers_per_block is not None:
        cfg["layers_per_block"] = layers_per_block
    model = ref_unet.UNet3DConditionModel.from_config(cfg, **TESTING_YAML_UNET_KWARGS)
    return model.eval()


def load_reference_context_stacks():
    """Import the reference driver stage2_batchtest_rcdms_model.py (its `fine_stack` / `semantic_stack` classes,
    :117-149, live there, not under src/) as module `refdriver` and return (fine_stack, semantic_stack).
    The script's unrelated top-level imports that this container lacks (cv2, omegaconf, skimage, h5py,
    torchvision.transforms, transformers' CLIP classes, diffusers model/scheduler classes) are satisfied with empty placeholders: none of them
    is touched by the two classes, which are nn.Linear + torch.nn.MultiheadAttention.  Its `from src...` imports are
    pointed at the reference package (alias refsrc) with the pipeline module replaced by a placeholder, because
    RCDMs_pipeline.py needs the real diffusers."""
    if "refdriver" in sys.modules:
        m = sys.modules["refdriver"]
        return m.fine_stack, m.semantic_stack
    load_reference_models()
    d = sys.modules["diffusers"]
    for name in ("AutoencoderKL", "DDPMScheduler", "UNet2DConditionModel", "DDIMScheduler"):
        if not hasattr(d, name):
            setattr(d, name, type(name, (), {}))
    placeholders = {
        "cv2": {}, "h5py": {}, "omegaconf": {"OmegaConf": type("OmegaConf", (), {})},
        "skimage": {}, "skimage.metrics": {"structural_similarity": None},
        "torchvision.transforms": {},
    }
    saved = {}
    for name, attrs in placeholders.items():
        if name not in sys.modules:
            _mod(name, **attrs)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    # the driver says `from src.models.unet import ...` / `from src.pipelines.RCDMs_pipeline import RCDMsPipeline`
    for k in ("src", "src.models", "src.models.unet", "src.pipelines", "src.pipelines.RCDMs_pipeline", "transformers"):
        saved[k] = sys.modules.get(k)
    try:
        # the installed transformers refuses to import beside the spec-less torchvision placeholder; the driver only
        # names six of its classes at import time
        _mod("transformers", **{n: type(n, (), {}) for n in (
            "CLIPVisionModelWithProjection", "CLIPTextModelWithProjection", "CLIPVisionModel", "CLIPImageProcessor",
            "CLIPTextModel", "CLIPTokenizer")})
        importlib.import_module("refsrc.models.unet")
        sys.modules["src"] = sys.modules["refsrc"]
        sys.modules["src.models"] = sys.modules["refsrc.models"]
        sys.modules["src.models.unet"] = sys.modules["refsrc.models.unet"]
        _mod("src.pipelines")
        _mod("src.pipelines.RCDMs_pipeline", RCDMsPipeline=type("RCDMsPipeline", (), {}))
        spec = importlib.util.spec_from_file_location("refdriver", os.path.join(REF_ROOT, "stage2_batchtest_rcdms_model.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules["refdriver"] = m
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m.fine_stack, m.semantic_stack


PRIOR_MOTION_KWARGS = dict(num_attention_heads=8, num_transformer_block=1,
                           attention_block_types=["Temporal_Self", "Temporal_Self"], temporal_position_encoding=True,
                           temporal_position_encoding_max_len=5, temporal_attention_dim_div=1)


def build_reference_prior(num_layers=20, heads=32, head_dim=64, embedding_dim=1280, num_embeddings=91,
                          additional_embeddings=6):
    """The reference's MyPriorTransformer (src/models/myprior_transformer.py) as stage1_batchtest_rcdms_model.py:99
    builds it: the Kandinsky-2.2 prior config with num_embeddings / additional_embeddings overridden
    (myprior_transformer.py:428-429) and the testing.yaml motion-module kwargs.  Smaller sizes for fixtures."""
    load_reference_models()
    mod = importlib.import_module("refsrc.models.myprior_transformer")
    cfg = dict(num_attention_heads=heads, attention_head_dim=head_dim, num_layers=num_layers,
               embedding_dim=embedding_dim, num_embeddings=num_embeddings, additional_embeddings=additional_embeddings,
               dropout=0.0, time_embed_act_fn="silu", norm_in_type=None, embedding_proj_norm_type=None,
               encoder_hid_proj_type="linear", added_emb_type="prd")
    return mod.MyPriorTransformer(**cfg, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
                                  use_motion_module=True, motion_module_type="Vanilla",
                                  motion_module_kwargs=dict(PRIOR_MOTION_KWARGS)).eval()
