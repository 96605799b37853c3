"""Mirror of the reference's src/pipelines/RCDMs_pipeline.py interface: `RCDMsPipeline` (and the alias
`AnimationPipeline` that stage2_batchtest_rcdms_model.py:246 actually instantiates), `RCDMsPipelineOutput`,
`local_feature`, same constructor (RCDMs_pipeline.py:64-80) and `__call__` signature (:374-398).

What runs where:
  * the denoising loop (:455-503) — T x [UNet + CFG + DDIM] — runs on MI355X as replays of one captured hipGraph
    (rcdms_amd.sampler.DenoiseLoop); this is the hot path and has no torch fallback;
  * prompt / VAE / context-stack glue around it calls the user-supplied torch modules exactly as the reference
    does (they are inputs of this pipeline, not part of it).
Reference quirks kept by default: exactly 5 frames per story, batch 1 per call, and the context
rows re-joined as cat([seen, unseen]) (:450) — pass fix_context_order=True to restore (b f) row order.
Generalised: any (height, width) divisible by 64, CFG on or off."""
import inspect
from dataclasses import dataclass
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from rcdms_amd.sampler import DenoiseLoop
from ..models.unet import UNet3DConditionModel, _Config


class local_feature(nn.Module):
    """Text-queries x visual-keys MHA "stack" (reference RCDMs_pipeline.py:35-52): user-side torch module."""

    def __init__(self, text_dim, vis_dim, hidden_dim, num_heads):
        super().__init__()
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.text_fc = nn.Linear(text_dim, hidden_dim)
        self.vis_fc = nn.Linear(vis_dim, hidden_dim)
        self.multihead_attn = nn.MultiheadAttention(embed_dim=hidden_dim, num_heads=num_heads)

    def forward(self, vis_f, text_f):
        q = self.text_fc(text_f).transpose(0, 1)
        kv = self.vis_fc(vis_f).transpose(0, 1)
        return self.multihead_attn(q, kv, kv)[0].transpose(0, 1)


@dataclass
class RCDMsPipelineOutput:
    videos: Union[torch.Tensor, np.ndarray]


def _force_config(obj, key, value):
    cfg = dict(obj.config)
    cfg[key] = value
    obj._internal_dict = _Config(cfg)


class RCDMsPipeline:
    _optional_components = []
    FRAMES = 5  # hard-coded in the reference (:261,:430,:476); the PE table of the motion modules has 5 rows

    def __init__(self, vae, text_encoder, tokenizer, unet: UNet3DConditionModel, local_module, global_module, scheduler,
                 vae_decoder=None):
        # the two scheduler-config mutations of the reference ctor (:84-109)
        if hasattr(scheduler, "config"):
            if hasattr(scheduler.config, "steps_offset") and scheduler.config.steps_offset != 1:
                _force_config(scheduler, "steps_offset", 1)
            if hasattr(scheduler.config, "clip_sample") and scheduler.config.clip_sample is True:
                _force_config(scheduler, "clip_sample", False)
        if hasattr(unet, "config") and getattr(unet.config, "sample_size", None) is not None \
                and unet.config.sample_size < 64 and hasattr(unet.config, "_diffusers_version"):
            _force_config(unet, "sample_size", 64)
        self.vae, self.text_encoder, self.tokenizer, self.unet = vae, text_encoder, tokenizer, unet
        # optional rcdms_amd.vae.AutoencoderKLDecoder: decodes the 5 frames in one HIP launch plan instead of five
        # `self.vae.decode` calls (:281); `vae` is still used for `encode`.  Passing rcdms_amd.vae.AutoencoderKL as
        # `vae` itself puts both halves on the HIP path.
        from rcdms_amd.vae import AutoencoderKLDecoder
        if vae_decoder is None and isinstance(vae, AutoencoderKLDecoder):
            vae_decoder = vae
        self.vae_decoder = vae_decoder
        self.local_module, self.global_module, self.scheduler = local_module, global_module, scheduler
        boc = getattr(getattr(vae, "config", None), "block_out_channels", None)
        self.vae_scale_factor = 2 ** (len(boc) - 1) if boc is not None else 8
        self._loop = None
        self._loop_key = None
        self._progress_bar_config = {}

    # ---- DiffusionPipeline conveniences used by the reference's callers ------------------------------------------
    @property
    def components(self):
        return dict(vae=self.vae, text_encoder=self.text_encoder, tokenizer=self.tokenizer, unet=self.unet,
                    local_module=self.local_module, global_module=self.global_module, scheduler=self.scheduler)

    def to(self, device):
        for m in (self.vae, self.text_encoder, self.unet, self.local_module, self.global_module):
            if isinstance(m, nn.Module):
                m.to(device)
        return self

    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def enable_sequential_cpu_offload(self, gpu_id=0):
        raise NotImplementedError("the HIP path keeps the UNet resident in HBM (288 GB); CPU offload is not supported")

    def set_progress_bar_config(self, **kwargs):
        self._progress_bar_config = kwargs

    def progress_bar(self, total):
        from tqdm import tqdm
        return tqdm(total=total, **self._progress_bar_config)

    # ---- glue identical in behaviour to the reference ----------------------------------------------------------------
    def _encode_prompt(self, prompt, device, num_videos_per_prompt, do_classifier_free_guidance, negative_prompt):
        """(:175-256) tokenise to text_encoder.max_position_embeddings, take last_hidden_state; uncond rows first."""
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        max_len = self.text_encoder.max_position_embeddings

        def embed(texts):
            ids = self.tokenizer(texts, padding="max_length", max_length=max_len, truncation=False,
                                 return_tensors="pt").input_ids
            return self.text_encoder(ids.to(device)).last_hidden_state

        text = embed(prompt)
        if not do_classifier_free_guidance:
            return text
        if negative_prompt is None:
            uncond_tokens = [""] * batch_size
        elif type(prompt) is not type(negative_prompt):
            raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                            f" {type(prompt)}.")
        elif isinstance(negative_prompt, str):
            uncond_tokens = [negative_prompt]
        elif batch_size != len(negative_prompt):
            raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                             f" {prompt} has batch size {batch_size}. Please make sure that passed `negative_prompt`"
                             " matches the batch size of `prompt`.")
        else:
            uncond_tokens = negative_prompt
        uncond = embed(uncond_tokens)
        seq = uncond.shape[1]
        uncond = uncond.repeat(1, num_videos_per_prompt, 1).view(batch_size * num_videos_per_prompt, seq, -1)
        return torch.cat([uncond, text])

    def encode_mask(self, mask_label, num_videos_per_prompt, do_classifier_free_guidance):
        """(:259-272) (5, h, w) -> (10, h, w): the same mask for the unconditional half."""
        if not do_classifier_free_guidance:
            return mask_label
        seq_len = mask_label.shape[1]
        uncond = mask_label.repeat(1, num_videos_per_prompt, 1).view(self.FRAMES * num_videos_per_prompt, seq_len, -1)
        return torch.cat([uncond, mask_label])

    def mask2list_label(self, mask_label, encoder_hidden_states, do_classifier_free_guidance):
        """(:350-371) split the text rows into seen (mask all 1) / unseen (mask all 0) frames."""
        seen = self._seen_rows(mask_label)
        return encoder_hidden_states[seen], encoder_hidden_states[~seen]

    @staticmethod
    def _seen_rows(mask_label):
        flat = mask_label.reshape(mask_label.shape[0], -1)
        all0, all1 = (flat == 0).all(dim=1), (flat == 1).all(dim=1)
        if not bool((all0 | all1).all()):
            raise ValueError('please check mask label')
        return all1.cpu()

    def build_context(self, text_embeddings, masked_label, image_embeds_1, proj_embeds_0, fix_context_order=False):
        """(:444-450) rich context = cat([local(seen rows), global(unseen rows)]); the reference's row order is
        seen-first, which differs from the latents' (b f) order whenever CFG is on."""
        dev, dt = text_embeddings.device, text_embeddings.dtype
        ehs_1, ehs_0 = self.mask2list_label(masked_label, text_embeddings, True)
        seen = self._seen_rows(masked_label)
        feature_1 = self.local_module(image_embeds_1.to(dtype=dt, device=dev), ehs_1)
        feature_0 = self.global_module(proj_embeds_0.to(dtype=dt, device=dev), ehs_0)
        ctx = torch.cat([feature_1, feature_0], dim=0)
        if fix_context_order:
            order = torch.cat([torch.nonzero(seen).flatten(), torch.nonzero(~seen).flatten()])
            fixed = torch.empty_like(ctx)
            fixed[order.to(ctx.device)] = ctx
            ctx = fixed
        return ctx

    def decode_latents(self, latents):
        """(:274-287) one frame at a time through the user's VAE."""
        f = latents.shape[2]
        latents = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(-1, latents.shape[1], *latents.shape[3:])
        if self.vae_decoder is not None:
            video = self.vae_decoder.decode(latents).sample
        else:
            frames = [self.vae.decode(latents[i:i + 1]).sample for i in range(latents.shape[0])]
            video = torch.cat(frames)
        video = video.reshape(-1, f, *video.shape[1:]).permute(0, 2, 1, 3, 4)
        return (video / 2 + 0.5).clamp(0, 1).cpu().float().numpy()

    def prepare_extra_step_kwargs(self, generator, eta):
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    def check_inputs(self, prompt, height, width, callback_steps):
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type"
                             f" {type(callback_steps)}.")

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective"
                             f" batch size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn(shape, generator=g, device=device, dtype=dtype) for g in generator], dim=0)
            else:
                latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        elif latents.shape != shape:
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
        return latents.to(device) * self.scheduler.init_noise_sigma

    # ---- the hot loop ----------------------------------------------------------------------------------------------
    def denoise(self, latents, mask, masked_latents, context, num_inference_steps, guidance_scale, callback=None,
                callback_steps=1):
        """latents (S,4,f,h,w); mask (R*S,1,f,h,w); masked_latents (R*S,4,f,h,w); context (R*S*f, L, D)."""
        S, _, f, h, w = latents.shape
        key = (S, f, h, w, context.shape[1], float(guidance_scale), int(num_inference_steps), id(self.scheduler),
               id(self.unet), self.unet._weights_gen)
        if self._loop is None or self._loop_key != key:
            self._loop = DenoiseLoop(self.unet, S, f, h, w, context.shape[1], guidance_scale, self.scheduler,
                                     num_inference_steps)
            self._loop_key = (S, f, h, w, context.shape[1], float(guidance_scale), int(num_inference_steps),
                              id(self.scheduler), id(self.unet), self.unet._weights_gen)
        self._loop.load(latents, mask, masked_latents, context)
        return self._loop.run(callback=callback, callback_steps=callback_steps)

    @torch.no_grad()
    def __call__(self, prompt, source_img, image_embeds_1, proj_embeds_0, mask_label, video_length: Optional[int],
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 50,
                 guidance_scale: float = 7.5, negative_prompt=None, num_videos_per_prompt: Optional[int] = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 output_type: Optional[str] = "tensor", return_dict: bool = True,
                 callback: Optional[Callable[[int, int, torch.Tensor], None]] = None, callback_steps: Optional[int] = 1,
                 fix_context_order: bool = False, **kwargs):
        height = height or self.unet.config.sample_size * self.vae_scale_factor
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(prompt, height, width, callback_steps)
        if eta != 0.0:
            raise NotImplementedError("eta > 0 is not used by the reference and has no fused HIP step")
        if video_length != self.FRAMES:
            raise ValueError(f"a story has exactly {self.FRAMES} frames (reference RCDMs_pipeline.py:261,430,476)")
        batch_size = 1
        device = self._execution_device
        cfg_on = guidance_scale > 1.0
        reps = 2 if cfg_on else 1

        prompt = prompt if isinstance(prompt, list) else [prompt] * batch_size
        if negative_prompt is not None and not isinstance(negative_prompt, list):
            negative_prompt = [negative_prompt] * batch_size
        text_embeddings = self._encode_prompt(prompt, device, num_videos_per_prompt, cfg_on, negative_prompt)

        # source frames -> masked latents (:427-432)
        src = source_img.unsqueeze(0)
        src = src.reshape(-1, *src.shape[2:])
        masked_latents = self.vae.encode(src.to(dtype=text_embeddings.dtype, device=device)).latent_dist.sample(
            generator=generator)
        masked_latents = masked_latents.reshape(-1, self.FRAMES, *masked_latents.shape[1:]).permute(0, 2, 1, 3, 4) * 0.18215
        masked_latents = torch.cat([masked_latents] * reps * num_videos_per_prompt) if cfg_on else masked_latents

        masked_label = mask_label.squeeze().to(dtype=text_embeddings.dtype, device=device)
        masked_label = self.encode_mask(masked_label, num_videos_per_prompt, cfg_on)
        if cfg_on:
            image_embeds_1 = torch.cat([image_embeds_1] * 2 * num_videos_per_prompt)
            proj_embeds_0 = torch.cat([proj_embeds_0] * 2 * num_videos_per_prompt)
        context = self.build_context(text_embeddings, masked_label, image_embeds_1, proj_embeds_0, fix_context_order)

        latents = self.prepare_latents(batch_size * num_videos_per_prompt, 4, video_length, height, width,
                                       text_embeddings.dtype, device, generator, latents)
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        mask5 = masked_label.view(reps, 1, self.FRAMES, h, w)   # the reference hard-codes (2,1,5,64,64) at :476

        with self.progress_bar(total=num_inference_steps) as bar:
            def on_step(i, t, lat):
                bar.update()
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, lat)
            final = self.denoise(latents, mask5, masked_latents, context, num_inference_steps, guidance_scale,
                                 callback=on_step if (callback is not None) else None, callback_steps=1)
            if callback is None:
                bar.update(num_inference_steps)

        video = self.decode_latents(final.to(text_embeddings.dtype))
        if output_type == "tensor":
            video = torch.from_numpy(video)
        if not return_dict:
            return video
        return RCDMsPipelineOutput(videos=video)


# stage2_batchtest_rcdms_model.py:30 imports RCDMsPipeline but :246 instantiates AnimationPipeline
AnimationPipeline = RCDMsPipeline
