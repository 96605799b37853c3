"""Mirror of the reference's src/pipelines/prior_pipeline.py interface: `Seq_Inpaint_Prior_Pipeline` (:84-374), the
stage-1 pipeline that predicts the CLIP image embeddings of the frames to generate.

`tokenizer`, `text_encoder` and `image_encoder` are the caller's CLIP modules (transformers), used exactly where the
reference uses them (`_encode_prompt` :136-232, `get_zero_embed` :124-133).  The sampling loop (:293-344) — prior
forward, classifier-free guidance, UnCLIPScheduler.step — runs as replays of one hipGraph (rcdms_amd.sampler.PriorLoop).
The scheduler must expose the rcdms_amd.scheduler.UnCLIPScheduler protocol (alphas_cumprod, coefficients())."""
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Union

import torch

from rcdms_amd.sampler import PriorLoop
from src.models.myprior_transformer import MyPriorTransformer


@dataclass
class KandinskyPriorPipelineOutput:
    image_embeds: torch.Tensor
    negative_image_embeds: torch.Tensor


class Seq_Inpaint_Prior_Pipeline:
    def __init__(self, prior: MyPriorTransformer, image_encoder, text_encoder, tokenizer, scheduler):
        self.prior, self.image_encoder, self.text_encoder = prior, image_encoder, text_encoder
        self.tokenizer, self.scheduler = tokenizer, scheduler
        self._guidance_scale = 4.0
        self._num_timesteps = 0
        self._loops = {}

    def to(self, device):
        for m in (self.prior, self.image_encoder, self.text_encoder):
            if hasattr(m, "to"):
                m.to(device)
        return self

    @property
    def device(self):
        return self.prior.device

    _execution_device = device

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def num_timesteps(self):
        return self._num_timesteps

    def prepare_latents(self, shape, dtype, device, generator, latents, scheduler):
        """prior_pipeline.py:110-121."""
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        else:
            if tuple(latents.shape) != tuple(shape):
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * scheduler.init_noise_sigma

    def get_zero_embed(self, batch_size=1, device=None):
        """prior_pipeline.py:124-133."""
        device = device or self.device
        size = self.image_encoder.config.image_size
        zero_img = torch.zeros(1, 3, size, size).to(device=device, dtype=self.image_encoder.dtype)
        return self.image_encoder(zero_img)["image_embeds"].repeat(batch_size, 1)

    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None):
        """prior_pipeline.py:136-232 (KandinskyPriorPipeline._encode_prompt): returns prompt_embeds (B, E),
        text_encoder_hidden_states (B, T, E), text_mask (B, T) with the unconditional half FIRST."""
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        max_len = self.text_encoder.max_position_embeddings if hasattr(self.text_encoder, "max_position_embeddings") \
            else self.tokenizer.model_max_length

        def encode(texts):
            tok = self.tokenizer(texts, padding="max_length", max_length=max_len, truncation=True, return_tensors="pt")
            out = self.text_encoder(tok.input_ids.to(device))
            return out.text_embeds, out.last_hidden_state, tok.attention_mask.bool().to(device)

        emb, hid, mask = encode(prompt)
        emb = emb.repeat_interleave(num_images_per_prompt, dim=0)
        hid = hid.repeat_interleave(num_images_per_prompt, dim=0)
        mask = mask.repeat_interleave(num_images_per_prompt, dim=0)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                uncond = [""] * batch_size
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt` has batch size {len(negative_prompt)}, but `prompt` has batch size "
                                 f"{batch_size}.")
            else:
                uncond = negative_prompt
            u_emb, u_hid, u_mask = encode(uncond)
            u_emb = u_emb.repeat_interleave(num_images_per_prompt, dim=0)
            u_hid = u_hid.repeat_interleave(num_images_per_prompt, dim=0)
            u_mask = u_mask.repeat_interleave(num_images_per_prompt, dim=0)
            emb, hid, mask = torch.cat([u_emb, emb]), torch.cat([u_hid, hid]), torch.cat([u_mask, mask])
        return emb, hid, mask

    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]], imgs_proj_embeds1, mask_label, video_length: Optional[int],
                 height: Optional[int] = None, width: Optional[int] = None, num_videos_per_prompt: Optional[int] = 1,
                 negative_prompt: Optional[Union[str, List[str]]] = None, num_inference_steps: int = 25,
                 generator=None, latents: Optional[torch.Tensor] = None, guidance_scale: float = 4.0,
                 output_type: Optional[str] = "pt", return_dict: bool = True,
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"]):
        """prior_pipeline.py:245-374.  Differences: the loop runs on the device as graph replays, so
        `callback_on_step_end` is not supported (NotImplementedError)."""
        if callback_on_step_end is not None:
            raise NotImplementedError("per-step callbacks: the sampling loop runs as hipGraph replays")
        if negative_prompt is not None:
            prompt = prompt + negative_prompt
            negative_prompt = 2 * negative_prompt
        device = self._execution_device
        batch_size = 1
        self._guidance_scale = guidance_scale
        emb, hid, mask = self._encode_prompt(prompt, device, num_videos_per_prompt, self.do_classifier_free_guidance,
                                             negative_prompt)
        E = self.prior.config.embedding_dim
        latents = self.prepare_latents((batch_size * video_length, E), emb.dtype, device, generator, latents, self.scheduler)
        if self.do_classifier_free_guidance:
            imgs_proj_embeds1 = torch.cat([imgs_proj_embeds1] * 2)
            mask_label = torch.cat([mask_label] * 2)
        key = (video_length, hid.shape[1], float(guidance_scale), int(num_inference_steps))
        if key not in self._loops:
            self._loops[key] = PriorLoop(self.prior, video_length, hid.shape[1], guidance_scale, self.scheduler,
                                         num_inference_steps)
        loop = self._loops[key]
        self._num_timesteps = loop.T
        loop.load(latents / self.scheduler.init_noise_sigma, emb, hid, imgs_proj_embeds1, mask_label, mask,
                  generator=generator)
        latents = loop.run().clone().to(emb.dtype)
        image_embeddings = self.prior.post_process_latents(latents)
        if negative_prompt is None:
            zero_embeds = self.get_zero_embed(latents.shape[0], device=latents.device)
        else:
            image_embeddings, zero_embeds = image_embeddings.chunk(2)
        if output_type not in ["pt", "np"]:
            raise ValueError(f"Only the output types `pt` and `np` are supported not output_type={output_type}")
        if output_type == "np":
            image_embeddings, zero_embeds = image_embeddings.cpu().numpy(), zero_embeds.cpu().numpy()
        if not return_dict:
            return (image_embeddings, zero_embeds)
        return KandinskyPriorPipelineOutput(image_embeds=image_embeddings, negative_image_embeds=zero_embeds)
