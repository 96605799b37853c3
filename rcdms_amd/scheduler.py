"""DDIM scheduler with the object protocol RCDMsPipeline expects from diffusers' DDIMScheduler
(reference: built at stage2_batchtest_rcdms_model.py:247 from configs/testing.yaml:18-21, mutated at
src/pipelines/RCDMs_pipeline.py:84-109, used at :455-456,483,497).

The arithmetic is diffusers==0.24.0's (not vendored by the reference, not installed here): restated from the
DDIM paper eq. (12) with eta = 0 / epsilon prediction — "parity unpinned" by any reference test, pinned by
closed-form known-answer tests (tests/test_scheduler.py).  `step()` is the host-visible (torch) form for
callers that drive the loop themselves; the product's hot loop uses the same coefficients through the fused
rcdm_cfg_ddim_step kernel (rcdms_amd/sampler.py)."""
from dataclasses import dataclass

import torch


class _FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", timestep_spacing="leading"):
        self._internal_dict = _FrozenDict(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
            set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset, prediction_type=prediction_type,
            timestep_spacing=timestep_spacing)
        if prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only epsilon prediction with 'leading' spacing (the reference's configuration)")
        if trained_betas is not None:
            betas = torch.as_tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        self.betas = betas
        self.alphas = 1.0 - betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)

    @property
    def config(self):
        return self._internal_dict

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {c.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        ratio = c.num_train_timesteps // num_inference_steps
        ts = (torch.arange(num_inference_steps) * ratio).flip(0).to(torch.int64) + c.steps_offset
        self.timesteps = ts.to(device) if device is not None else ts

    def coefficients(self):
        """[n][4] fp32 = sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev) per inference step (fp64 math)."""
        c = self.config
        ratio = c.num_train_timesteps // self.num_inference_steps
        ac = self.alphas_cumprod.double()
        rows = []
        for t in self.timesteps.tolist():
            prev = t - ratio
            a_t = ac[t]
            a_p = ac[prev] if prev >= 0 else self.final_alpha_cumprod.double()
            rows.append([a_t.sqrt(), (1 - a_t).sqrt(), a_p.sqrt(), (1 - a_p).sqrt()])
        return torch.tensor(rows, dtype=torch.float32)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta != 0.0:
            raise NotImplementedError("eta > 0 (stochastic DDIM) is not used by the reference")
        c = self.config
        t = int(timestep)
        prev = t - c.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].to(sample.device)
        a_p = (self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod).to(sample.device)
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        if c.clip_sample:
            x0 = x0.clamp(-1.0, 1.0)
            model_output = (sample - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5
        prev_sample = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=x0)
