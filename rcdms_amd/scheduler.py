"""DDIM (and PNDM / UnCLIP) schedulers with the object protocol RCDMsPipeline expects from diffusers' DDIMScheduler
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
        if use_clipped_model_output:   # diffusers 0.24: epsilon is re-derived from the clipped x0 only on request
            model_output = (sample - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5
        prev_sample = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output
        if not return_dict:
            return (prev_sample,)
        return DDIMSchedulerOutput(prev_sample=prev_sample, pred_original_sample=x0)


@dataclass
class PNDMSchedulerOutput:
    prev_sample: torch.Tensor


class PNDMScheduler:
    """PNDM / PLMS (the scheduler type RCDMsPipeline's constructor also accepts, src/pipelines/RCDMs_pipeline.py:72-79;
    the class an SD-1.5 `scheduler/scheduler_config.json` names) with `skip_prk_steps=True`, the only mode Stable
    Diffusion checkpoints configure.

    Arithmetic of diffusers==0.24.0 `PNDMScheduler` (not vendored by the reference, not installed here: restated from
    the published class — "parity unpinned", pinned by closed-form known-answer tests in tests/test_scheduler.py):
      * "leading" spacing: _t = arange(n) * (N // n) + steps_offset; the PLMS list is [_t[:-1], _t[-2], _t[-1]] reversed,
        i.e. n + 1 model evaluations with the second timestep repeated (981, 961, 961, 941, ... for n = 50, offset 1);
      * step_plms: a linear multistep combination of the last <= 4 noise predictions (Adams-Bashforth weights
        1 | 1/2,1/2 (the repeated call, restarting from the saved first sample) | 3/2,-1/2 | 23/12,-16/12,5/12 |
        55/24,-59/24,37/24,-9/24), then
        x' = sqrt(a'/a) x - (a' - a) e / (a sqrt(1 - a') + sqrt(a (1 - a) a'))            (`_get_prev_sample`).
    `step()` is the host-visible torch form (stateful, like diffusers'); `plms_table()` gives the per-call rows the fused
    rcdm_cfg_pndm_step kernel consumes (rcdms_amd/sampler.py)."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, skip_prk_steps=False, set_alpha_to_one=False, prediction_type="epsilon",
                 timestep_spacing="leading", steps_offset=0):
        self._internal_dict = _FrozenDict(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, skip_prk_steps=skip_prk_steps,
            set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type, timestep_spacing=timestep_spacing,
            steps_offset=steps_offset)
        if prediction_type != "epsilon" or timestep_spacing != "leading":
            raise NotImplementedError("only epsilon prediction with 'leading' spacing (the reference's configuration)")
        if not skip_prk_steps:
            raise NotImplementedError("PNDMScheduler: only skip_prk_steps=True (PLMS, what Stable Diffusion checkpoints "
                                      "configure) is built; the Runge-Kutta warm-up steps are not")
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
        self.pndm_order = 4
        self.num_inference_steps = None
        self.timesteps = torch.arange(num_train_timesteps - 1, -1, -1, dtype=torch.int64)
        self.ets, self.counter, self.cur_sample = [], 0, None

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
        t = torch.arange(num_inference_steps, dtype=torch.int64) * ratio + c.steps_offset
        plms = torch.cat([t[:-1], t[-2:-1], t[-1:]]).flip(0).contiguous()
        self.timesteps = plms.to(device) if device is not None else plms
        self.ets, self.counter, self.cur_sample = [], 0, None

    def _ab(self, timestep, prev_timestep):
        """(sample coefficient, model-output coefficient) of `_get_prev_sample`, in fp64."""
        ac = self.alphas_cumprod.double()
        a_t = ac[timestep]
        a_p = ac[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod.double()
        denom = a_t * (1 - a_p).sqrt() + (a_t * (1 - a_t) * a_p).sqrt()
        return float((a_p / a_t).sqrt()), float(-(a_p - a_t) / denom)

    @staticmethod
    def _weights(n_hist):
        """Multistep weights over (newest, ..., oldest) for n_hist stored predictions."""
        return {1: (1.0,), 2: (1.5, -0.5), 3: (23 / 12, -16 / 12, 5 / 12), 4: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[n_hist]

    def plms_table(self):
        """fp32 [n_calls][12] for rcdm_cfg_pndm_step, one row per model evaluation of the current schedule:
        (a, b, w_now, w1, w2, w3, slot_now (-1: this prediction is not stored), s1, s2, s3, mode, 0) with
        x' = a x_src + b (w_now e + w1 hist[s1] + w2 hist[s2] + w3 hist[s3]); mode 1: also save x as the restart sample
        (first call), mode 2: x_src is that saved sample (second call: the repeated timestep), else x_src = x."""
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        rows, stored = [], []   # stored: ring slots of the kept predictions, oldest first
        for i, t in enumerate(self.timesteps.tolist()):
            if i == 1:
                a, b = self._ab(t + ratio, t)
                rows.append([a, b, 0.5, 0.5, 0.0, 0.0, -1, stored[-1], 0, 0, 2, 0])
                continue
            a, b = self._ab(t, t - ratio)
            slot = i % 4 if i < 2 else (i - 1) % 4
            prev = stored[-3:]
            stored = prev + [slot]
            w = self._weights(len(stored))
            hist = list(reversed(prev))            # newest stored first
            ws = list(w[1:]) + [0.0] * (3 - len(hist))
            ss = hist + [0] * (3 - len(hist))
            rows.append([a, b, w[0], ws[0], ws[1], ws[2], slot, ss[0], ss[1], ss[2], 1 if i == 0 else 0, 0])
        return torch.tensor(rows, dtype=torch.float32)

    def step(self, model_output, timestep, sample, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        t = int(timestep)
        prev_t = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        a, b = self._ab(t, prev_t)
        prev_sample = a * sample + b * model_output
        self.counter += 1
        return PNDMSchedulerOutput(prev_sample=prev_sample) if return_dict else (prev_sample,)


@dataclass
class UnCLIPSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: torch.Tensor = None


class UnCLIPScheduler:
    """The stage-1 prior's scheduler (reference: `UnCLIPScheduler.from_pretrained(..., subfolder="scheduler")`,
    stage1_batchtest_rcdms_model.py:101; used at src/pipelines/prior_pipeline.py:293-294,338-344).

    Arithmetic of diffusers==0.24.0 `UnCLIPScheduler` (not vendored, not installed: restated, "parity unpinned"):
    squaredcos_cap_v2 betas, timesteps = round(arange(n) * (N-1)/(n-1))[::-1], and `step` for prediction_type "sample" |
    "epsilon" with variance_type "fixed_small_log".  Defaults are the published Kandinsky-2.2 prior scheduler config
    (clip_sample True, clip_sample_range 10, prediction_type "sample").  `coefficients()` gives the per-step
    (k0, k1, k2) of  prev = k0 x0 + k1 x_t + k2 noise  that the fused rcdm_cfg_unclip_step kernel consumes."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, variance_type="fixed_small_log", clip_sample=True, clip_sample_range=10.0,
                 prediction_type="sample", beta_schedule="squaredcos_cap_v2"):
        if variance_type != "fixed_small_log" or beta_schedule != "squaredcos_cap_v2":
            raise NotImplementedError("UnCLIPScheduler: fixed_small_log / squaredcos_cap_v2 only (the prior's config)")
        if prediction_type not in ("sample", "epsilon"):
            raise NotImplementedError(f"prediction_type {prediction_type}")
        self._internal_dict = _FrozenDict(num_train_timesteps=num_train_timesteps, variance_type=variance_type,
                                          clip_sample=clip_sample, clip_sample_range=clip_sample_range,
                                          prediction_type=prediction_type, beta_schedule=beta_schedule)
        import math
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        n = num_train_timesteps
        betas = [min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)]
        self.betas = torch.tensor(betas, dtype=torch.float64)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.timesteps = torch.arange(n - 1, -1, -1, dtype=torch.int64)
        self.num_inference_steps = None

    @property
    def config(self):
        return self._internal_dict

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        n = self.config.num_train_timesteps
        ratio = (n - 1) / (num_inference_steps - 1)
        import numpy as np
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts).to(device) if device is not None else torch.from_numpy(ts)

    def _coef(self, t, prev_t):
        ac = self.alphas_cumprod
        a_t = ac[t]
        a_prev = ac[prev_t] if prev_t >= 0 else torch.tensor(1.0, dtype=torch.float64)
        b_t, b_prev = 1 - a_t, 1 - a_prev
        if prev_t == t - 1:
            beta, alpha = self.betas[t], self.alphas[t]
        else:
            beta = 1 - a_t / a_prev
            alpha = 1 - beta
        k0 = a_prev.sqrt() * beta / b_t
        k1 = alpha.sqrt() * b_prev / b_t
        std = torch.tensor(0.0, dtype=torch.float64)
        if t > 0:
            var = torch.clamp(b_prev / b_t * beta, min=1e-20)
            std = torch.exp(0.5 * torch.log(var))
        return k0, k1, std, a_t, b_t

    def coefficients(self):
        """fp32 [n_steps][3] = (k0, k1, std) for the current timesteps (prev_timestep = the next entry, None at the end:
        prior_pipeline.py:335-338)."""
        ts = self.timesteps.tolist()
        rows = []
        for i, t in enumerate(ts):
            prev = ts[i + 1] if i + 1 < len(ts) else t - 1
            k0, k1, std, _, _ = self._coef(t, prev)
            rows.append([float(k0), float(k1), float(std)])
        return torch.tensor(rows, dtype=torch.float32)

    def step(self, model_output, timestep, sample, prev_timestep=None, generator=None, return_dict=True, noise=None):
        t = int(timestep)
        prev_t = t - 1 if prev_timestep is None else int(prev_timestep)
        k0, k1, std, a_t, b_t = self._coef(t, prev_t)
        if self.config.prediction_type == "epsilon":
            x0 = (sample - b_t.sqrt().to(sample.dtype) * model_output) / a_t.sqrt().to(sample.dtype)
        else:
            x0 = model_output
        if self.config.clip_sample:
            x0 = torch.clamp(x0, -self.config.clip_sample_range, self.config.clip_sample_range)
        prev = k0.to(sample.dtype) * x0 + k1.to(sample.dtype) * sample
        if t > 0:
            if noise is None:
                noise = torch.randn(model_output.shape, dtype=model_output.dtype, device=model_output.device,
                                    generator=generator)
            prev = prev + std.to(sample.dtype) * noise
        return UnCLIPSchedulerOutput(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)
