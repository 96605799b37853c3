"""CPU: host-side logic of the launch planner that needs no GPU."""
import math

import torch

from oracle import unet_oracle as O


class _FakePacker:
    def __init__(self, sd):
        self.sd = sd

    def has(self, k):
        return k in self.sd


def test_attn_score_bound_dominates_every_score():
    """engine.attn_score_bound (the plan-time guard of the matrix-pipe-softmax flash kernel, rcdm.h: |scaled score| < 2^15):
    for random weights with per-channel gains and offsets, the bound is >= the largest |scale log2(e) q.k| of
    LayerNorm -> to_q / to_k on adversarially scaled inputs (LayerNorm makes the row norm input-independent)."""
    from rcdms_amd.engine import attn_score_bound, MSUB_SCORE_LIMIT
    g = torch.Generator().manual_seed(0)
    C, heads = 64, 4
    d = C // heads
    for trial, wscale in enumerate((1.0, 7.0, 300.0)):
        sd = {"a.to_q.weight": torch.randn(C, C, generator=g) * wscale * C ** -0.5,
              "a.to_k.weight": torch.randn(C, C, generator=g) * wscale * C ** -0.5,
              "a.to_q.bias": torch.randn(C, generator=g) * 0.3}
        gamma = 1.0 + 0.5 * torch.randn(C, generator=g)
        beta = 0.3 * torch.randn(C, generator=g)
        bound = attn_score_bound(_FakePacker(sd), "a.", (gamma, beta), heads)
        x = torch.randn(3, 50, C, generator=g, dtype=torch.float64) * torch.tensor([1e-3, 1.0, 1e4], dtype=torch.float64)[:, None, None]
        x[1, :, 5] += 40.0   # an outlier channel
        ln = torch.nn.functional.layer_norm(x, (C,), gamma.double(), beta.double(), 1e-5)
        q = (ln @ sd["a.to_q.weight"].double().t() + sd["a.to_q.bias"].double()).view(3, 50, heads, d)
        k = (ln @ sd["a.to_k.weight"].double().t()).view(3, 50, heads, d)
        smax = torch.einsum("blhd,bmhd->bhlm", q, k).abs().max().item() * d ** -0.5 * math.log2(math.e)
        assert smax <= bound * (1 + 1e-9), (trial, smax, bound)
        assert bound < 60 * max(smax, 1.0)            # ... and is not vacuous (Cauchy-Schwarz + Frobenius: a small factor)
        assert (bound >= MSUB_SCORE_LIMIT) == (wscale >= 300.0)
    assert attn_score_bound(_FakePacker(sd), "a.", None, heads) == float("inf")   # no LayerNorm in front: nothing bounds the rows


def test_denoise_loop_refuses_non_ddim_timestep_lists():
    """DenoiseLoop's fused DDIM step must not run on a scheduler whose timestep list is not DDIM's (a diffusers
    PNDMScheduler object has alphas_cumprod but no plms_table(): N + 1 timesteps, the second one repeated) — and accepts the
    product's own DDIM / PNDM classes.  Host logic only: the launch plan is built on first load()."""
    import pytest
    from rcdms_amd.sampler import DenoiseLoop
    from rcdms_amd.scheduler import DDIMScheduler, PNDMScheduler

    class FakeUNet:
        device = torch.device("cpu")

    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1)
    lp = DenoiseLoop(FakeUNet(), 1, 5, 8, 8, 13, 2.0, DDIMScheduler(clip_sample=False, **kw), 10)
    assert lp.T == 10 and not lp.pndm and tuple(lp.coef.shape) == (10, 4)
    lp = DenoiseLoop(FakeUNet(), 1, 5, 8, 8, 13, 2.0, PNDMScheduler(skip_prk_steps=True, **kw), 10)
    assert lp.T == 11 and lp.pndm

    real = PNDMScheduler(skip_prk_steps=True, **kw)

    class DiffusersLikePNDM:           # what `from diffusers import PNDMScheduler` hands the pipeline: no plms_table()
        config = real.config
        alphas_cumprod = real.alphas_cumprod
        init_noise_sigma = 1.0

        def set_timesteps(self, n, device=None):
            real.set_timesteps(n)
            self.timesteps = real.timesteps

    with pytest.raises(NotImplementedError, match="not a DDIM schedule"):
        DenoiseLoop(FakeUNet(), 1, 5, 8, 8, 13, 2.0, DiffusersLikePNDM(), 10)

    class OddStride(DDIMScheduler):    # a DDIM-shaped object whose list does not have the constant stride
        def set_timesteps(self, n, device=None):
            super().set_timesteps(n, device)
            self.timesteps = self.timesteps.clone()
            self.timesteps[1] -= 3

    with pytest.raises(NotImplementedError, match="not a DDIM schedule"):
        DenoiseLoop(FakeUNet(), 1, 5, 8, 8, 13, 2.0, OddStride(clip_sample=False, **kw), 10)
