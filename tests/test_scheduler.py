"""CPU: known-answer tests for the restated third-party arithmetic (diffusers 0.24.0 DDIMScheduler, Timesteps):
"parity unpinned" by the reference (it holds no tests), pinned here by closed forms."""
import math

import torch

from oracle import unet_oracle as O
from rcdms_amd.scheduler import DDIMScheduler


def make():
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")


def test_timesteps_leading_with_offset():
    s = make()
    s._internal_dict["steps_offset"] = 1
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [981, 961, 941] and s.timesteps[-1].item() == 1 and len(s.timesteps) == 50
    s.set_timesteps(20)
    assert s.timesteps[:2].tolist() == [951, 901] and s.timesteps[-1].item() == 1


def test_alphas_and_step_closed_form():
    s = make(); s._internal_dict["steps_offset"] = 1; s._internal_dict["clip_sample"] = False
    s.set_timesteps(50)
    betas = torch.linspace(0.00085, 0.012, 1000)
    ac = torch.cumprod(1 - betas, 0)
    assert torch.allclose(s.alphas_cumprod, ac)
    x, eps = torch.full((2, 3), 0.7), torch.full((2, 3), -0.2)
    out = s.step(eps, 981, x).prev_sample
    a_t, a_p = ac[981].double(), ac[961].double()
    x0 = (0.7 - math.sqrt(1 - a_t) * -0.2) / math.sqrt(a_t)
    want = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * -0.2
    assert abs(out[0, 0].item() - want) < 1e-5
    last = s.step(eps, 1, x).prev_sample          # prev timestep < 0 -> final_alpha_cumprod = 1 -> returns x0
    x0l = (0.7 - math.sqrt(1 - ac[1].double()) * -0.2) / math.sqrt(ac[1].double())
    assert abs(last[0, 0].item() - x0l) < 1e-5
    coef = s.coefficients()
    assert coef.shape == (50, 4) and abs(coef[0, 0].item() - math.sqrt(a_t)) < 1e-6 and coef[-1, 3].item() == 0.0


def test_oracle_scheduler_agrees_with_product_scheduler():
    s = make(); s._internal_dict["steps_offset"] = 1; s._internal_dict["clip_sample"] = False
    o = O.DDIMOracle()
    for n in (20, 50):
        s.set_timesteps(n); o.set_timesteps(n)
        assert torch.equal(s.timesteps, o.timesteps)
        g = torch.Generator().manual_seed(n)
        x, e = torch.randn(4, 5, generator=g), torch.randn(4, 5, generator=g)
        for t in (int(s.timesteps[0]), int(s.timesteps[n // 2]), 1):
            assert torch.allclose(s.step(e, t, x).prev_sample, o.step(e, t, x), atol=1e-6)


def test_timestep_embedding_known_answers():
    e = O.timestep_embedding(torch.tensor([981, 0]), 320)
    assert e.shape == (2, 320)
    assert abs(e[0, 0].item() - math.cos(981.0)) < 1e-5          # flip_sin_to_cos: cos first, w_0 = 1
    assert abs(e[0, 160].item() - math.sin(981.0)) < 1e-5
    w159 = math.exp(-math.log(10000.0) * 159 / 160)
    assert abs(e[0, 159].item() - math.cos(981.0 * w159)) < 1e-5
    assert torch.allclose(e[1, :160], torch.ones(160)) and torch.allclose(e[1, 160:], torch.zeros(160))
