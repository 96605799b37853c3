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


def test_clip_sample_follows_diffusers_semantics():
    """diffusers 0.24 DDIMScheduler.step: clip_sample clamps the predicted x0; epsilon is re-derived from the clipped x0
    ONLY with use_clipped_model_output=True.  Closed form on a sample whose x0 leaves [-1, 1]."""
    s = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", steps_offset=1)   # clip_sample=True
    s.set_timesteps(50)
    ac = s.alphas_cumprod.double()
    a_t, a_p = ac[981], ac[961]
    x, eps = torch.full((1, 2), 0.9), torch.full((1, 2), -0.3)
    x0 = (0.9 - math.sqrt(1 - a_t) * -0.3) / math.sqrt(a_t)
    assert x0 > 1.0
    keep = s.step(eps, 981, x).prev_sample[0, 0].item()
    assert abs(keep - (math.sqrt(a_p) * 1.0 + math.sqrt(1 - a_p) * -0.3)) < 1e-5            # clipped x0, ORIGINAL eps
    redo = s.step(eps, 981, x, use_clipped_model_output=True).prev_sample[0, 0].item()
    eps2 = (0.9 - math.sqrt(a_t) * 1.0) / math.sqrt(1 - a_t)
    assert abs(redo - (math.sqrt(a_p) * 1.0 + math.sqrt(1 - a_p) * eps2)) < 1e-5


def test_geglu_half_order_known_answer_oracle():
    """GEGLU = hidden * gelu(gate) with hidden the FIRST half of the projection (diffusers 0.24 `chunk(2, dim=-1)`):
    identity hidden rows, zero gate rows with bias 10 (gelu(10) == 10 in fp32), identity output layer -> 10 x."""
    C = 6
    sd = {"net.0.proj.weight": torch.cat([torch.eye(C), torch.zeros(C, C)]),
          "net.0.proj.bias": torch.cat([torch.zeros(C), torch.full((C,), 10.0)]),
          "net.2.weight": torch.eye(C), "net.2.bias": torch.zeros(C)}
    x = torch.randn(4, C, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(O.feed_forward_geglu(sd, "", x), 10.0 * x, rtol=1e-6, atol=1e-6)


# ---- PNDM / PLMS (diffusers 0.24.0 PNDMScheduler(skip_prk_steps=True); RCDMs_pipeline.py:72-79 accepts it) ----------------

def make_pndm(**kw):
    from rcdms_amd.scheduler import PNDMScheduler
    args = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", skip_prk_steps=True, steps_offset=1)
    args.update(kw)
    return PNDMScheduler(**args)


def test_pndm_timesteps_repeat_the_second_one():
    s = make_pndm()
    s.set_timesteps(50)
    ts = s.timesteps.tolist()
    assert len(ts) == 51 and ts[:4] == [981, 961, 961, 941] and ts[-2:] == [21, 1]
    s.set_timesteps(20)
    assert s.timesteps.tolist()[:4] == [951, 901, 901, 851] and len(s.timesteps) == 21
    import pytest
    with pytest.raises(NotImplementedError):
        make_pndm(skip_prk_steps=False)


def test_pndm_constant_eps_is_the_ddim_trajectory():
    """With a constant prediction every multistep weight set sums to one, so PLMS degenerates to x' = a x + b e per
    call, and `_get_prev_sample`'s (a, b) are algebraically DDIM's (eta 0): the 51-call PNDM trajectory equals the 50-step
    DDIM one — the repeated second call restarts from the saved first sample and lands on the same point."""
    s = make_pndm()
    d = make()
    d._internal_dict["steps_offset"] = 1; d._internal_dict["clip_sample"] = False
    d._internal_dict["set_alpha_to_one"] = False
    d.final_alpha_cumprod = d.alphas_cumprod[0]
    s.set_timesteps(50); d.set_timesteps(50)
    e = torch.full((2, 3), 0.3, dtype=torch.float64)
    x = torch.full((2, 3), -0.9, dtype=torch.float64)
    xd = x.clone()
    traj = []
    for t in s.timesteps.tolist():
        x = s.step(e, t, x).prev_sample
        traj.append(x)
    dd = []
    for t in d.timesteps.tolist():
        xd = d.step(e, t, xd).prev_sample
        dd.append(xd)
    assert torch.allclose(traj[0], dd[0], atol=1e-6) and torch.allclose(traj[1], dd[0], atol=1e-6)   # warm-up pair = one DDIM step
    for k in range(1, 50):
        assert torch.allclose(traj[k + 1], dd[k], atol=1e-5), k


def test_pndm_multistep_weights_and_known_answer():
    s = make_pndm()
    for n in (1, 2, 3, 4):
        assert abs(sum(s._weights(n)) - 1.0) < 1e-12
    assert s._weights(4) == (55 / 24, -59 / 24, 37 / 24, -9 / 24)
    s.set_timesteps(50)
    ac = torch.cumprod(1 - torch.linspace(0.00085, 0.012, 1000, dtype=torch.float64), 0)
    a_t, a_p = ac[981], ac[961]
    a, b = s._ab(981, 961)
    assert abs(a - math.sqrt(a_p / a_t)) < 1e-6
    assert abs(b + (a_p - a_t) / (a_t * math.sqrt(1 - a_p) + math.sqrt(a_t * (1 - a_t) * a_p))) < 1e-6
    # last call: prev timestep < 0 -> final_alpha_cumprod = alphas_cumprod[0] (set_alpha_to_one False, PNDM's default)
    a1, _ = s._ab(1, -19)
    assert abs(a1 - math.sqrt(ac[0] / ac[1])) < 1e-5
    # third call (second stored prediction): (3 e2 - e0) / 2
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, generator=g, dtype=torch.float64)
    e0, e1, e2 = (torch.randn(4, generator=g, dtype=torch.float64) for _ in range(3))
    x1 = s.step(e0, 981, x).prev_sample
    x1b = s.step(e1, 961, x1).prev_sample
    assert torch.allclose(x1b, a * x + b * (e0 + e1) / 2, atol=1e-7)           # restart from the saved sample
    x2 = s.step(e2, 961, x1b).prev_sample
    a2, b2 = s._ab(961, 941)
    assert torch.allclose(x2, a2 * x1b + b2 * (3 * e2 - e0) / 2, atol=1e-7)


def test_pndm_table_reproduces_step():
    """plms_table() (what rcdm_cfg_pndm_step consumes) replayed on the host == the stateful step(), call by call."""
    for n in (20, 50):
        s = make_pndm(beta_schedule="scaled_linear")
        s.set_timesteps(n)
        tab = s.plms_table().double()
        assert tab.shape == (n + 1, 12)
        g = torch.Generator().manual_seed(n)
        x = torch.randn(3, 5, generator=g, dtype=torch.float64)
        y = x.clone()
        hist, saved = [None] * 4, None
        for i, t in enumerate(s.timesteps.tolist()):
            e = torch.randn(3, 5, generator=g, dtype=torch.float64)
            x = s.step(e, t, x).prev_sample
            r = tab[i]
            mo = r[2] * e
            for w, si in ((r[3], int(r[7])), (r[4], int(r[8])), (r[5], int(r[9]))):
                if w != 0:
                    mo = mo + w * hist[si]
            src = saved if int(r[10]) == 2 else y
            if int(r[10]) == 1:
                saved = y.clone()
            if int(r[6]) >= 0:
                hist[int(r[6])] = e
            y = r[0] * src + r[1] * mo
            assert torch.allclose(x, y, atol=1e-6), i


def test_pndm_oracle_agrees_with_product_scheduler():
    for sched_kind in ("linear", "scaled_linear"):
        s = make_pndm(beta_schedule=sched_kind)
        o = O.PNDMOracle(beta_schedule=sched_kind)
        s.set_timesteps(25); o.set_timesteps(25)
        assert torch.equal(s.timesteps, o.timesteps)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(4, 5, generator=g, dtype=torch.float64)
        y = x.clone()
        for t in s.timesteps.tolist():
            e = torch.randn(4, 5, generator=g, dtype=torch.float64)
            x, y = s.step(e, t, x).prev_sample, o.step(e, t, y)
            assert torch.allclose(x, y.double(), atol=1e-5)
