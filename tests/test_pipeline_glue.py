"""CPU: the host-side glue of the mirrored pipeline class (no GPU, no kernels): constructor side effects, mask /
context bookkeeping incl. the reference's context-row ordering quirk, argument checks."""
import pytest
import torch
import torch.nn as nn

from rcdms_amd.scheduler import DDIMScheduler
from src.pipelines.RCDMs_pipeline import AnimationPipeline, RCDMsPipeline, RCDMsPipelineOutput, local_feature


class _Tag(nn.Module):
    """Stand-in context stack: returns the text rows plus a marker so provenance is visible."""

    def __init__(self, marker):
        super().__init__()
        self.marker = marker

    def forward(self, vis, text):
        return text + self.marker


class _FakeUNet:
    class config:
        sample_size = 64
    _programs = {}
    device = torch.device("cpu")


def make_pipe():
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear")
    assert sched.config.steps_offset == 0 and sched.config.clip_sample is True
    pipe = RCDMsPipeline(vae=None, text_encoder=None, tokenizer=None, unet=_FakeUNet(), local_module=_Tag(1000.0),
                         global_module=_Tag(2000.0), scheduler=sched)
    return pipe, sched


def test_alias_and_ctor_mutates_scheduler_config():
    assert AnimationPipeline is RCDMsPipeline
    pipe, sched = make_pipe()
    assert sched.config.steps_offset == 1 and sched.config.clip_sample is False      # RCDMs_pipeline.py:84-109
    assert pipe.vae_scale_factor == 8
    assert isinstance(RCDMsPipelineOutput(videos=torch.zeros(1)).videos, torch.Tensor)


def test_ctor_accepts_pndm_and_forces_its_offset():
    """The constructor's scheduler annotation (RCDMs_pipeline.py:72-79) includes PNDMScheduler; its steps_offset is forced
    to 1 like DDIM's (:84-97), after which the PLMS timesteps are 981, 961, 961, ... for 50 steps."""
    from rcdms_amd.scheduler import PNDMScheduler
    sched = PNDMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", skip_prk_steps=True)
    assert sched.config.steps_offset == 0
    pipe = RCDMsPipeline(vae=None, text_encoder=None, tokenizer=None, unet=_FakeUNet(), local_module=_Tag(1.0),
                         global_module=_Tag(2.0), scheduler=sched)
    assert sched.config.steps_offset == 1 and pipe.scheduler is sched
    sched.set_timesteps(50)
    assert sched.timesteps[:3].tolist() == [981, 961, 961] and "eta" not in pipe.prepare_extra_step_kwargs(None, 0.0)


def test_encode_mask_and_context_order_quirk():
    pipe, _ = make_pipe()
    mask = torch.zeros(5, 4, 4); mask[0] = 1.0                       # first frame seen
    m10 = pipe.encode_mask(mask, 1, True)
    assert m10.shape == (10, 4, 4) and torch.equal(m10[:5], m10[5:])
    text = torch.arange(10, dtype=torch.float32).view(10, 1, 1).expand(10, 3, 2).clone()   # row id in every entry
    seen, unseen = pipe.mask2list_label(m10, text, True)
    assert seen[:, 0, 0].tolist() == [0.0, 5.0] and unseen[:, 0, 0].tolist() == [1, 2, 3, 4, 6, 7, 8, 9]
    ctx = pipe.build_context(text, m10, torch.zeros(2, 1, 1), torch.zeros(8, 1, 1))
    # reference order: cat([seen rows, unseen rows]) -> [u0, c0, u1..u4, c1..c4]
    assert ctx[:, 0, 0].tolist() == [1000.0, 1005.0, 2001.0, 2002.0, 2003.0, 2004.0, 2006.0, 2007.0, 2008.0, 2009.0]
    fixed = pipe.build_context(text, m10, torch.zeros(2, 1, 1), torch.zeros(8, 1, 1), fix_context_order=True)
    assert fixed[:, 0, 0].tolist() == [1000.0, 2001.0, 2002.0, 2003.0, 2004.0, 1005.0, 2006.0, 2007.0, 2008.0, 2009.0]


def test_mixed_mask_rejected_and_input_checks():
    pipe, _ = make_pipe()
    bad = torch.zeros(10, 4, 4); bad[0, 0, 0] = 1.0
    with pytest.raises(ValueError, match="please check mask label"):
        pipe.mask2list_label(bad, torch.zeros(10, 3, 2), True)
    with pytest.raises(ValueError):
        pipe.check_inputs(3, 512, 512, 1)
    with pytest.raises(ValueError):
        pipe.check_inputs("a", 510, 512, 1)
    with pytest.raises(ValueError):
        pipe.check_inputs("a", 512, 512, 0)
    lat = pipe.prepare_latents(1, 4, 5, 512, 512, torch.float32, torch.device("cpu"), torch.Generator().manual_seed(0))
    assert lat.shape == (1, 4, 5, 64, 64)
    with pytest.raises(ValueError):
        pipe.prepare_latents(1, 4, 5, 512, 512, torch.float32, torch.device("cpu"), None, latents=torch.zeros(1, 4, 5, 8, 8))


def test_local_feature_matches_reference_stack_shape():
    m = local_feature(text_dim=768, vis_dim=1664, hidden_dim=768, num_heads=8)
    out = m(torch.randn(2, 257, 1664), torch.randn(2, 85, 768))
    assert out.shape == (2, 85, 768)
    assert sorted(k.split(".")[0] for k in m.state_dict()) == sorted(
        ["text_fc"] * 2 + ["vis_fc"] * 2 + ["multihead_attn"] * 4)
