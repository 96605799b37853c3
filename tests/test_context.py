"""Context builders: fine_stack / semantic_stack of stage2_batchtest_rcdms_model.py:117-149.
CPU: the oracle restatement against golden outputs minted from the reference's own classes, the mirrored classes'
state-dict layout, and the fail-loudly rule.  GPU: the HIP path (rcdms_amd.context) against oracle and golden."""
import os

import numpy as np
import pytest
import torch

from oracle import context_oracle as CO
from rcdms_amd import context, hip, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [("ctx_fine", context.fine_stack), ("ctx_semantic", context.semantic_stack),
         ("ctx_fine_ragged", context.fine_stack)]


def _case(name, cls):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    k, lv, vis_dim, seed = int(g["k"]), int(g["lv"]), int(g["vis_dim"]), int(g["seed"])
    m = cls(text_dim=768, vis_dim=vis_dim).eval()
    sd = synth.procedural_state_dict({n: v.shape for n, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    vis = synth.normal_tensor(name + ".vis", (k, lv, vis_dim), seed)
    text = synth.normal_tensor(name + ".text", (k, 85, 768), seed)
    return m, sd, vis, text, torch.from_numpy(g["out"])


@pytest.mark.parametrize("name,cls", CASES)
def test_oracle_matches_reference_golden(name, cls):
    _, sd, vis, text, want = _case(name, cls)
    got = CO.context_stack_forward(sd, vis, text)
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5), float((got - want).abs().max())


def test_state_dict_layout_is_the_reference_layout():
    m = context.fine_stack(text_dim=768, vis_dim=1664)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {
        "text_fc.weight": (768, 768), "text_fc.bias": (768,), "vis_fc.weight": (768, 1664), "vis_fc.bias": (768,),
        "multihead_attn.in_proj_weight": (2304, 768), "multihead_attn.in_proj_bias": (2304,),
        "multihead_attn.out_proj.weight": (768, 768), "multihead_attn.out_proj.bias": (768,)}


def test_cpu_module_fails_loudly():
    m = context.semantic_stack(text_dim=768, vis_dim=1280)
    with pytest.raises(hip.RcdmError):
        m(torch.zeros(1, 1, 1280), torch.zeros(1, 85, 768))


@pytest.mark.gpu
@pytest.mark.parametrize("name,cls", CASES)
def test_hip_context_stack(name, cls):
    m, sd, vis, text, want = _case(name, cls)
    m = m.to("cuda")
    got = m(vis.cuda(), text.cuda()).float().cpu()
    ref = CO.context_stack_forward(sd, vis, text)
    scale = float(want.abs().max())
    # f16 storage, fp32 accumulation, composed projection matrices rounded to f16 once: 4e-3 of the output range
    assert float((got - want).abs().max()) <= 4e-3 * scale, (float((got - want).abs().max()), scale)
    assert float((got - ref).abs().max()) <= 4e-3 * scale
    rel_rms = float(((got - want) ** 2).mean().sqrt() / (want ** 2).mean().sqrt())
    assert rel_rms < 2e-3, rel_rms


@pytest.mark.gpu
def test_hip_context_stack_rejects_bad_shapes():
    m = context.fine_stack(text_dim=768, vis_dim=1664).to("cuda")
    with pytest.raises(ValueError):
        m(torch.zeros(2, 257, 1664, device="cuda"), torch.zeros(3, 85, 768, device="cuda"))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 257, 1280, device="cuda"), torch.zeros(2, 85, 768, device="cuda"))
