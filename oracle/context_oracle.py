"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU fp32 restatement of the reference's two context builders,
`fine_stack` / `semantic_stack` (stage2_batchtest_rcdms_model.py:117-149): both are

    query     = text_fc(text_f)                 (k, L, 768)      :128 / :145
    key_value = vis_fc(vis_f)                   (k, Lv, 768)     :129 / :146
    out       = nn.MultiheadAttention(768, 8)(query^T, key_value^T, key_value^T)[0]^T        :130-132 / :147-149

with the MultiheadAttention spelled out (packed in_proj rows [q | k | v], scale d^-1/2, softmax over the Lv keys,
out_proj; no mask, no dropout in eval).  Pinned by tests/golden/ctx_*.npz, minted from the reference classes
themselves (oracle/make_golden.py --only ctx).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may
import this file; the product (rcdms_amd/context.py) never does."""
import torch


def context_stack_forward(sd, vis_f, text_f, num_heads=8):
    """sd: the stack's state dict (text_fc.*, vis_fc.*, multihead_attn.{in_proj_weight,in_proj_bias,out_proj.*})."""
    f32 = lambda k: sd[k].detach().to(torch.float32)
    q_in = text_f.float() @ f32("text_fc.weight").T + f32("text_fc.bias")       # (k, L, E)
    kv_in = vis_f.float() @ f32("vis_fc.weight").T + f32("vis_fc.bias")         # (k, Lv, E)
    E = q_in.shape[-1]
    w, b = f32("multihead_attn.in_proj_weight"), f32("multihead_attn.in_proj_bias")
    q = q_in @ w[:E].T + b[:E]
    k = kv_in @ w[E:2 * E].T + b[E:2 * E]
    v = kv_in @ w[2 * E:].T + b[2 * E:]
    n, L, _ = q.shape
    Lv = k.shape[1]
    d = E // num_heads
    split = lambda x, T: x.reshape(n, T, num_heads, d).permute(0, 2, 1, 3)      # (k, h, T, d)
    s = (split(q, L) * d ** -0.5) @ split(k, Lv).transpose(-1, -2)
    o = torch.softmax(s, dim=-1) @ split(v, Lv)
    o = o.permute(0, 2, 1, 3).reshape(n, L, E)
    return o @ f32("multihead_attn.out_proj.weight").T + f32("multihead_attn.out_proj.bias")
