"""numerics_report: the first thing to run after load_state_dict() of a REAL checkpoint.

The HIP path stores every activation between kernels as f16 (limit 65504), takes the d = 40 self-attention's softmax
argument from the matrix pipe where a weight-norm bound allows it (|scaled score| < 2^15, else the RCDM_ATTN_WIDE_RANGE
kernel), and below the 64x64 level defers every LayerNorm into the GEMM behind it (rstd (x W'^T) - (mean rstd) S + b': a
cancellation when |mean rstd S| is large against the result).  All parity evidence in this repository is on synthetic weight
families (no SD-1.5 / RCDMs checkpoint exists here); trained weights are heavier-tailed.  This module runs ONE
UNet3DConditionModel.forward (reference: src/models/unet.py:322-463, with the weights loaded the way unet.py:465-509 and
stage2_batchtest_rcdms_model.py:225-243 load them) op by op on the launch plan and reports, for the given input:

  buffers     per plan buffer of f16 rows: the largest |value| any launch left in it, the op that produced it, and the
              headroom 65504 / max
  attention   per self-attention site: the data-independent weight-norm bound of |scale log2(e) q.k| (engine.attn_score_bound),
              whether the site therefore runs the wide-range kernel, and a data-dependent bound for THIS input
              (max |q_i| max |k_j| per head, Cauchy-Schwarz)
  layernorm   per deferred-LayerNorm consumer: max |mean rstd| over its rows, max |S| over its columns, and their product
              (the term the epilogue subtracts); the consumer's fp32 accumulators hold ~7 decimal digits

Anything non-finite in any f16 buffer after any launch raises RcdmError naming the launch."""
import torch

from . import hip

F16_MAX = 65504.0


def _f16_view(buf):
    return buf.t.view(torch.float16)


def _rows_view(rows):
    t = rows.buf.t.view(torch.float16)
    return torch.as_strided(t, (rows.M, rows.C), (rows.ld, 1), rows.off)


def numerics_report(prog, sample, timestep, ctx, raise_on_nonfinite=True):
    """prog: rcdms_amd.engine.UNetProgram (UNet3DConditionModel.numerics_report builds it).  Returns a dict with the
    lists "buffers", "attention", "layernorm", the scalars "min_headroom" / "max_ln_cancel" / "n_ops", and "text"
    (the formatted table).  Runs the plan eagerly, one launch at a time (a few seconds): a diagnostic, not a hot path."""
    b, f, H, W = prog.b, prog.f, prog.H, prog.W
    dev = prog.device
    prog.set_context(ctx, force=True)
    x = sample.detach().to(dev, torch.float32).contiguous()
    t = torch.as_tensor(timestep, dtype=torch.float32, device=dev).reshape(-1)
    plan = prog.plan
    bufs = [bf for bf in plan.bufs.values() if bf.f16 and bf.t is not None]
    lnx_at = {}
    for i, A, S, tag in plan.lnx_sites:
        lnx_at.setdefault(i, []).append((A, S, tag))
    attn_at = {s[0]: s for s in plan.attn_sites}
    per_op, ln_rows, at_rows = [], [], []
    cur = torch.cuda.current_stream(dev)
    prog.stream.wait_stream(cur)
    with torch.cuda.stream(prog.stream), torch.no_grad():
        prog.t_dev.copy_(t.expand(b))
        from .unet_program import CIN_PAD
        hip.ncfhw_to_rows(x.data_ptr(), b, prog.in_channels, f, H, W, prog.x_in.ptr, prog.x_in.ld, CIN_PAD)
        for i, op in enumerate(plan.ops):
            op()
            # (one small reduction per buffer and launch; everything stays on the device until the end)
            per_op.append(torch.stack([torch.stack(_f16_view(bf).aminmax()).float().abs().amax() for bf in bufs]))
            for A, S, tag in lnx_at.get(i, ()):
                xa = _rows_view(A).float()
                mean = xa.mean(dim=1)
                rstd = torch.rsqrt(xa.var(dim=1, unbiased=False) + 1e-5)
                ln_rows.append((i, tag, (mean * rstd).abs().amax(), S.abs().amax()))
            if i in attn_at:
                _, tag, bound, wide, q, k, heads, d = attn_at[i]
                qn = _rows_view(q).float().reshape(q.M, heads, d).norm(dim=2).amax(dim=0)
                kn = _rows_view(k).float().reshape(k.M, heads, d).norm(dim=2).amax(dim=0)
                at_rows.append((i, tag, bound, wide, (qn * kn).amax() * d ** -0.5 * 1.4426950408889634))
    cur.wait_stream(prog.stream)
    x.record_stream(prog.stream)
    table = torch.stack(per_op).cpu()                      # [n_ops][n_bufs]
    bad = ~torch.isfinite(table)
    if bad.any() and raise_on_nonfinite:
        i = int(bad.any(dim=1).nonzero()[0])
        j = int(bad[i].nonzero()[0])
        raise hip.RcdmError(f"numerics_report: non-finite values in buffer '{bufs[j].name}' after launch {i} ({plan.tags[i]}) — "
                            "this checkpoint / input leaves the f16 range of the HIP path")
    rep = {"buffers": [], "attention": [], "layernorm": [], "n_ops": len(plan.ops)}
    for j, bf in enumerate(bufs):
        col = table[:, j]
        m = float(col.max())
        i = int((col >= m).nonzero()[0])                   # the first launch after which the buffer held its maximum
        rep["buffers"].append(dict(buffer=bf.name, max_abs=m, op=i, tag=plan.tags[i], headroom=F16_MAX / m if m > 0 else float("inf")))
    for i, tag, bound, wide, seen in at_rows:
        rep["attention"].append(dict(op=i, tag=tag, weight_bound=bound, wide_range=wide, input_bound=float(seen)))
    for i, tag, mr, smax in ln_rows:
        mr, smax = float(mr), float(smax)
        rep["layernorm"].append(dict(op=i, tag=tag, max_mean_rstd=mr, max_colsum=smax, cancel=mr * smax))
    rep["min_headroom"] = min((r["headroom"] for r in rep["buffers"]), default=float("inf"))
    rep["max_ln_cancel"] = max((r["cancel"] for r in rep["layernorm"]), default=0.0)
    rep["wide_sites"] = sum(1 for r in rep["attention"] if r["wide_range"])
    lines = [f"numerics_report: {len(plan.ops)} launches, {len(bufs)} f16 buffers; min headroom x{rep['min_headroom']:.1f} "
             f"(f16 limit {F16_MAX:.0f}); self-attention sites {len(rep['attention'])} ({rep['wide_sites']} on the wide-range kernel); "
             f"deferred LayerNorms {len(rep['layernorm'])}, max |mean rstd S| {rep['max_ln_cancel']:.1f}",
             f"{'buffer':16s} {'max |x|':>10s} {'headroom':>9s}  produced by"]
    for r in sorted(rep["buffers"], key=lambda r: -r["max_abs"])[:12]:
        lines.append(f"{r['buffer']:16s} {r['max_abs']:10.2f} {r['headroom']:9.1f}  op {r['op']} {r['tag']}")
    lines.append(f"{'self-attention':44s} {'weight bound':>12s} {'this input':>11s}  kernel")
    for r in rep["attention"]:
        lines.append(f"op {r['op']:4d} {r['tag'][:38]:38s} {r['weight_bound']:12.1f} {r['input_bound']:11.1f}  "
                     f"{'wide range (fp32 argument)' if r['wide_range'] else 'matrix-pipe argument (< 2^15)'}")
    lines.append(f"{'deferred LayerNorm consumer':52s} {'|mean rstd|':>11s} {'max |S|':>9s} {'product':>9s}")
    for r in sorted(rep["layernorm"], key=lambda r: -r["cancel"])[:12]:
        lines.append(f"op {r['op']:4d} {r['tag'][:46]:46s} {r['max_mean_rstd']:11.3f} {r['max_colsum']:9.2f} {r['cancel']:9.2f}")
    rep["text"] = "\n".join(lines)
    return rep
