"""Block-level runners behind the mirrored nn.Module classes' own forward() (src/models/{resnet,attention,motion_module,
unet_blocks}.py: ResnetBlock3D, Transformer3DModel, CrossAttention, BasicTransformerBlock, VanillaTemporalModule,
Down/Upsample3D, InflatedConv3d, InflatedGroupNorm): pack, plan, run eagerly on the HIP path — no CPU / torch fallback."""
import torch

from . import hip
from .emit_blocks import emit_basic_block, emit_ctx_kv, emit_motion, emit_resnet, emit_transformer, wino_level
from .emit_ops import emit_conv3x3, emit_flash_attn, emit_gemm, emit_groupnorm
from .packer import Packer, pack_attention, pack_basic_block, pack_motion, pack_resnet, pack_transformer
from .plan import Geo, Plan, Rows

def rows_from_ncfhw(x):
    b, c, f, h, w = x.shape
    x = x.detach().to(torch.float32).contiguous()
    cp = (c + 7) // 8 * 8
    rows = torch.empty(b * f * h * w, cp, dtype=torch.float16, device=x.device)
    hip.ncfhw_to_rows(x.data_ptr(), b, c, f, h, w, rows.data_ptr(), cp, cp)
    return rows


def ncfhw_from_rows(rows, ld, b, c, f, h, w):
    out = torch.empty(b, c, f, h, w, dtype=torch.float32, device=rows.device)
    hip.rows_to_ncfhw(rows.data_ptr(), ld, b, c, f, h, w, out.data_ptr())
    return out


class _Holder:
    def __init__(self, t):
        self.t = t
        self.nbytes = t.numel() * t.element_size()

    @property
    def ptr(self):
        return self.t.data_ptr()


def _as_rows(t, M, C, ld):
    return Rows(_Holder(t), 0, M, C, ld)


def _tokens16(x):
    """(B, L, C) any float dtype -> f16 rows [B*L][C] on the device."""
    x32 = x.detach().to(torch.float32).contiguous()
    rows = torch.empty(x32.shape[0] * x32.shape[1], x32.shape[2], dtype=torch.float16, device=x32.device)
    hip.pack_f16(x32.data_ptr(), rows.data_ptr(), x32.numel())
    return rows, x32


def run_tokens(kind, sd, x, ctx=None, heads=8):
    """Run a token-level reference module on the HIP path.  x (B, Lq, C) -> (B, Lq, C) fp32.
      "attention": CrossAttention.forward (attention.py:113-168) — self-attention when ctx is None;
      "block":     BasicTransformerBlock.forward (attention.py:479-526), ctx (B, L, D) when the block has attn2."""
    if not x.is_cuda:
        raise hip.RcdmError("rcdms_amd runs on MI355X only: input tensor is not on a CUDA/HIP device (no CPU fallback)")
    if x.dim() != 3:
        raise ValueError(f"expected (batch, tokens, channels), got {tuple(x.shape)}")
    device = x.device
    B, Lq, C = x.shape
    plan = Plan(device)
    pk = Packer(sd, device)
    xr_t, x32 = _tokens16(x)
    M = B * Lq
    tok = _as_rows(xr_t, M, C, C)
    c16 = None
    if ctx is not None:
        if ctx.dim() != 3 or ctx.shape[0] != B:
            raise ValueError(f"encoder_hidden_states must be (batch, L, D) with batch {B}, got {tuple(ctx.shape)}")
        c_t, c32 = _tokens16(ctx.to(device))
        L = ctx.shape[1]
        c16 = _as_rows(c_t, B * L, ctx.shape[2], ctx.shape[2])
    if kind == "attention":
        inner = sd["to_q.weight"].shape[0]
        d_head = inner // heads
        out = plan.rows("out", M, sd["to_out.0.weight"].shape[0], unique=True)
        ao = plan.rows("attn_out", M, inner)
        if c16 is None:
            w = pack_attention(pk, "", True)
            qkv = plan.rows("qkv", M, 3 * inner)
            emit_gemm(plan, tok, w.qkv, 3 * inner, C, qkv, bias=w.qkv_b)
            emit_flash_attn(plan, qkv.cols(0, inner), qkv.cols(inner, inner), qkv.cols(2 * inner, inner), B, heads, Lq, Lq,
                            d_head, ao, wide=True)   # (a bare CrossAttention.forward: no LayerNorm in front bounds its rows)
        else:
            w = pack_attention(pk, "", False)
            q = plan.rows("qkv", M, inner)
            emit_gemm(plan, tok, w.q, inner, C, q, bias=w.q_b)
            kv = plan.rows("ctx_kv", c16.M, 2 * inner, unique=True)
            emit_gemm(plan, c16, w.kv, 2 * inner, c16.C, kv, bias=w.kv_b)
            emit_flash_attn(plan, q, kv.cols(0, inner), kv.cols(inner, inner), B, heads, Lq, ctx.shape[1], d_head, ao)
        emit_gemm(plan, ao, w.o, out.C, inner, out, bias=w.o_b)
        res_rows = out
    elif kind == "block":
        w = pack_basic_block(pk, "")
        kv = None
        if w.has_cross:
            if c16 is None:
                raise ValueError("this BasicTransformerBlock has a cross-attention: encoder_hidden_states is required")
            kv = plan.rows("ctx_kv", c16.M, 2 * C, unique=True)
            img = emit_ctx_kv(plan, w, c16, kv, B, ctx.shape[1], heads)
        a = plan.rows("norm", M, C)
        emit_basic_block(plan, w, tok, B, Lq, heads, a, kv, ctx.shape[1] if ctx is not None else 0, ctx_img=img if kv is not None else None)
        res_rows = tok
    else:
        raise ValueError(kind)
    pk.done()
    plan.materialize()
    plan.run()
    torch.cuda.synchronize(device)
    t16 = res_rows.buf.t.view(torch.float16)[:M * res_rows.ld].view(M, res_rows.ld)[:, :res_rows.C]
    return t16.float().reshape(B, Lq, res_rows.C)


def run_block(kind, sd, x, device=None, **kw):
    """Run ONE reference block on the HIP path: kind in {"resnet","transformer","motion","down","up","conv"}.
    x (b,C,f,H,W); returns (b,C',f,H',W') fp32.  Used by the mirrored nn.Module classes' forward()."""
    if not x.is_cuda:
        raise hip.RcdmError("rcdms_amd runs on MI355X only: input tensor is not on a CUDA/HIP device (no CPU fallback)")
    device = x.device
    b, c, f, H, W = x.shape
    geo = Geo(b, f, H, W)
    plan = Plan(device)
    pk = Packer(sd, device)
    xr_t = rows_from_ncfhw(x)
    xr = _as_rows(xr_t, geo.M, c, xr_t.shape[1])
    groups = kw.get("groups", 32)
    if kind == "resnet":
        shp = sd["conv1.weight"].shape
        w = pack_resnet(pk, "", wino=wino_level(geo, shp[1], shp[0]))
        temb = kw["temb"].detach().to(device, torch.float32)
        tp_w = pk.mat_f16("time_emb_proj.weight")
        tp_b = pk.vec("time_emb_proj.bias")
        tproj = torch.empty(b, w.cout, dtype=torch.float32, device=device)
        for r0 in range(0, b, 8):
            r = min(8, b - r0)
            hip.small_linear(temb.data_ptr() + 4 * r0 * temb.shape[1], r, temb.shape[1], tp_w.data_ptr(),
                             tp_b.data_ptr(), w.cout, 1, 0, tproj.data_ptr() + 4 * r0 * w.cout)
        out = plan.rows("out", geo.M, w.cout, unique=True)
        emit_resnet(plan, w, xr, geo, (tproj, 0, w.cout), out, kw.get("eps", 1e-5), groups,
                    1.0 / kw.get("output_scale_factor", 1.0))
        oc, oh, ow = w.cout, H, W
    elif kind == "transformer":
        w = pack_transformer(pk, "")
        ctx = kw["ctx"].detach().to(device, torch.float32).contiguous()
        L = ctx.shape[1]
        ctx16 = plan.rows("ctx16", geo.n_img * L, w.ctx_dim, unique=True)
        kv = plan.rows("ctx_kv", geo.n_img * L, 2 * w.C, unique=True)
        plan.add(lambda: hip.pack_f16(ctx.data_ptr(), ctx16.ptr, ctx.numel()))
        img = emit_ctx_kv(plan, w, ctx16, kv, geo.n_img, L, kw["heads"])
        out = plan.rows("out", geo.M, c, unique=True)
        emit_transformer(plan, w, xr, geo, kv, L, kw["heads"], out, groups, ctx_img=img)
        oc, oh, ow = c, H, W
    elif kind == "motion":
        w = pack_motion(pk, "", kw["n_attn"])
        out = plan.rows("out", geo.M, c, unique=True)
        emit_motion(plan, w, xr, geo, kw["heads"], out, groups)
        oc, oh, ow = c, H, W
    elif kind == "groupnorm":   # InflatedGroupNorm.forward (resnet.py:21-29): nn.GroupNorm applied frame by frame
        out = plan.rows("out", geo.M, c, unique=True)
        emit_groupnorm(plan, xr, geo.n_img, geo.hw, pk.vec("weight"), pk.vec("bias"), kw["eps"], False, out, groups)
        oc, oh, ow = c, H, W
    elif kind in ("down", "up", "conv"):
        cw = sd["weight"] if kind == "conv" else sd["conv.weight"]
        cbk = "bias" if kind == "conv" else "conv.bias"
        oc = cw.shape[0]
        ocp = (oc + 7) // 8 * 8
        ksz = cw.shape[-1]
        stride = kw.get("stride", 2 if kind == "down" else 1)
        up = 1 if kind == "up" else 0
        cp = xr_t.shape[1]
        bias = None
        if cbk in sd and sd[cbk] is not None:
            bias = torch.cat([pk.vec(cbk), torch.zeros(ocp - oc, device=device)]).contiguous()
        if ksz == 3:
            wk = "weight" if kind == "conv" else "conv.weight"
            wt = pk.conv3x3(wk, cin_pad=cp, cout_pad=ocp)
            oh, ow = ((H << up) - 1) // stride + 1, ((W << up) - 1) // stride + 1
            out = plan.rows("out", geo.n_img * oh * ow, ocp, unique=True)
            emit_conv3x3(plan, Rows(xr.buf, 0, geo.M, cp, cp), geo.n_img, H, W, wt, cp, ocp, out, stride=stride, up=up,
                         bias=bias)
        elif ksz == 1:
            w2 = torch.zeros(ocp, cp, device=device)
            w2[:oc, :c] = pk.f32("weight").reshape(oc, c)
            wt = torch.empty(ocp, cp, dtype=torch.float16, device=device)
            hip.pack_f16(w2.data_ptr(), wt.data_ptr(), w2.numel())
            oh, ow = H, W
            out = plan.rows("out", geo.M, ocp, unique=True)
            emit_gemm(plan, Rows(xr.buf, 0, geo.M, cp, cp), wt, ocp, cp, out, bias=bias)
            plan.keep.append(w2)
        else:
            raise hip.RcdmError(f"conv kernel size {ksz} not supported on the HIP path")
    else:
        raise ValueError(kind)
    pk.done()
    plan.materialize()
    plan.run()
    res = ncfhw_from_rows(out.buf.t.view(torch.float16), out.ld, b, oc, f, oh, ow)
    torch.cuda.synchronize(device)
    return res
