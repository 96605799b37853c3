"""Weight packing, once per state dict: the reference's fp32 parameter layout (the 1286 keys of UNet3DConditionModel,
src/models/unet.py) -> the f16 kernel layouts of librcdm_hip.so (fused [q;k;v] / [k;v], GEGLU 16|16 row interleave,
conv3x3 tap-major, the phase images of the upsamplers, fragment-major weight streams of the row chains) and the
pack-time algebra (LayerNorm gamma / beta folded into the matrix behind it, proj_out composed with ff.net.2, conv_shortcut
behind conv2's taps).  torch is used for device memory and the one-time fp32 algebra only."""
import math

import torch

from . import hip
from . import switches as SW
from .plan import _NS


class Packer:
    def __init__(self, sd, device):
        self.sd, self.device = sd, torch.device(device)
        self._tmp = []

    def f32(self, key):
        t = self.sd[key].detach().to(self.device, torch.float32).contiguous()
        return t

    def has(self, key):
        return key in self.sd

    def vec(self, key):
        return self.f32(key)

    def mat_f16(self, *keys):
        """rows of several [n_i][K] matrices stacked -> f16 [sum n_i][K]"""
        src = torch.cat([self.f32(k).reshape(self.sd[k].shape[0], -1) for k in keys], dim=0).contiguous()
        dst = torch.empty(src.shape, dtype=torch.float16, device=self.device)
        hip.pack_f16(src.data_ptr(), dst.data_ptr(), src.numel())
        self._tmp.append(src)
        return dst

    def conv3x3(self, key, cin_pad=None, cout_pad=None):
        w = self.f32(key)
        cout, cin = w.shape[0], w.shape[1]
        cin_pad = cin_pad or cin
        if cout_pad and cout_pad > cout:
            w = torch.cat([w, torch.zeros(cout_pad - cout, cin, 3, 3, device=self.device)], dim=0).contiguous()
            cout = cout_pad
        dst = torch.empty(cout, 9 * cin_pad, dtype=torch.float16, device=self.device)
        hip.pack_conv3x3(w.data_ptr(), cout, cin, cin_pad, dst.data_ptr())
        self._tmp.append(w)
        return dst

    def ffz(self, ff2_key, ff2_bkey, po_key, po_bkey):
        """proj_out behind the feed-forward's second Linear as ONE matrix over [h | tok] (two linear maps in a row, no
        nonlinearity between: attention.py:514 + :361, motion_module.py:243 + :178):
            proj_out(tok + ff2 h + b2) + b_po = [W_po W_ff2 | W_po] [h | tok]^T + (W_po b2 + b_po)
        -> (f16 [C][5C], fp32 [C]); the products are formed in fp32 and rounded once.  None when FFZ is off."""
        if not SW.FFZ:
            return None
        w2, b2 = self.f32(ff2_key), self.f32(ff2_bkey)
        wpo = self.f32(po_key)
        wpo = wpo.reshape(wpo.shape[0], -1)
        bpo = self.f32(po_bkey)
        src = torch.cat([hip.matmul_f32(wpo, w2), wpo], dim=1).contiguous()
        dst = torch.empty(src.shape, dtype=torch.float16, device=self.device)
        hip.pack_f16(src.data_ptr(), dst.data_ptr(), src.numel())
        self._tmp.append(src)
        return _NS(W=dst, b=(hip.matmul_f32(wpo, b2) + bpo).contiguous())

    def conv3x3_wino(self, key):
        """Winograd F(2x2, 3x3) weights of a stride-1 conv (rcdm_conv3x3_wino): f16 [16][cout][cin] = G g G^T, formed in fp32."""
        w = self.f32(key)
        cout, cin = w.shape[0], w.shape[1]
        dst = torch.empty(16, cout, cin, dtype=torch.float16, device=self.device)
        hip.pack_conv3x3_wino(w.data_ptr(), cout, cin, dst.data_ptr())
        self._tmp.append(w)
        return dst

    def conv3x3_taps(self, key, cout_pad=None):
        """Tap-plane weights of a conv3x3 (rcdm_gemm + rcdm_conv_taps_gather): f16 [9 * cout][cin], row tap * cout + c =
        weight[c][:][ky][kx]; cout_pad: zero output channels appended first."""
        w = self.f32(key)
        cout, cin = w.shape[0], w.shape[1]
        if cout_pad and cout_pad > cout:
            w = torch.cat([w, torch.zeros(cout_pad - cout, cin, 3, 3, device=self.device)], dim=0).contiguous()
            cout = cout_pad
        src = w.permute(2, 3, 0, 1).reshape(9 * cout, cin).contiguous()
        dst = torch.empty(9 * cout, cin, dtype=torch.float16, device=self.device)
        hip.pack_f16(src.data_ptr(), dst.data_ptr(), src.numel())
        self._tmp += [w, src]
        return dst

    def conv3x3_up2(self, key):
        """Phase weights of an Upsample3D conv (rcdm_conv3x3 with upsample = 2): f16 [4][cout][4 * cin]."""
        w = self.f32(key)
        cout, cin = w.shape[0], w.shape[1]
        dst = torch.empty(4, cout, 4 * cin, dtype=torch.float16, device=self.device)
        hip.pack_conv3x3_up2(w.data_ptr(), cout, cin, dst.data_ptr())
        self._tmp.append(w)
        return dst

    def geglu(self, wkey, bkey):
        w, b = self.f32(wkey), self.f32(bkey)
        n_out, K = w.shape
        wd = torch.empty(n_out, K, dtype=torch.float16, device=self.device)
        bd = torch.empty(n_out, dtype=torch.float32, device=self.device)
        hip.pack_geglu_rows(w.data_ptr(), b.data_ptr(), n_out, K, wd.data_ptr(), bd.data_ptr())
        self._tmp += [w, b]
        return wd, bd

    def lnx_mat(self, keys, gamma, beta, bias=None, pe=None):
        """A LayerNorm folded into the stacked [n_i][K] matrices behind it (rcdm_gemm_lnx consumer operands):
        W = f16(W diag(gamma)), S[n] = sum_c W[n][c] (of the ROUNDED matrix: what the MFMA sums), b = bias + W beta;
        pe [F][K] (motion modules): tab[f] = b + W pe_f, the per-frame row table."""
        if not SW.LNX:
            return None
        w = torch.cat([self.f32(k).reshape(self.sd[k].shape[0], -1) for k in keys], dim=0).contiguous()
        wg = (w * gamma[None, :]).contiguous()
        dst = torch.empty(wg.shape, dtype=torch.float16, device=self.device)
        hip.pack_f16(wg.data_ptr(), dst.data_ptr(), wg.numel())   # (torch's current stream: ordered with the torch ops around it)
        S = dst.float().sum(dim=1).contiguous()
        b = (w * beta[None, :]).sum(dim=1)
        if bias is not None:
            b = b + bias
        tab = None
        if pe is not None:
            tab = torch.stack([b + (w * pe[f][None, :]).sum(dim=1) for f in range(pe.shape[0])]).contiguous()
        self._tmp += [w, wg]
        return _NS(W=dst, S=S, b=b.contiguous(), tab=tab)

    def lnx_geglu(self, wkey, bkey, gamma, beta):
        """The same for the GEGLU projection: folded, then packed like Packer.geglu (16 | 16 row interleave)."""
        if not SW.LNX:
            return None
        w, b = self.f32(wkey), self.f32(bkey)
        n_out, K = w.shape
        wg = (w * gamma[None, :]).contiguous()
        bb = (b + (w * beta[None, :]).sum(dim=1)).contiguous()
        wd = torch.empty(n_out, K, dtype=torch.float16, device=self.device)
        bd = torch.empty(n_out, dtype=torch.float32, device=self.device)
        hip.pack_geglu_rows(wg.data_ptr(), bb.data_ptr(), n_out, K, wd.data_ptr(), bd.data_ptr())
        S = wd.float().sum(dim=1).contiguous()
        self._tmp += [w, b, wg, bb]
        return _NS(W=wd, S=S, b=bd, tab=None)

    def ff_stream(self, w1key, b1key, w2key):
        """(weight stream, packed b1) for rcdm_ff_fused, or None when the library has no fused kernel for this width."""
        w1, b1, w2 = self.f32(w1key), self.f32(b1key), self.f32(w2key)
        Cc = w2.shape[0]
        if not (SW.FF_FUSE and w1.shape == (8 * Cc, Cc) and w2.shape == (Cc, 4 * Cc) and hip.ff_fused_supported(Cc)):
            return None
        ws = torch.empty(hip.ff_stream_bytes(Cc), dtype=torch.uint8, device=self.device)
        b1p = torch.empty(8 * Cc, dtype=torch.float32, device=self.device)
        hip.pack_ff_stream(w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), Cc, ws.data_ptr(), b1p.data_ptr())
        self._tmp += [w1, b1, w2]
        return ws, b1p

    def chain(self, wa_key, tail, wt_keys=(), ff_keys=None):
        """(weight stream, packed b1 | None) for rcdm_rowchain: stage-A matrix wa_key [C][C] (a Linear or 1x1 conv weight),
        then tail 1 / 3: the stacked [tail*C][C] matrices wt_keys, tail 0: the feed-forward (w1, b1, w2) keys, tail 2: the
        feed-forward keys and wt_keys = (the trailing [C][C] projection,) whose bias the launch takes separately.  None
        when the library has no chain kernel for this width or a tail 1 / 3 projection carries a bias."""
        if not SW.ROW_CHAIN or any(not self.has(k) for k in (wa_key, *wt_keys, *(ff_keys or ()))):
            return None
        Cc = self.sd[wa_key].shape[0]
        if not hip.rowchain_supported(Cc) or self.sd[wa_key].numel() != Cc * Cc:
            return None
        if tail != 2 and any(self.has(k.replace(".weight", ".bias")) for k in wt_keys):
            return None
        wa = self.f32(wa_key).reshape(Cc, Cc).contiguous()
        ws = torch.empty(hip.rowchain_stream_bytes(Cc, tail), dtype=torch.uint8, device=self.device)
        if tail == 2:
            w1, b1, w2 = (self.f32(k) for k in ff_keys)
            wz = self.f32(wt_keys[0])
            if w1.shape != (8 * Cc, Cc) or w2.shape != (Cc, 4 * Cc) or wz.numel() != Cc * Cc:
                return None
            wz = wz.reshape(Cc, Cc).contiguous()
            b1p = torch.empty(8 * Cc, dtype=torch.float32, device=self.device)
            hip.pack_rowchain(wa.data_ptr(), Cc, 2, wz.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), ws.data_ptr(),
                              b1p.data_ptr())
            self._tmp += [wa, wz, w1, b1, w2]
            return ws, b1p
        if tail:
            wt = torch.cat([self.f32(k) for k in wt_keys], dim=0).contiguous()
            if tuple(wt.shape) != (tail * Cc, Cc):
                return None
            hip.pack_rowchain(wa.data_ptr(), Cc, tail, wt.data_ptr(), 0, 0, 0, ws.data_ptr(), 0)
            self._tmp += [wa, wt]
            return ws, None
        w1, b1, w2 = (self.f32(k) for k in ff_keys)
        if w1.shape != (8 * Cc, Cc) or w2.shape != (Cc, 4 * Cc):
            return None
        b1p = torch.empty(8 * Cc, dtype=torch.float32, device=self.device)
        hip.pack_rowchain(wa.data_ptr(), Cc, 0, 0, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), ws.data_ptr(), b1p.data_ptr())
        self._tmp += [wa, w1, b1, w2]
        return ws, b1p

    def done(self):
        torch.cuda.synchronize(self.device)
        self._tmp.clear()


def pack_resnet(pk, p, wino=(False, False)):
    """wino = (conv1, conv2): which of the block's two 3x3 convolutions take the Winograd form (w.wino1 / w.wino2 instead of
    w.conv1 / w.conv2; with w.wino2 the 1x1 conv_shortcut stays a plain f16 matrix, w.shortcut, and rides as four parity GEMMs
    of conv2's batched launch)."""
    w = _NS(cin=pk.sd[p + "conv1.weight"].shape[1], cout=pk.sd[p + "conv1.weight"].shape[0])
    w.g1, w.b1 = pk.vec(p + "norm1.weight"), pk.vec(p + "norm1.bias")
    w.g2, w.b2 = pk.vec(p + "norm2.weight"), pk.vec(p + "norm2.bias")
    u1, u2 = (bool(wino), bool(wino)) if isinstance(wino, bool) else wino
    w.wino1 = w.wino2 = None
    if u1:
        w.wino1, w.cb1, w.conv1 = pk.conv3x3_wino(p + "conv1.weight"), pk.vec(p + "conv1.bias"), None
    if u2:
        w.wino2, w.cb2 = pk.conv3x3_wino(p + "conv2.weight"), pk.vec(p + "conv2.bias")
        w.conv2 = w.shortcut = w.conv2sc = None
        if pk.has(p + "conv_shortcut.weight"):
            w.shortcut, w.sb = pk.mat_f16(p + "conv_shortcut.weight"), pk.vec(p + "conv_shortcut.bias")
            w.cb2sc = (w.cb2 + w.sb).contiguous()
    if not u1:
        w.conv1, w.cb1 = pk.conv3x3(p + "conv1.weight"), pk.vec(p + "conv1.bias")
    if u2:
        return w
    w.conv2, w.cb2 = pk.conv3x3(p + "conv2.weight"), pk.vec(p + "conv2.bias")
    w.shortcut = w.conv2sc = None
    if pk.has(p + "conv_shortcut.weight"):
        w.shortcut, w.sb = pk.mat_f16(p + "conv_shortcut.weight"), pk.vec(p + "conv_shortcut.bias")
        if SW.SC_FOLD and w.cin % 64 == 0 and w.cout % 64 == 0:
            # conv2(h) + conv_shortcut(x) as ONE implicit GEMM over K = 9 cout + cin (rcdm_conv3x3_add1x1): the 1x1 weight's
            # columns behind the nine taps' (a copy of the already rounded halfs), the two biases summed in fp32
            w.conv2sc = torch.cat([w.conv2, w.shortcut], dim=1).contiguous()
            w.cb2sc = (w.cb2 + w.sb).contiguous()
            w.conv2 = w.shortcut = None   # (not read again: no second copy of the block's largest matrix)
    return w


def pack_attention(pk, a, fused_self):
    """CrossAttention parameters (attention.py:31-91): fused [q;k;v] for self-attention, q + [k;v] for cross."""
    w = _NS()
    has_b = pk.has(a + "to_q.bias")
    if fused_self:
        w.qkv = pk.mat_f16(a + "to_q.weight", a + "to_k.weight", a + "to_v.weight")
        w.qkv_b = torch.cat([pk.vec(a + f"to_{n}.bias") for n in "qkv"]).contiguous() if has_b else None
    else:
        w.q = pk.mat_f16(a + "to_q.weight")
        w.q_b = pk.vec(a + "to_q.bias") if has_b else None
        w.kv = pk.mat_f16(a + "to_k.weight", a + "to_v.weight")
        w.kv_b = torch.cat([pk.vec(a + "to_k.bias"), pk.vec(a + "to_v.bias")]).contiguous() if has_b else None
    w.o, w.o_b = pk.mat_f16(a + "to_out.0.weight"), pk.vec(a + "to_out.0.bias")
    return w


MSUB_SCORE_LIMIT = 2.0 ** 15   # documented range of the matrix-pipe-softmax flash kernel (include/rcdm.h)


def attn_score_bound(pk, a, ln, heads):
    """Data-independent upper bound of |scale * log2(e) * q.k| over every input, for a self-attention behind a LayerNorm
    (attention.py:482-493): ||LayerNorm(x) before gamma|| <= sqrt(C), so per head |q| <= sqrt(C) ||W_q,h diag(gamma)||_F +
    ||W_q,h beta + b_q,h|| (Frobenius >= spectral norm), likewise |k|, and |q.k| <= |q| |k|.  inf when there is no
    LayerNorm in front (nothing bounds the rows).  Evaluated once per block at pack time, in fp64 ON THE HOST (numpy: two
    matrix-vector products per site — no device kernel, no vendor library in any trace)."""
    if ln is None or not pk.has(a + "to_q.weight"):
        return float("inf")
    import numpy as np
    gamma = ln[0].detach().cpu().numpy().astype(np.float64)
    beta = ln[1].detach().cpu().numpy().astype(np.float64)
    C = gamma.size
    out = []
    for n in "qk":
        W = pk.sd[a + f"to_{n}.weight"].detach().float().cpu().numpy().astype(np.float64)
        bias = pk.sd[a + f"to_{n}.bias"].detach().float().cpu().numpy().astype(np.float64) if pk.has(a + f"to_{n}.bias") else 0.0
        d = W.shape[0] // heads
        Wh = (W * gamma[None, :]).reshape(heads, d, C)
        off = (W @ beta + bias).reshape(heads, d)
        out.append(math.sqrt(C) * np.sqrt((Wh ** 2).sum(axis=(1, 2))) + np.sqrt((off ** 2).sum(axis=1)))
    d = pk.sd[a + "to_q.weight"].shape[0] // heads
    return float((out[0] * out[1]).max() * d ** -0.5 * 1.4426950408889634)


def pack_basic_block(pk, b, lnx=True):
    """BasicTransformerBlock parameters (attention.py:368-477); attn2 / norm2 are absent in the stage-1 prior's blocks.
    lnx=False: the deferred-LayerNorm operands (a second, gamma-folded f16 copy of q|k|v, attn2.to_q and the GEGLU
    projection) are not packed — for blocks whose plan takes the row-stationary chains and never reads them."""
    C = pk.sd[b + "norm1.weight"].shape[0]
    w = _NS(C=C, has_cross=pk.has(b + "attn2.to_q.weight"), pk=pk)
    w.ln = [(pk.vec(b + f"norm{i}.weight"), pk.vec(b + f"norm{i}.bias")) if pk.has(b + f"norm{i}.weight") else None
            for i in (1, 2, 3)]
    a1 = pack_attention(pk, b + "attn1.", True)
    w.qkv1, w.qkv1_b, w.o1, w.o1_b = a1.qkv, a1.qkv_b, a1.o, a1.o_b
    w.attn1_ln, w.attn1_key = w.ln[0], b + "attn1."   # emit_basic_block: score bound of the self-attention (heads known there)
    if w.has_cross:
        a2 = pack_attention(pk, b + "attn2.", False)
        w.q2, w.q2_b, w.kv2, w.kv2_b, w.o2, w.o2_b = a2.q, a2.q_b, a2.kv, a2.kv_b, a2.o, a2.o_b
        w.ctx_dim = pk.sd[b + "attn2.to_k.weight"].shape[1]
    w.geglu = pk.has(b + "ff.net.0.proj.weight") and pk.sd[b + "ff.net.0.proj.weight"].shape[0] == 8 * C
    w.ff_stream = None
    if w.geglu:
        w.ff1, w.ff1_b = pk.geglu(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias")
        w.ff_stream = pk.ff_stream(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight")
    else:
        w.ff1, w.ff1_b = pk.mat_f16(b + "ff.net.0.proj.weight"), pk.vec(b + "ff.net.0.proj.bias")
    w.ff2, w.ff2_b = pk.mat_f16(b + "ff.net.2.weight"), pk.vec(b + "ff.net.2.bias")
    # deferred LayerNorm operands (rcdm_gemm_lnx): norm1 -> [q;k;v], norm2 -> attn2.to_q, norm3 -> GEGLU projection
    qkv_keys = tuple(b + f"attn1.to_{n}.weight" for n in "qkv")
    w.lnx_qkv = pk.lnx_mat(qkv_keys, *w.ln[0], bias=w.qkv1_b) if (lnx and w.ln[0] is not None) else None
    w.lnx_q2 = pk.lnx_mat((b + "attn2.to_q.weight",), *w.ln[1], bias=w.q2_b) if (lnx and w.has_cross and w.ln[1] is not None) else None
    w.lnx_ff = pk.lnx_geglu(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", *w.ln[2]) if (lnx and w.geglu and w.ln[2] is not None) else None
    # row-stationary chains: attn1.to_out + res -> norm2 -> attn2.to_q, and attn2.to_out + res -> norm3 -> ff -> + res
    w.ch_in_qkv = w.ch_o1_q = w.ch_o2_ff = None
    if w.has_cross and w.geglu:
        w.ch_o1_q = pk.chain(b + "attn1.to_out.0.weight", 1, wt_keys=(b + "attn2.to_q.weight",))
        w.ch_o2_ff = pk.chain(b + "attn2.to_out.0.weight", 0,
                              ff_keys=(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight"))
    return w


def pack_transformer(pk, p, lnx=True, ffz=True):
    """lnx / ffz False: the deferred-LayerNorm operands / the composed [W_po W_ff2 | W_po] matrix are left out (plans at or
    above the chain kernels' row count never read them: UNetProgram passes what its geometry needs)."""
    w = pack_basic_block(pk, p + "transformer_blocks.0.", lnx=lnx)
    b = p + "transformer_blocks.0.attn1."
    w.ch_in_qkv = pk.chain(p + "proj_in.weight", 3, wt_keys=(b + "to_q.weight", b + "to_k.weight", b + "to_v.weight"))
    w.gn_g, w.gn_b = pk.vec(p + "norm.weight"), pk.vec(p + "norm.bias")
    w.proj_in, w.proj_in_b = pk.mat_f16(p + "proj_in.weight"), pk.vec(p + "proj_in.bias")
    w.proj_out, w.proj_out_b = pk.mat_f16(p + "proj_out.weight"), pk.vec(p + "proj_out.bias")
    w.ffz = None
    if w.geglu and ffz:
        t = p + "transformer_blocks.0."
        w.ffz = pk.ffz(t + "ff.net.2.weight", t + "ff.net.2.bias", p + "proj_out.weight", p + "proj_out.bias")
    # ... and the block's last chain with proj_out + the transformer's residual behind the feed-forward
    w.ch_o2_ffz = None
    if w.ch_o2_ff is not None:
        t = p + "transformer_blocks.0."
        w.ch_o2_ffz = pk.chain(t + "attn2.to_out.0.weight", 2, wt_keys=(p + "proj_out.weight",),
                               ff_keys=(t + "ff.net.0.proj.weight", t + "ff.net.0.proj.bias", t + "ff.net.2.weight"))
    return w


def pack_motion(pk, p, n_attn, lnx=True):
    """lnx=False: neither the deferred-LayerNorm operands nor the composed proj_out matrix (see pack_transformer)."""
    p = p + "temporal_transformer."
    C = pk.sd[p + "norm.weight"].shape[0]
    b = p + "transformer_blocks.0."
    w = _NS(C=C)
    w.gn_g, w.gn_b = pk.vec(p + "norm.weight"), pk.vec(p + "norm.bias")
    if pk.has(p + "prior_norm.weight"):  # LayerNorm used instead of the GroupNorm when prior_state (stage 1)
        w.prior_g, w.prior_b = pk.vec(p + "prior_norm.weight"), pk.vec(p + "prior_norm.bias")
    w.proj_in, w.proj_in_b = pk.mat_f16(p + "proj_in.weight"), pk.vec(p + "proj_in.bias")
    w.proj_out, w.proj_out_b = pk.mat_f16(p + "proj_out.weight"), pk.vec(p + "proj_out.bias")
    w.attn = []
    for i in range(n_attn):
        a = b + f"attention_blocks.{i}."
        pe = pk.f32(a + "pos_encoder.pe")[0].contiguous() if pk.has(a + "pos_encoder.pe") else None
        ln_g, ln_b = pk.vec(b + f"norms.{i}.weight"), pk.vec(b + f"norms.{i}.bias")
        w.attn.append(_NS(
            ln_g=ln_g, ln_b=ln_b, pe=pe,
            qkv=pk.mat_f16(a + "to_q.weight", a + "to_k.weight", a + "to_v.weight"),
            lnx=pk.lnx_mat((a + "to_q.weight", a + "to_k.weight", a + "to_v.weight"), ln_g, ln_b, pe=pe) if lnx else None,
            o=pk.mat_f16(a + "to_out.0.weight"), o_b=pk.vec(a + "to_out.0.bias")))
    w.ff_ln = (pk.vec(b + "ff_norm.weight"), pk.vec(b + "ff_norm.bias"))
    w.lnx_ff = pk.lnx_geglu(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", *w.ff_ln) if lnx else None
    w.ff1, w.ff1_b = pk.geglu(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias")
    w.ff2, w.ff2_b = pk.mat_f16(b + "ff.net.2.weight"), pk.vec(b + "ff.net.2.bias")
    w.ff_stream = pk.ff_stream(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight")
    w.ffz = pk.ffz(b + "ff.net.2.weight", b + "ff.net.2.bias", p + "proj_out.weight", p + "proj_out.bias") if lnx else None
    # row-stationary chains: proj_in -> norms[0] + pe -> qkv;  to_out + res -> norms[1] + pe -> qkv;  to_out + res ->
    # ff_norm -> ff -> + res
    w.chains = w.chain_ffz = None
    if n_attn == 2:
        a0, a1 = b + "attention_blocks.0.", b + "attention_blocks.1."
        qkv = lambda a: (a + "to_q.weight", a + "to_k.weight", a + "to_v.weight")
        ch = [pk.chain(p + "proj_in.weight", 3, wt_keys=qkv(a0)), pk.chain(a0 + "to_out.0.weight", 3, wt_keys=qkv(a1)),
              pk.chain(a1 + "to_out.0.weight", 0,
                       ff_keys=(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight"))]
        if all(c is not None for c in ch) and all(at.pe is not None for at in w.attn):
            w.chains = ch
            # proj_out + the module's residual behind the feed-forward (zero-initialised proj_out included)
            w.chain_ffz = pk.chain(a1 + "to_out.0.weight", 2, wt_keys=(p + "proj_out.weight",),
                                   ff_keys=(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight"))
    return w
