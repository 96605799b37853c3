"""Block emitters: the reference's module graph as launch sequences over channels-last f16 rows — ResnetBlock3D
(src/models/resnet.py:182-212), Transformer3DModel + BasicTransformerBlock (src/models/attention.py:318-365,479-526),
VanillaTemporalModule ... TemporalTransformerBlock (src/models/motion_module.py:87-93,147-182,234-246).  Each picks,
per geometry, between the row-stationary chains of the 64x64 level and the tile-parallel launches with deferred
LayerNorms / composed proj_out below it."""
import torch

from . import hip
from . import switches as SW
from .emit_ops import (XATTN_MAX_KEYS, assert_no_pending_gn, conv3x3_wino_ok, emit_conv3x3, emit_conv3x3_wino, emit_flash_attn, emit_gemm,
                       emit_groupnorm, emit_groupnorm_stats, emit_layernorm, emit_temporal_attn, emit_xattn, emit_xattn_pack, gemm_lnx_ok)
from .packer import MSUB_SCORE_LIMIT, attn_score_bound
from .plan import Rows, _NS


def emit_resnet(plan, w, x, geo, temb, out, eps=1e-5, groups=32, out_scale=1.0, dup_rows=0, out_gn=None):
    """ResnetBlock3D.forward (src/models/resnet.py:182-212).  temb = (tensor [b][ldt] fp32, col offset, ldt)
    = this block's slice of the batched time_emb_proj(silu(emb)) table.  x may be a concat view.
    out_gn = (samples, rows_per_sample, groups) of the GroupNorm that reads `out` as the very next op (the norm in front of
    the transformer / motion module / ResNet block behind this one), or None: a split-K conv2 then leaves its statistics."""
    g = geo
    u1, u2 = getattr(w, "wino1", None) is not None, getattr(w, "wino2", None) is not None
    rv = (temb[0], temb[1], temb[2], g.f * g.hw)
    cross = (g.b, g.f * g.hw, groups)
    assert not (u1 or u2) or (not dup_rows and x.C == w.cin)
    h1 = plan.rows("res_h1", g.M, w.cout)
    # ---- norm1 + conv1.  Winograd form (wino_level): the norm is a statistics-only launch, its apply + SiLU rides in the input
    # transform; its output transform leaves norm2's per-tile statistics when conv2 takes that form too
    res = x
    if u1:
        gn1 = emit_groupnorm_stats(plan, x, g.b, g.f * g.hw, w.g1, w.b1, eps, groups)   # (first: x may carry a producer's statistics)
        emit_conv3x3_wino(plan, x, g.n_img, g.H, g.W, w.wino1, w.cin, w.cout, h1, bias=w.cb1, rowvec=rv, gn=gn1,
                          gn_out=cross if u2 else None)
    else:
        a1 = plan.rows("norm", g.M, x.C)
        emit_groupnorm(plan, x, g.b, g.f * g.hw, w.g1, w.b1, eps, True, a1, groups)   # (first: x may carry a producer's statistics)
        if not u2 and w.shortcut is not None:
            res = plan.rows("res_sc", g.M, w.cout)
            emit_gemm(plan, x, w.shortcut, w.cout, w.cin, res, bias=w.sb)
        emit_conv3x3(plan, a1, g.n_img, g.H, g.W, w.conv1, w.cin, w.cout, h1, bias=w.cb1, rowvec=rv, gn=cross)
    # ---- norm2 + conv2 (+ conv_shortcut(x) | + x)
    if u2:
        gn2 = emit_groupnorm_stats(plan, h1, g.b, g.f * g.hw, w.g2, w.b2, eps, groups)
        if w.shortcut is not None:   # conv_shortcut(x) as four parity GEMMs of conv2's batched launch
            emit_conv3x3_wino(plan, h1, g.n_img, g.H, g.W, w.wino2, w.cout, w.cout, out, bias=w.cb2sc, scale=out_scale, gn=gn2,
                              x2=x, W2=w.shortcut, gn_out=out_gn, gn_out_apply=True)
        else:
            emit_conv3x3_wino(plan, h1, g.n_img, g.H, g.W, w.wino2, w.cout, w.cout, out, bias=w.cb2, residual=x, scale=out_scale,
                              gn=gn2, gn_out=out_gn, gn_out_apply=True)
        return
    fold_sc = w.conv2sc is not None
    assert not fold_sc or x.C == w.cin
    a2 = plan.rows("norm", g.M, w.cout)
    emit_groupnorm(plan, h1, g.b, g.f * g.hw, w.g2, w.b2, eps, True, a2, groups)
    if fold_sc:   # (resnet.py:205-212 with a conv_shortcut: its 1x1 convolution of x rides in conv2's accumulators)
        emit_conv3x3(plan, a2, g.n_img, g.H, g.W, w.conv2sc, w.cout, w.cout, out, bias=w.cb2sc, scale=out_scale,
                     dup_rows=dup_rows, gn=out_gn, x2=x)
        return
    emit_conv3x3(plan, a2, g.n_img, g.H, g.W, w.conv2, w.cout, w.cout, out, bias=w.cb2, residual=res, scale=out_scale,
                 dup_rows=dup_rows, gn=out_gn)


WINO_MIN_CHANNELS = SW.WINO_MIN_C   # k-loops of >= 10 steps per position GEMM; narrower convs keep the nine-tap form at every size
WINO_SHORTCUT_MIN_SIDE = 16   # below: a conv2 that carries a conv_shortcut keeps the nine-tap form (measured 49 against 47 us at 8x8)


def wino_level(geo, cin, cout):
    """(conv1, conv2): which 3x3 convolutions of a ResNet block of this geometry take the Winograd form (switches.wino_side: the
    levels it is used at — sides <= 32, where the nine-tap implicit GEMM is a split-K latency chain, not a stream)."""
    if not SW.wino_side(geo.H, geo.W):
        return (False, False)
    u1 = min(cin, cout) >= WINO_MIN_CHANNELS and conv3x3_wino_ok(geo.n_img, geo.H, geo.W, cin, cout)
    cin2 = cin if cin != cout else 0
    u2 = (cout >= WINO_MIN_CHANNELS and (not cin2 or geo.H >= WINO_SHORTCUT_MIN_SIDE) and
          conv3x3_wino_ok(geo.n_img, geo.H, geo.W, cout, cout, cin2))
    return (u1, u2)


# a chain launch (rcdm_rowchain / rcdm_ff_fused, rowff.hip) is one block of 160 rows per CU: below ~3/4 of a chip's worth of
# rows (the 256x256 configuration has 10240 token rows at this width = 64 blocks) the separate tile-parallel launches are faster
class _ChainMinRows:
    """Rows from which the row-stationary chain launches are used: 3/4 of a chip's worth of 160-row blocks, one block per
    CU — read from the device (hipDeviceAttributeMultiprocessorCount through torch) the first time a plan compares against
    it, 256 CUs (MI355X: 160 * 192 = 30720 rows) when no device is visible; switches.CHAIN_MIN_ROWS overrides."""

    def __init__(self):
        self._v = None

    def value(self):
        if self._v is None:
            env = SW.CHAIN_MIN_ROWS
            if env:
                self._v = int(env)
            else:
                cus = 256
                if torch.cuda.is_available():
                    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count or 256
                self._v = 160 * (3 * cus // 4)
        return self._v

    def __le__(self, other):   # other >= CHAIN_MIN_ROWS
        return self.value() <= other

    def __gt__(self, other):   # other < CHAIN_MIN_ROWS
        return self.value() > other

    def __int__(self):
        return self.value()

    def __repr__(self):
        return str(self.value())


CHAIN_MIN_ROWS = _ChainMinRows()


def emit_rowchain(plan, a_in, res, tok, a_bias, ln, pe, stream, tail, out, rows_per_frame=1, frames=1, b2=None, gn=None,
                  z=None):
    """tok = a_in W_a^T + a_bias (+ res);  y = LayerNorm(tok) (+ pe);  tail 1 / 3: out = y W_t^T;  tail 0: out = tok + FF(y).
    ln = (gamma, beta); stream = Packer.chain(...); gn = emit_groupnorm_stats(...): a_in is the RAW input of that
    GroupNorm and the kernel applies it while loading its rows (res must be None).  tail 2, z = (z_res rows, z_bias):
    out = z_res + (tok + FF(y)) W_z^T + z_bias, the feed-forward's own output rows are not stored."""
    ws, b1p = stream
    M, C = a_in.M, a_in.C
    assert_no_pending_gn(plan, "a row-chain launch")
    d = hip.RowChainDesc(M, C, a_in.ld, res.ld if res is not None else 0, tok.ld, out.ld, tail, rows_per_frame, frames, 1e-5,
                         gn[3] if gn else 0, gn[4] if gn else 0, z[0].ld if z else 0)

    def op():
        hip.rowchain(d, a_in.ptr, res.ptr if res is not None else 0, tok.ptr, a_bias.data_ptr(), ln[0].data_ptr(),
                     ln[1].data_ptr(), pe.data_ptr() if pe is not None else 0, ws.data_ptr(),
                     b1p.data_ptr() if b1p is not None else 0, b2.data_ptr() if b2 is not None else 0, out.ptr,
                     gn_stat=gn[0].ptr if gn else 0, gn_g=gn[1].data_ptr() if gn else 0, gn_b=gn[2].data_ptr() if gn else 0,
                     z_res=z[0].ptr if z else 0, z_bias=z[1].data_ptr() if z else 0)
    plan.add(op, f"rowchain M={M} C={C} tail={tail} res={int(res is not None)} pe={int(pe is not None)} gn={int(gn is not None)}")
    plan.keep += [a_bias, ln[0], ln[1], pe, ws, b1p, b2, z[1] if z else None]
    plan.n_launch += 1


def emit_ff(plan, tok, ln_g, ln_b, ff1, ff1_b, ff2, ff2_b, a, M, C, stream=None, tok_stat=None, lnx=None, z=None, out_gn=None):
    """x += FeedForward_geglu(LayerNorm(x))  (attention.py:514 / motion_module.py:243), in place on tok.
    stream: (fragment-major weight stream, packed b1) of Packer.ff_stream, or None for the unfused chain.
    tok_stat / lnx: row statistics of tok from the GEMM that wrote it + Packer.lnx_geglu operands — the LayerNorm then
    rides in the GEGLU projection's epilogue (deferred LayerNorm).
    z = (ffz, cat, x, out): tok is the last C columns of cat [M][5C] (ffz_rows); the feed-forward's own result is not
    stored — out = x + proj_out(tok + FF(..)) comes out of ONE K = 5C GEMM over [h | tok] (Packer.ffz)."""
    if z is not None:
        ffz, cat, x, out = z
        gg = cat.cols(0, 4 * C)
        assert tok.ptr_key() == cat.cols(4 * C, C).ptr_key() and M < CHAIN_MIN_ROWS
        if tok_stat is not None and lnx is not None and gemm_lnx_ok(M, 8 * C, C, tok.ld, gg.ld, geglu=True):
            emit_gemm(plan, tok, lnx.W, 8 * C, C, gg, bias=lnx.b, geglu=True, lnx=(tok_stat, lnx.S))
        else:
            emit_layernorm(plan, tok, ln_g, ln_b, a)
            emit_gemm(plan, a, ff1, 8 * C, C, gg, bias=ff1_b, geglu=True)
        emit_gemm(plan, cat, ffz.W, C, 5 * C, out, bias=ffz.b, residual=x, gn=out_gn)   # (out_gn: the norm that reads `out` next)
        return
    if stream is not None and M >= CHAIN_MIN_ROWS:
        ws, b1p = stream
        d = hip.FFDesc(M, C, tok.ld, tok.ld, 1e-5)

        def op():
            hip.ff_fused(d, tok.ptr, ln_g.data_ptr(), ln_b.data_ptr(), ws.data_ptr(), b1p.data_ptr(), ff2_b.data_ptr(), tok.ptr)
        plan.add(op, f"ff_fused M={M} C={C}")
        plan.keep += [ln_g, ln_b, ws, b1p, ff2_b]
        plan.n_launch += 1
        return
    gg = plan.rows("geglu", M, 4 * C)
    if tok_stat is not None and lnx is not None and gemm_lnx_ok(M, 8 * C, C, tok.ld, gg.ld, geglu=True):
        emit_gemm(plan, tok, lnx.W, 8 * C, C, gg, bias=lnx.b, geglu=True, lnx=(tok_stat, lnx.S))
    else:
        emit_layernorm(plan, tok, ln_g, ln_b, a)
        emit_gemm(plan, a, ff1, 8 * C, C, gg, bias=ff1_b, geglu=True)
    emit_gemm(plan, gg, ff2, C, 4 * C, tok, bias=ff2_b, residual=tok)


def emit_basic_block(plan, w, tok, n_seq, Lq, heads, a, ctx_kv=None, L=0, shared_half=False, ctx_img=None, pre=None, post=None,
                     tok_stat=None, z=None, out_gn=None, rank1=None):
    """BasicTransformerBlock.forward (src/models/attention.py:479-526) in place on tok [n_seq*Lq][C]:
    h += attn1(LN1(h)); h += attn2(LN2(h), ctx); h += FF(LN3(h)).  ctx_kv: Rows [n_seq*L][2C] = [K | V] of the context.
    shared_half: the two halves of tok (the CFG halves of a denoising step) hold IDENTICAL rows on entry — everything up
    to the query projection of the cross-attention is then computed on the first half only and stored to both.
    tok_stat: row statistics of tok from the GEMM that wrote it (emit_gemm(stat=True)); with them, and below the chain
    kernels' row count, the three LayerNorms are deferred into the epilogues of the GEMMs behind them (rcdm_gemm_lnx).
    rank1 (emit_rank1_ctx): the context rows of every sequence OUTSIDE rank1.runs are all equal (the unseen
    frames' semantic_stack output, stage2_batchtest_rcdms_model.py:117-132) — the softmax over equal scores is uniform, so
    attn2 (attention.py:139-144,170-199) of such a sequence is its V row for EVERY query, and to_out(V row) + bias is one
    row r per sequence.  At the chain kernels' row count the cross-attention runs on the full-rank runs only and the other
    rows of its output buffer hold the V row already (written once per context); below it norm2 / to_q / the attention /
    to_out run on the full-rank runs only and r rides in attn1.to_out's epilogue as a per-sequence row vector."""
    C, M = w.C, n_seq * Lq
    d_head = C // heads
    ns, Ms, dup = n_seq, M, 0
    if shared_half:
        assert w.has_cross and n_seq % 2 == 0
        ns, Ms, dup = n_seq // 2, M // 2, M // 2
    ao = plan.rows("attn_out", M, C)
    # self-attention over the Lq tokens of each sequence.  pre = (rows, bias): tok = rows proj_in^T + bias has NOT been
    # emitted yet and rides in the chain launch with norm1 and the q | k | v projection
    qkv = plan.rows("qkv", M, 3 * C).rows(0, Ms)
    big = M >= CHAIN_MIN_ROWS
    if pre is not None:
        emit_rowchain(plan, pre[0], None, tok, pre[1], w.ln[0], None, w.ch_in_qkv, 3, qkv, gn=pre[2])
    elif tok_stat is not None and w.lnx_qkv is not None and gemm_lnx_ok(Ms, 3 * C, C, tok.ld, qkv.ld):
        emit_gemm(plan, tok.rows(0, Ms), w.lnx_qkv.W, 3 * C, C, qkv, bias=w.lnx_qkv.b, lnx=(tok_stat, w.lnx_qkv.S))
    else:
        emit_layernorm(plan, tok.rows(0, Ms), w.ln[0][0], w.ln[0][1], a.rows(0, Ms))
        emit_gemm(plan, a.rows(0, Ms), w.qkv1, 3 * C, C, qkv, bias=w.qkv1_b)
    if not hasattr(w, "score_bound"):
        w.score_bound = attn_score_bound(w.pk, w.attn1_key, w.attn1_ln, heads)
    emit_flash_attn(plan, qkv.cols(0, C), qkv.cols(C, C), qkv.cols(2 * C, C), ns, heads, Lq, Lq, d_head, ao.rows(0, Ms),
                    wide=w.score_bound >= MSUB_SCORE_LIMIT, bound=w.score_bound)
    chain_q = w.has_cross and w.ch_o1_q is not None and not shared_half and big
    # (below CHAIN_MIN_ROWS only: with more rows the N = C producers run on the ping-pong kernel, which has no statistics
    # epilogue — the b = 8 configuration measured 59.5 ms per step without and 60.0 with the deferred form at 40960 rows)
    want_ff = SW.LNX and not big and w.geglu and w.lnx_ff is not None          # statistics for norm3 -> GEGLU
    # statistics for norm2 -> attn2.to_q (also for the shared half of a big batch, which runs the separate launches on Ms rows)
    want_q2 = SW.LNX and (not big or (shared_half and Ms < CHAIN_MIN_ROWS)) and w.has_cross and w.lnx_q2 is not None
    st = None
    if rank1 is not None:
        assert w.has_cross and ctx_img is None and (big or not shared_half)
    if rank1 is not None and not big:
        # the rank-1 sequences: tok += r_i here (their attn2 does not depend on the query), statistics for norm2 (full-rank
        # rows) AND, for the rows no later launch touches, norm3
        st = emit_gemm(plan, ao, w.o1, C, C, tok, bias=w.o1_b, residual=tok, rowvec=(rank1.rtab, 0, C, Lq),
                       stat=want_q2 or want_ff)
        qc = plan.rows("qkv", M, C)
        for (i0, i1), img in zip(rank1.runs, rank1.imgs):
            r0, n = i0 * Lq, (i1 - i0) * Lq
            if st is not None and w.lnx_q2 is not None and gemm_lnx_ok(n, C, C, tok.ld, qc.ld):
                emit_gemm(plan, tok.rows(r0, n), w.lnx_q2.W, C, C, qc.rows(r0, n), bias=w.lnx_q2.b, lnx=(st, w.lnx_q2.S),
                          stat_row0=r0)
            else:
                emit_layernorm(plan, tok.rows(r0, n), w.ln[1][0], w.ln[1][1], a.rows(r0, n))
                emit_gemm(plan, a.rows(r0, n), w.q2, C, C, qc.rows(r0, n), bias=w.q2_b)
            emit_xattn(plan, qc.rows(r0, n), img, i1 - i0, heads, Lq, L, d_head, ao.rows(r0, n))
            if st is not None:   # the rewritten rows' statistics go into the same buffer (None: no tile with its slot count)
                st = emit_gemm(plan, ao.rows(r0, n), w.o2, C, C, tok.rows(r0, n), bias=w.o2_b, residual=tok.rows(r0, n),
                               stat_into=st, stat_row0=r0)
            else:
                emit_gemm(plan, ao.rows(r0, n), w.o2, C, C, tok.rows(r0, n), bias=w.o2_b, residual=tok.rows(r0, n))
        # (st None: some run had no tile with the buffer's slot count — norm3 then takes the stand-alone launch in emit_ff)
        assert w.geglu
        emit_ff(plan, tok, w.ln[2][0], w.ln[2][1], w.ff1, w.ff1_b, w.ff2, w.ff2_b, a, M, C, stream=w.ff_stream,
                tok_stat=st, lnx=w.lnx_ff, z=z, out_gn=out_gn)
        return
    if not chain_q:
        st = emit_gemm(plan, ao.rows(0, Ms), w.o1, C, C, tok.rows(0, Ms), bias=w.o1_b, residual=tok.rows(0, Ms), dup_rows=dup,
                       stat=want_q2 or (want_ff and not w.has_cross))
    if w.has_cross:
        # cross-attention over the L context rows of that sequence
        qc = plan.rows("qkv", M, C)
        if chain_q:   # attn1.to_out + residual -> norm2 -> attn2.to_q in one launch
            emit_rowchain(plan, ao, tok, tok, w.o1_b, w.ln[1], None, w.ch_o1_q, 1, qc)
        elif st is not None and w.lnx_q2 is not None and gemm_lnx_ok(Ms, C, C, tok.ld, qc.ld, dup_rows=dup):
            emit_gemm(plan, tok.rows(0, Ms), w.lnx_q2.W, C, C, qc.rows(0, Ms), bias=w.lnx_q2.b, dup_rows=dup,
                      lnx=(st, w.lnx_q2.S))
        else:
            emit_layernorm(plan, tok.rows(0, Ms), w.ln[1][0], w.ln[1][1], a.rows(0, Ms))
            emit_gemm(plan, a.rows(0, Ms), w.q2, C, C, qc.rows(0, Ms), bias=w.q2_b, dup_rows=dup)
        if rank1 is not None:     # (chain kernels' row count) the other rows of rank1.ao hold their sequence's V row already
            for (i0, i1), img in zip(rank1.runs, rank1.imgs):
                emit_xattn(plan, qc.rows(i0 * Lq, (i1 - i0) * Lq), img, i1 - i0, heads, Lq, L, d_head,
                           rank1.ao.rows(i0 * Lq, (i1 - i0) * Lq))
            ao = rank1.ao
        elif ctx_img is not None:   # short context: the per-context fragment image (emit_ctx_kv), scores in registers
            emit_xattn(plan, qc, ctx_img, n_seq, heads, Lq, L, d_head, ao)
        else:
            emit_flash_attn(plan, qc, ctx_kv.cols(0, C), ctx_kv.cols(C, C), n_seq, heads, Lq, L, d_head, ao)
        if post is not None and big:  # ... and the transformer's proj_out + residual behind it: post = (stream, x, bias, out)
            emit_rowchain(plan, ao, tok, tok, w.o2_b, w.ln[2], None, post[0], 2, post[3], b2=w.ff2_b, z=(post[1], post[2]))
            return
        if w.ch_o2_ff is not None and big:   # attn2.to_out + residual -> norm3 -> ff -> + residual in one launch
            emit_rowchain(plan, ao, tok, tok, w.o2_b, w.ln[2], None, w.ch_o2_ff, 0, tok, b2=w.ff2_b)
            return
        st = emit_gemm(plan, ao, w.o2, C, C, tok, bias=w.o2_b, residual=tok, stat=want_ff)
    if w.geglu:
        emit_ff(plan, tok, w.ln[2][0], w.ln[2][1], w.ff1, w.ff1_b, w.ff2, w.ff2_b, a, M, C, stream=w.ff_stream,
                tok_stat=st if st is not None and st.M >= M else None, lnx=w.lnx_ff, z=z, out_gn=out_gn)
    else:   # FeedForward("gelu"): Linear -> exact GELU -> Linear (stage-1 prior blocks)
        emit_layernorm(plan, tok, w.ln[2][0], w.ln[2][1], a)
        hid = plan.rows("geglu", M, 4 * C)
        emit_gemm(plan, a, w.ff1, 4 * C, C, hid, bias=w.ff1_b, gelu=True)
        emit_gemm(plan, hid, w.ff2, C, 4 * C, tok, bias=w.ff2_b, residual=tok)


def ffz_rows(plan, w, M, C, x, out):
    """(z, tok) for a transformer / motion module: below the chain kernels' row count, with a GEGLU feed-forward and
    Packer.ffz operands, the token rows are the last C columns of a [M][5C] buffer and z = (ffz, cat, x, out) tells emit_ff
    to fold proj_out (+ the module's residual x) into the feed-forward's second GEMM; else (None, plain token rows)."""
    if getattr(w, "ffz", None) is not None and M < CHAIN_MIN_ROWS:
        cat = plan.rows("ffcat", M, 5 * C)
        return (w.ffz, cat, x, out), cat.cols(4 * C, C)
    return None, plan.rows("tok", M, C)


def emit_transformer(plan, w, x, geo, ctx_kv, L, heads, out, groups=32, shared_half=False, ctx_img=None, out_gn=None,
                     rank1=None):
    """Transformer3DModel.forward + BasicTransformerBlock.forward (src/models/attention.py:318-365,479-526).
    ctx_kv: Rows [n_img*L][2C] = [K | V] projections of the context for this site (computed per context).
    shared_half: see emit_basic_block (x holds identical halves; both halves of `out` are still written in full)."""
    g, C = geo, w.C
    n_s, M_s = (g.n_img // 2, g.M // 2) if shared_half else (g.n_img, g.M)
    a = plan.rows("norm", g.M, C)
    z, tok = ffz_rows(plan, w, g.M, C, x, out)
    pre, tok_stat = None, None
    if w.ch_in_qkv is not None and not shared_half and g.M >= CHAIN_MIN_ROWS:
        if g.hw % 16 == 0 and g.hw >= 160:   # the norm's apply rides too: only its statistics are launched
            pre = (x, w.proj_in_b, emit_groupnorm_stats(plan, x, n_s, g.hw, w.gn_g, w.gn_b, 1e-6, groups))
        else:
            emit_groupnorm(plan, x, n_s, g.hw, w.gn_g, w.gn_b, 1e-6, False, a, groups)
            pre = (a, w.proj_in_b, None)       # proj_in rides with norm1 + qkv (emit_basic_block)
    else:
        emit_groupnorm(plan, x.rows(0, M_s), n_s, g.hw, w.gn_g, w.gn_b, 1e-6, False, a.rows(0, M_s), groups)
        tok_stat = emit_gemm(plan, a.rows(0, M_s), w.proj_in, C, C, tok.rows(0, M_s), bias=w.proj_in_b,
                             stat=SW.LNX and M_s < CHAIN_MIN_ROWS)   # (M_s: the shared half of a big batch qualifies too)
    post = (w.ch_o2_ffz, x, w.proj_out_b, out) if (getattr(w, "ch_o2_ffz", None) is not None and w.has_cross and
                                                   g.M >= CHAIN_MIN_ROWS) else None
    emit_basic_block(plan, w, tok, g.n_img, g.hw, heads, a, ctx_kv, L, shared_half=shared_half, ctx_img=ctx_img, pre=pre,
                     post=post, tok_stat=tok_stat, z=z, out_gn=out_gn, rank1=rank1)
    if post is None and z is None:
        emit_gemm(plan, tok, w.proj_out, C, C, out, bias=w.proj_out_b, residual=x, gn=out_gn)


def emit_ctx_kv(plan, w, ctx16, ctx_kv, n_seq=0, L=0, heads=0):
    """[K | V] = ctx [to_k; to_v]^T  (CrossAttention.forward attention.py:139-141) — context only.  With n_seq / L /
    heads given and L <= XATTN_MAX_KEYS, also the fragment image rcdm_xattn reads (returned; else None)."""
    emit_gemm(plan, ctx16, w.kv2, 2 * w.C, w.ctx_dim, ctx_kv, bias=getattr(w, "kv2_b", None))
    if heads and 0 < L <= XATTN_MAX_KEYS:
        return emit_xattn_pack(plan, ctx_kv.cols(0, w.C), ctx_kv.cols(w.C, w.C), n_seq, heads, L, w.C // heads)
    return None


def full_rank_runs(ctx):
    """Maximal runs [(i0, i1), ...] of images of ctx (n_img, L, D) whose L context rows are NOT all bit-identical; the
    images outside them are the rank-1 ones.  One comparison on the device + one small copy to the host; NaN
    rows compare unequal, i.e. count as full rank (the general path)."""
    full = (~(ctx == ctx[:, :1, :]).all(dim=2).all(dim=1)).tolist()
    runs, i = [], 0
    while i < len(full):
        if full[i]:
            j = i
            while j < len(full) and full[j]:
                j += 1
            runs.append((i, j))
            i = j
        else:
            i += 1
    return tuple(runs)


def emit_rank1_ctx(ctx_plan, plan, w, kv, geo, L, heads, runs, site):
    """Per-context part of the rank-1-context plan of ONE cross-attention site (run by UNetProgram.set_context, not per
    step): the fragment images of the full-rank runs, and for the images outside `runs` either — at the chain kernels' row
    count — their V row written to every row of the site's own cross-attention output buffer, or — below it — the table
    r[i] = to_out(V row of image i) + bias (zero rows for the full-rank images) that attn1.to_out's epilogue adds.
    Returns what emit_basic_block takes as rank1.  kv: Rows [n_img * L][2C] = [K | V] of the context (emit_ctx_kv)."""
    C, n_img, hw = w.C, geo.n_img, geo.hw
    d_head = C // heads
    r1 = _NS(runs=tuple(runs), imgs=[], rtab=None, ao=None)
    for i0, i1 in runs:
        sub = kv.rows(i0 * L, (i1 - i0) * L)
        r1.imgs.append(emit_xattn_pack(ctx_plan, sub.cols(0, C), sub.cols(C, C), i1 - i0, heads, L, d_head))
    in_run = [any(i0 <= i < i1 for i0, i1 in runs) for i in range(n_img)]
    low = torch.tensor([i for i in range(n_img) if not in_run[i]], dtype=torch.long, device=plan.device)   # the rank-1 images
    high = torch.tensor([i for i in range(n_img) if in_run[i]], dtype=torch.long, device=plan.device)
    plan.keep += [low, high]

    def kv_view():
        return kv.buf.t.view(torch.float16)[kv.off:kv.off + n_img * L * kv.ld].view(n_img, L, kv.ld)

    if geo.M >= CHAIN_MIN_ROWS:
        r1.ao = plan.rows(f"xattn_out{site}", geo.M, C, unique=True)

        def fill():   # (torch's current stream = the program's stream inside set_context)
            ao_t = r1.ao.buf.t.view(torch.float16)[:geo.M * C].view(n_img, hw, C)
            ao_t[low] = kv_view()[low, 0, C:2 * C][:, None, :].expand(-1, hw, -1)
        ctx_plan.add(fill, f"rank1_fill M={geo.M} C={C}")
    else:
        r16 = plan.rows(f"rank1_r16_{site}", n_img, C, unique=True)
        r1.rtab = torch.zeros(n_img, C, dtype=torch.float32, device=plan.device)
        plan.keep.append(r1.rtab)
        vrow = Rows(kv.buf, kv.off + C, n_img, C, L * kv.ld)    # row i = the V row of image i's first context token
        emit_gemm(ctx_plan, vrow, w.o2, C, C, r16, bias=w.o2_b)

        def table():
            t = r16.buf.t.view(torch.float16)[:n_img * C].view(n_img, C).float()
            t[high] = 0.0
            r1.rtab.copy_(t)
        ctx_plan.add(table, f"rank1_table n={n_img} C={C}")
    return r1


def emit_motion(plan, w, x, geo, heads, out, groups=32, prior_state=False, out_gn=None):
    """VanillaTemporalModule -> TemporalTransformer3DModel.forward -> TemporalTransformerBlock.forward
    (src/models/motion_module.py:87-93,147-182,234-246).  prior_state (stage-1 prior, :150-153,172-174): the rows are
    (b f) x n tokens (geo.hw = n), the leading norm is the LayerNorm `prior_norm` instead of the per-frame GroupNorm."""
    g, C = geo, w.C
    d_head = C // heads
    a = plan.rows("norm", g.M, C)
    chained = (w.chains is not None and not prior_state and g.M >= CHAIN_MIN_ROWS and
               hip.rowchain_config_supported(C, 3, g.f))   # pe table of g.f frames in the tail-3 chains
    gn = None
    if prior_state:
        emit_layernorm(plan, x, w.prior_g, w.prior_b, a)
    elif chained and g.hw % 16 == 0 and g.hw >= 160:
        gn = emit_groupnorm_stats(plan, x, g.n_img, g.hw, w.gn_g, w.gn_b, 1e-6, groups)
    else:
        emit_groupnorm(plan, x, g.n_img, g.hw, w.gn_g, w.gn_b, 1e-6, False, a, groups)
    z, tok = (None, plan.rows("tok", g.M, C)) if chained else ffz_rows(plan, w, g.M, C, x, out)
    if chained:
        # three chain launches + two temporal attentions + proj_out instead of twelve launches
        qkv = plan.rows("qkv", g.M, 3 * C)
        ao = plan.rows("attn_out", g.M, C)
        a0, a1 = w.attn
        emit_rowchain(plan, x if gn else a, None, tok, w.proj_in_b, (a0.ln_g, a0.ln_b), a0.pe, w.chains[0], 3, qkv, g.hw, g.f, gn=gn)
        emit_temporal_attn(plan, qkv, g.b, g.f, g.hw, heads, d_head, ao)
        emit_rowchain(plan, ao, tok, tok, a0.o_b, (a1.ln_g, a1.ln_b), a1.pe, w.chains[1], 3, qkv, g.hw, g.f)
        emit_temporal_attn(plan, qkv, g.b, g.f, g.hw, heads, d_head, ao)
        if w.chain_ffz is not None:
            emit_rowchain(plan, ao, tok, tok, a1.o_b, w.ff_ln, None, w.chain_ffz, 2, out, b2=w.ff2_b, z=(x, w.proj_out_b))
            return
        emit_rowchain(plan, ao, tok, tok, a1.o_b, w.ff_ln, None, w.chains[2], 0, tok, b2=w.ff2_b)
        emit_gemm(plan, tok, w.proj_out, C, C, out, bias=w.proj_out_b, residual=x)
        return
    # separate launches; the LayerNorms (+ positional encoding) deferred into the q | k | v / GEGLU epilogues (rcdm_gemm_lnx)
    want_stat = SW.LNX and g.M < CHAIN_MIN_ROWS
    st = emit_gemm(plan, a, w.proj_in, C, C, tok, bias=w.proj_in_b, stat=want_stat)
    for at in w.attn:
        qkv = plan.rows("qkv", g.M, 3 * C)
        lx = at.lnx
        if st is not None and lx is not None and gemm_lnx_ok(g.M, 3 * C, C, tok.ld, qkv.ld) and (at.pe is None or g.f <= at.pe.shape[0]):
            if at.pe is not None:   # (LayerNorm(x) + pe_f) W^T: the per-frame row table, one row per (sample, frame)
                tab = lx.tab[:g.f].repeat(g.b, 1).contiguous()
                emit_gemm(plan, tok, lx.W, 3 * C, C, qkv, rowvec=(tab, 0, 3 * C, g.hw), lnx=(st, lx.S))
            else:
                emit_gemm(plan, tok, lx.W, 3 * C, C, qkv, bias=lx.b, lnx=(st, lx.S))
        else:
            emit_layernorm(plan, tok, at.ln_g, at.ln_b, a, pe=at.pe, rows_per_frame=g.hw, frames=g.f)
            emit_gemm(plan, a, at.qkv, 3 * C, C, qkv)
        ao = plan.rows("attn_out", g.M, C)
        emit_temporal_attn(plan, qkv, g.b, g.f, g.hw, heads, d_head, ao)
        st = emit_gemm(plan, ao, at.o, C, C, tok, bias=at.o_b, residual=tok, stat=want_stat)
    emit_ff(plan, tok, w.ff_ln[0], w.ff_ln[1], w.ff1, w.ff1_b, w.ff2, w.ff2_b, a, g.M, C, stream=w.ff_stream, tok_stat=st,
            lnx=w.lnx_ff, z=z, out_gn=out_gn)
    if z is None:
        emit_gemm(plan, tok, w.proj_out, C, C, out, bias=w.proj_out_b, residual=x, gn=out_gn)
