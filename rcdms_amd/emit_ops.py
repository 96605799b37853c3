"""Single-launch emitters: each appends ONE librcdm_hip.so entry point (include/rcdm.h) to a Plan, with the descriptor,
workspace and operand pointers it needs, tagged with kind + shape for the per-op profilers (tools/opprof.py).  The
reference counterpart of every launch is cited at the block emitters (emit_blocks.py) and in rcdm.h."""
import torch

from . import hip
from . import switches as SW
from .plan import _NS


def _gn_handoff(plan, out, N, gn, ok_fn, d):
    """gn = (samples, rows_per_sample, groups) of the GroupNorm that reads `out` NEXT (or None).  Where the launch is split-K
    and the library takes the pair (rcdm_*_gnstat_ok), its reduce pass also leaves that norm's partial statistics in the
    shared "gn_ws" scratch: returns (GroupNormDesc, workspace Buf) and the caller records plan.gn_ready after adding its
    op; emit_groupnorm, if it is the very next op and reads exactly these rows, then launches finalize + apply only."""
    if gn is None or not SW.GN_PRESTAT:
        return None
    samples, rps, groups = gn
    if samples * rps != out.M or N % groups:
        return None
    gnd = hip.GroupNormDesc(samples, rps, N, groups, out.ld, out.ld, 1e-5, 0)
    if not ok_fn(d, gnd):
        return None
    return gnd, plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(gnd))


def assert_no_pending_gn(plan, who):
    """A producer that left GroupNorm statistics in the shared "gn_ws" scratch (plan.gn_ready, _gn_handoff) must be followed
    by the emit_groupnorm that consumes them; any other op that is emitted right behind it and touches that scratch (or
    simply is not that norm) means the planner asked for a hand-off nobody takes: wasted reduce work and stale state."""
    rdy = getattr(plan, "gn_ready", None)
    plan.gn_ready = None
    assert rdy is None or rdy["n_ops"] != len(plan.ops), f"a producer left GroupNorm statistics, but the next op is {who}"


def emit_gemm(plan, A, Wt, N, K, out, bias=None, rowvec=None, residual=None, geglu=False, scale=1.0, split_k=0,
              gelu=False, dup_rows=0, stat=False, lnx=None, gn=None, stat_into=None, stat_row0=0):
    """out[M][N or N/2] = epi(A[M][K] W[N][K]^T); rowvec = (tensor, elem_offset, ldt, rows_per_sample).
    Deferred LayerNorm (rcdm_gemm_lnx): stat=True — also write the row statistics of the stored rows and RETURN their handle
    (None when this shape has no statistics-producing launch: the caller then emits the stand-alone LayerNorm);
    lnx=(handle, S) — A holds the RAW rows whose LayerNorm this GEMM consumes, Wt / bias carry gamma / beta (Packer.lnx_*).
    Row subsets (the rank-1-context plan): stat_row0 — A / out are rows [stat_row0, stat_row0 + M) of the tensor the
    statistics handles index; stat_into=handle — write this launch's statistics INTO an earlier producer's buffer (its
    slot count, its plane stride: the rows this launch rewrites get new statistics, the others keep theirs) and return
    that handle, or None when the library has no tile with that slot count for this shape (rcdm_gemm_lnx_parts_ok)."""
    epi = 0
    if bias is not None:
        epi |= hip.EPI_BIAS
    if rowvec is not None:
        epi |= hip.EPI_ROWVEC
    if residual is not None:
        epi |= hip.EPI_RESIDUAL
    if geglu:
        epi |= hip.EPI_GEGLU
    if gelu:
        epi |= hip.EPI_GELU
    d = hip.GemmDesc(A.M, N, K, A.ld, out.ld, residual.ld if residual is not None else 0, epi,
                     rowvec[3] if rowvec else 1, rowvec[2] if rowvec else 0, scale, split_k, dup_rows)
    handle, x = None, None
    if stat_into is not None:
        assert stat_into.C == N and stat_into.M >= stat_row0 + A.M and not dup_rows
        if (not geglu and hip.gemm_lnx_parts_ok(d, stat_into.parts, consumer=lnx is not None) and
                not hip.gemm_lnx_workspace_bytes(d, producer=True, consumer=lnx is not None)):
            assert stat_into.gen == getattr(plan, "rowstat_gen", 0), "the statistics buffer was reused since"
            handle = stat_into
    elif stat and SW.LNX and not geglu and not hip.gemm_lnx_workspace_bytes(d, producer=True, consumer=lnx is not None):
        parts = hip.gemm_stat_parts(d, consumer=lnx is not None)   # (asked with the flags the launch will carry)
        if 0 < parts <= LNX_MAX_PARTS:
            # the statistics of ALL producers live in one scratch buffer: a handle carries the generation it was written in,
            # and a consumer checks that nothing has overwritten it since (emit order = execution order)
            buf = plan.scratch("rowstat", (A.M + dup_rows) * parts * 8)
            plan.rowstat_gen = getattr(plan, "rowstat_gen", 0) + 1
            handle = _NS(buf=buf, parts=parts, M=A.M, C=N, rows=A.M + dup_rows, gen=plan.rowstat_gen)
    # the workspace question is asked with the flags the launch will carry: a statistics producer / a consumer is steered
    # to other tile shapes (and splits) than a plain GEMM of the same shape
    if handle is not None or lnx is not None:
        wsb = hip.gemm_lnx_workspace_bytes(d, producer=handle is not None, consumer=lnx is not None)
    else:
        wsb = hip.gemm_workspace_bytes(d)
    ws = plan.scratch("splitk_ws", max(wsb, 256))
    bptr = bias.data_ptr() if bias is not None else 0
    rv_t, rv_off = (rowvec[0], rowvec[1]) if rowvec else (None, 0)
    if lnx is not None:
        assert not wsb, "deferred LayerNorm consumer cannot be a split-K launch (gemm_lnx_ok)"
        assert lnx[0].C == K and lnx[0].M >= stat_row0 + A.M
        # (a call that is consumer AND producer reads its rows' statistics at kernel start and writes the new ones in its
        # epilogue, into the same buffer: legal only because both sides index it by the same rows of the same launch)
        assert lnx[0].gen >= getattr(plan, "rowstat_gen", 0) - (1 if handle is not None else 0), \
            "row statistics were overwritten by a later producer before this consumer was emitted"
    if handle is not None or lnx is not None:
        x = hip.Lnx(0, handle.parts if handle else 0, handle.rows if handle else 0, 0, lnx[0].parts if lnx else 0,
                    lnx[0].rows if lnx else 0, lnx[1].data_ptr() if lnx else 0, 1e-5, K)

    hand = _gn_handoff(plan, out, N, gn, hip.gemm_gnstat_ok, d) if (x is None and not geglu and not gelu) else None

    def op():
        rvp = (rv_t.data_ptr() + 4 * rv_off) if rv_t is not None else 0
        if hand is not None:
            hip.gemm_gnstat(d, hand[0], A.ptr, Wt.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr,
                            ws.ptr, ws.nbytes, hand[1].ptr, hand[1].nbytes)
            return
        if x is not None:
            x.stat_out = handle.buf.ptr + 8 * stat_row0 if handle is not None else 0     # slot-major planes of float2 per row
            x.stat_in = lnx[0].buf.ptr + 8 * stat_row0 if lnx is not None else 0
            hip.gemm_lnx(d, x, A.ptr, Wt.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr,
                         ws.ptr, ws.nbytes)
            return
        hip.gemm(d, A.ptr, Wt.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"gemm M={A.M} N={N} K={K} epi={epi}" + (" lnx" if lnx is not None else "") + (" stat" if handle is not None else "")
             + (" gnstat" if hand is not None else ""))
    plan.keep += [Wt, bias, rv_t, x, lnx[1] if lnx else None]
    plan.op_weights[len(plan.ops) - 1] = Wt
    plan.op_desc[len(plan.ops) - 1] = ("gemm", d, handle is not None, lnx is not None)
    if lnx is not None:
        plan.lnx_sites.append((len(plan.ops) - 1, A, lnx[1], plan.tags[-1]))
    plan.n_launch += 2 if wsb else 1
    if hand is not None:
        plan.gn_ready = dict(n_ops=len(plan.ops), key=out.ptr_key(), M=out.M, C=N, gn=gn)
    return handle


def gemm_lnx_ok(M, N, K, lda, ldc, geglu=False, dup_rows=0):
    """Whether a deferred-LayerNorm consumer GEMM of this shape is a single launch (no split-K slabs), asked the way the
    launch itself decides (consumer flag set: rcdm_gemm_lnx_workspace_bytes)."""
    d = hip.GemmDesc(M, N, K, lda, ldc, 0, hip.EPI_GEGLU if geglu else 0, 1, 0, 1.0, 0, dup_rows)
    return hip.gemm_lnx_workspace_bytes(d, consumer=True) == 0


def emit_conv3x3(plan, x, n_img, H, W, Wt, cin, cout, out, stride=1, up=0, bias=None, rowvec=None, residual=None,
                 scale=1.0, split_k=0, pad_after_only=0, dup_rows=0, gn=None, x2=None):
    """x2 (Rows of the output's row count): a second input whose 1x1 convolution is accumulated into the same output
    (rcdm_conv3x3_add1x1); Wt then carries its [cout][x2.C] columns behind the nine taps'."""
    epi = 0
    if bias is not None:
        epi |= hip.EPI_BIAS
    if rowvec is not None:
        epi |= hip.EPI_ROWVEC
    if residual is not None:
        epi |= hip.EPI_RESIDUAL
    d = hip.ConvDesc(n_img, H, W, cin, cout, stride, up, x.ld, out.ld, residual.ld if residual is not None else 0,
                     epi, rowvec[3] if rowvec else 1, rowvec[2] if rowvec else 0, scale, split_k, pad_after_only, dup_rows,
                     x2.C if x2 is not None else 0, x2.ld if x2 is not None else 0)
    wsb = hip.conv3x3_workspace_bytes(d)
    ws = plan.scratch("splitk_ws", max(wsb, 256))
    bptr = bias.data_ptr() if bias is not None else 0
    rv_t, rv_off = (rowvec[0], rowvec[1]) if rowvec else (None, 0)

    hand = _gn_handoff(plan, out, cout, gn, hip.conv3x3_gnstat_ok, d) if up != 2 else None

    def op():
        rvp = (rv_t.data_ptr() + 4 * rv_off) if rv_t is not None else 0
        rp = residual.ptr if residual is not None else 0
        if x2 is not None:
            if hand is not None:
                hip.conv3x3_add1x1_gnstat(d, hand[0], x.ptr, x2.ptr, Wt.data_ptr(), bptr, rvp, rp, out.ptr, ws.ptr, ws.nbytes,
                                          hand[1].ptr, hand[1].nbytes)
            else:
                hip.conv3x3_add1x1(d, x.ptr, x2.ptr, Wt.data_ptr(), bptr, rvp, rp, out.ptr, ws.ptr, ws.nbytes)
            return
        if hand is not None:
            hip.conv3x3_gnstat(d, hand[0], x.ptr, Wt.data_ptr(), bptr, rvp, rp, out.ptr,
                               ws.ptr, ws.nbytes, hand[1].ptr, hand[1].nbytes)
            return
        hip.conv3x3(d, x.ptr, Wt.data_ptr(), bptr, rvp, rp, out.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"conv3x3 {n_img}x{H}x{W} {cin}->{cout} s={stride} up={up} epi={epi}" + (f" add1x1={x2.C}" if x2 is not None else "")
             + (" gnstat" if hand is not None else ""))
    plan.op_weights[len(plan.ops) - 1] = Wt
    plan.op_desc[len(plan.ops) - 1] = ("conv", d)
    if hand is not None:
        plan.gn_ready = dict(n_ops=len(plan.ops), key=out.ptr_key(), M=out.M, C=cout, gn=gn)
    plan.keep += [Wt, bias, rv_t]
    plan.n_launch += 2 if wsb else 1


UP9_MIN_CHANNELS = 640   # the tap-plane form's GEMM has K = c: narrower convs keep the phase form


def emit_upsample_conv(plan, pk, wkey, x, n_img, H, W, c, out, bias):
    """Upsample3D.forward (src/models/resnet.py:60-79): F.interpolate(scale 2, nearest) + conv3x3, c -> c channels.
    Three exact forms, by cost: nine tap planes over the SOURCE pixels as one GEMM (N = 9 c) + a gather (9 products per source
    pixel), four 2x2 phase convolutions (16), the nine-tap implicit GEMM over the upsampled grid (36)."""
    if SW.UP9 and c >= UP9_MIN_CHANNELS and c % 8 == 0 and n_img * H * W * 9 * c * 2 < (1 << 31):
        P = plan.rows("up_taps", n_img * H * W, 9 * c)
        emit_gemm(plan, x, pk.conv3x3_taps(wkey), 9 * c, c, P)

        def op():
            hip.upsample_taps_gather(P.ptr, P.ld, n_img, H, W, c, bias.data_ptr() if bias is not None else 0, out.ptr, out.ld)
        plan.add(op, f"upsample_gather {n_img}x{H}x{W} C={c}")
        plan.keep += [bias]
        plan.n_launch += 1
        return
    d2 = hip.ConvDesc(n_img, H, W, c, c, 1, 2, x.ld, out.ld, 0, hip.EPI_BIAS if bias is not None else 0, 1, 0, 1.0, 0, 0, 0)
    if SW.UP2 and hip.conv3x3_up2_supported(d2):
        emit_conv3x3(plan, x, n_img, H, W, pk.conv3x3_up2(wkey), c, c, out, up=2, bias=bias)
    else:
        emit_conv3x3(plan, x, n_img, H, W, pk.conv3x3(wkey), c, c, out, up=1, bias=bias)


def emit_groupnorm(plan, x, samples, rows_per_sample, gamma, beta, eps, silu, out, groups=32):
    d = hip.GroupNormDesc(samples, rows_per_sample, x.C, groups, x.ld, out.ld, eps, int(silu))
    ws = plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(d))
    rdy = getattr(plan, "gn_ready", None)
    plan.gn_ready = None
    match = (rdy is not None and rdy["n_ops"] == len(plan.ops) and rdy["key"] == x.ptr_key() and rdy["M"] == x.M and
             rdy["C"] == x.C and rdy["gn"] == (samples, rows_per_sample, groups))
    tiles = match and rdy.get("tiles") is not None     # per-tile partials of a Winograd conv's output transform (its own geometry)
    pre = match and not tiles and hip.groupnorm_prestat_ok(d)
    assert rdy is None or rdy["n_ops"] != len(plan.ops) or pre or tiles, "a producer left GroupNorm statistics that nobody consumes"
    stat = plan.scratch("gn_stat", samples * groups * 2 * 4) if tiles else None

    def op():
        if tiles:   # finalize the producer's per-tile partials, then the apply launch alone
            hip.groupnorm_finalize(samples, groups, rdy["tiles"][1], eps, rdy["tiles"][0].ptr, stat.ptr)
            hip.groupnorm_apply(d, x.ptr, stat.ptr, gamma.data_ptr(), beta.data_ptr(), out.ptr)
        elif pre:   # the partial statistics are in ws already (the producer's reduce pass)
            hip.groupnorm_silu_prestat(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), out.ptr, ws.ptr, ws.nbytes)
        else:
            hip.groupnorm_silu(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), out.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"groupnorm S={samples} R={rows_per_sample} C={x.C} silu={int(silu)}" + (" prestat" if pre or tiles else ""))
    plan.keep += [gamma, beta]
    plan.n_launch += 2 if (pre or tiles) else 3


def emit_groupnorm_stats(plan, x, samples, rows_per_sample, gamma, beta, eps, groups=32):
    """(mean, rstd) of a GroupNorm only — the consumer (emit_rowchain's `gn`, emit_conv3x3_wino's `gn`) applies it while loading
    its rows.  Returns the `gn` tuple they take.  Where the producer of x left this norm's partial statistics (plan.gn_ready,
    _gn_handoff) only the finalize launch is emitted."""
    d = hip.GroupNormDesc(samples, rows_per_sample, x.C, groups, x.ld, x.ld, eps, 0)
    ws = plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(d))
    stat = plan.scratch("gn_stat", samples * groups * 2 * 4)
    rdy = getattr(plan, "gn_ready", None)
    plan.gn_ready = None
    match = (rdy is not None and rdy["n_ops"] == len(plan.ops) and rdy["key"] == x.ptr_key() and rdy["M"] == x.M and
             rdy["C"] == x.C and rdy["gn"] == (samples, rows_per_sample, groups))
    tiles = match and rdy.get("tiles") is not None     # per-tile partials of a Winograd conv's output transform (its own geometry)
    pre = match and (tiles or hip.groupnorm_prestat_ok(d))
    assert rdy is None or rdy["n_ops"] != len(plan.ops) or pre, "a producer left GroupNorm statistics that nobody consumes"

    def op():
        if tiles:
            hip.groupnorm_finalize(samples, groups, rdy["tiles"][1], eps, rdy["tiles"][0].ptr, stat.ptr)
        elif pre:
            hip.groupnorm_stats_prestat(d, stat.ptr, ws.ptr, ws.nbytes)
        else:
            hip.groupnorm_stats(d, x.ptr, stat.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"groupnorm_stats S={samples} R={rows_per_sample} C={x.C}" + (" prestat" if pre else ""))
    plan.keep += [gamma, beta]
    plan.n_launch += 1 if pre else 2
    return (stat, gamma, beta, groups, rows_per_sample)


def conv3x3_wino_ok(n_img, H, W, cin, cout, cin2=0):
    """Whether rcdm_conv3x3_wino takes a stride-1 conv of this geometry (even image sides, whole 64-channel k-steps)."""
    d = hip.ConvDesc(n_img, H, W, cin, cout, 1, 0, cin, cout, 0, 0, 1, 0, 1.0, 0, 0, 0, cin2, cin2)
    return hip.conv3x3_wino_supported(d)


def emit_conv3x3_wino(plan, x, n_img, H, W, U, cin, cout, out, bias=None, rowvec=None, residual=None, scale=1.0, split_k=0,
                      gn=None, silu=True, x2=None, W2=None, gn_out=None, gn_out_apply=False):
    """Stride-1 conv3x3 in the Winograd F(2x2, 3x3) form (rcdm_conv3x3_wino; ResnetBlock3D conv1 / conv2, resnet.py:188,205-212).
    gn = emit_groupnorm_stats(...) of the norm in front of the conv: x holds its RAW input and the input transform applies the
    normalisation (+ SiLU) on the way.  x2 / W2: the second input and its plain f16 [cout][x2.C] 1x1 weight (conv_shortcut).
    gn_out = (samples, rows_per_sample, groups) of the GroupNorm whose STATISTICS-ONLY launch (emit_groupnorm_stats) is the very
    next op on `out`: the output transform leaves per-tile partials and that launch becomes the finalize alone.  gn_out_apply:
    the next op is the norm's full form (emit_groupnorm: finalize + apply then) — only worth it where that norm has three launches."""
    epi = 0
    if bias is not None:
        epi |= hip.EPI_BIAS
    if rowvec is not None:
        epi |= hip.EPI_ROWVEC
    if residual is not None:
        epi |= hip.EPI_RESIDUAL
    d = hip.ConvDesc(n_img, H, W, cin, cout, 1, 0, x.ld, out.ld, residual.ld if residual is not None else 0, epi,
                     rowvec[3] if rowvec else 1, rowvec[2] if rowvec else 0, scale, split_k, 0, 0,
                     x2.C if x2 is not None else 0, x2.ld if x2 is not None else 0)
    assert hip.conv3x3_wino_supported(d) and (x2 is None) == (W2 is None)
    ws = plan.scratch("wino_ws", hip.conv3x3_wino_workspace_bytes(d))
    assert_no_pending_gn(plan, "a Winograd conv launch")
    gnd = None
    if gn is not None:
        stat, gamma, beta, groups, rps = gn
        gnd = hip.GroupNormDesc(x.M // rps, rps, cin, groups, x.ld, x.ld, 1e-5, int(silu))
    god, gop = None, None
    if gn_out is not None and SW.GN_PRESTAT and gn_out[0] * gn_out[1] == out.M and gn_out[1] % (H * W) == 0 and cout % gn_out[2] == 0 \
            and cout * 8 <= 65536 and gn_out[2] <= 64:
        god = hip.GroupNormDesc(gn_out[0], gn_out[1], cout, gn_out[2], out.ld, out.ld, 1e-5, 0)
        if gn_out_apply and not hip.groupnorm_prestat_ok(god):
            god = None     # the consumer is a single-launch norm (<= 512 rows per sample): nothing to save
        else:
            gop = plan.scratch("gn_tile_part", gn_out[0] * gn_out[2] * (gn_out[1] // 4) * 3 * 4)
    bptr = bias.data_ptr() if bias is not None else 0
    rv_t, rv_off = (rowvec[0], rowvec[1]) if rowvec else (None, 0)

    def op():
        rvp = (rv_t.data_ptr() + 4 * rv_off) if rv_t is not None else 0
        hip.conv3x3_wino(d, x.ptr, U.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr, ws.ptr, ws.nbytes,
                         x2=x2.ptr if x2 is not None else 0, W2=W2.data_ptr() if W2 is not None else 0, gn=gnd,
                         gn_stat=gn[0].ptr if gn is not None else 0, gn_gamma=gn[1].data_ptr() if gn is not None else 0,
                         gn_beta=gn[2].data_ptr() if gn is not None else 0, gn_out=god, gn_out_partial=gop.ptr if gop is not None else 0)
    plan.add(op, f"conv3x3_wino {n_img}x{H}x{W} {cin}->{cout} epi={epi}" + (f" add1x1={x2.C}" if x2 is not None else "")
             + (" gn" if gn is not None else "") + (" gnstat" if god is not None else ""))
    if god is not None:
        plan.gn_ready = dict(n_ops=len(plan.ops), key=out.ptr_key(), M=out.M, C=cout, gn=tuple(gn_out), tiles=(gop, gn_out[1] // 4))
    plan.op_weights[len(plan.ops) - 1] = U
    plan.op_desc[len(plan.ops) - 1] = ("wino", d)
    plan.keep += [U, W2, bias, rv_t, gnd, god]
    plan.n_launch += 3


def emit_layernorm(plan, x, gamma, beta, out, pe=None, rows_per_frame=1, frames=1):
    d = hip.LayerNormDesc(x.M, x.C, x.ld, out.ld, 1e-5, rows_per_frame, frames)

    def op():
        hip.layernorm(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), pe.data_ptr() if pe is not None else 0, out.ptr)
    plan.add(op, f"layernorm M={x.M} C={x.C} pe={int(pe is not None)}")
    plan.keep += [gamma, beta, pe]
    plan.n_launch += 1


def emit_flash_attn(plan, q, k, v, batch, heads, Lq, Lk, d_head, out, wide=False, bound=None):
    """wide: the caller has no bound |scaled score| < 2^15 for this site (rcdm.h, rcdm_flash_attn): the fp32-argument softmax
    kernel is used where the d = 40 kernel would take its softmax argument from the matrix pipe (attn_score_bound).
    bound: that weight-norm bound, recorded with the site for numerics_report."""
    d = hip.AttnDesc(batch, heads, Lq, Lk, d_head, q.ld, k.ld, v.ld, out.ld, d_head ** -0.5, hip.ATTN_WIDE_RANGE if wide else 0)

    def op():
        hip.flash_attn(d, q.ptr, k.ptr, v.ptr, out.ptr)
    plan.add(op, f"flash_attn B={batch} H={heads} Lq={Lq} Lk={Lk} d={d_head}")
    if bound is not None:
        plan.attn_sites.append((len(plan.ops) - 1, plan.tags[-1], float(bound), bool(wide), q, k, heads, d_head))
    plan.n_launch += 1


LNX_MAX_PARTS = 20
XATTN_MAX_KEYS = 96   # rcdm_xattn: cross-attention with all scores of a query in registers


def emit_xattn_pack(plan, k, v, batch, heads, Lk, d_head):
    """Fragment-major K / V image of a context for rcdm_xattn (written once per context, next to its [K | V] GEMM)."""
    img = torch.empty(hip.xattn_image_bytes(batch, heads, d_head), dtype=torch.uint8, device=plan.device)
    plan.keep.append(img)

    def op():
        hip.xattn_pack_kv(k.ptr, v.ptr, batch, Lk, heads, d_head, k.ld, v.ld, img.data_ptr())
    plan.add(op, f"xattn_pack B={batch} H={heads} Lk={Lk} d={d_head}")
    plan.n_launch += 1
    return img


def emit_xattn(plan, q, img, batch, heads, Lq, Lk, d_head, out):
    d = hip.AttnDesc(batch, heads, Lq, Lk, d_head, q.ld, 0, 0, out.ld, d_head ** -0.5)

    def op():
        hip.xattn(d, q.ptr, img.data_ptr(), out.ptr)
    plan.add(op, f"xattn B={batch} H={heads} Lq={Lq} Lk={Lk} d={d_head}")
    plan.n_launch += 1


def emit_flash_attn_masked(plan, q, k, v, batch, heads, Lq, Lk, d_head, out, key_valid, causal):
    """key_valid: uint8 tensor [batch][Lk] (or None); causal: bool or a callable evaluated at launch time."""
    d = hip.AttnDesc(batch, heads, Lq, Lk, d_head, q.ld, k.ld, v.ld, out.ld, d_head ** -0.5)

    def op():
        c = causal() if callable(causal) else causal
        hip.flash_attn_masked(d, q.ptr, k.ptr, v.ptr, key_valid.data_ptr() if key_valid is not None else 0, bool(c), out.ptr)
    plan.add(op, f"flash_attn_masked B={batch} H={heads} L={Lq} d={d_head}")
    plan.keep += [key_valid]
    plan.n_launch += 1


def emit_temporal_attn(plan, qkv, samples, frames, pixels, heads, d_head, out):
    d = hip.TemporalAttnDesc(samples, frames, pixels, heads, d_head, qkv.ld, out.ld, d_head ** -0.5)

    def op():
        hip.temporal_attn(d, qkv.ptr, out.ptr)
    plan.add(op, f"temporal_attn S={samples} F={frames} P={pixels} H={heads} d={d_head}")
    plan.n_launch += 1
