"""Host-side launch planner for the stage-2 denoiser on MI355X.

Turns the reference's module graph (UNet3DConditionModel.forward, src/models/unet.py:322-463 and the
block wiring of src/models/unet_blocks.py) into a flat, static list of librcdm_hip.so launches over
pre-allocated HBM buffers, so that one denoising step can be captured once into a hipGraph and
replayed (RCDMs_pipeline.py:480-503 calls it T times per story).

Data layout in HBM: every activation is channels-last f16 rows
X[(b f y x)][C] with an explicit row stride, so
  * every einops permute / .contiguous() of the reference is index arithmetic, and
  * torch.cat([h, skip], dim=1) (unet_blocks.py:644,754) is free: each skip tensor is WRITTEN by its
    producer straight into the right-hand columns of the concat buffer its consumer will read.
Weights are repacked once to f16 kernel layouts (fused [q;k;v], [k;v], GEGLU row interleave,
conv3x3 tap-major); cross-attention K/V of the context are computed once per context, not per step.

torch is used for device memory, streams, host<->device copies and the one-time pack-time weight algebra (folding a
LayerNorm's gamma / beta into the matrix behind it, composing proj_out with the feed-forward's second Linear); nothing
torch computes is on the per-step path.
"""
import math
import os

import torch

from . import hip


class Buf:
    """A device buffer whose size is the max over all requests made while planning."""
    __slots__ = ("name", "nbytes", "t")

    def __init__(self, name, nbytes):
        self.name, self.nbytes, self.t = name, int(nbytes), None

    @property
    def ptr(self):
        return self.t.data_ptr()


class Rows:
    """View of f16 rows [M][C] with row stride ld (elements) inside a Buf at element offset off."""
    __slots__ = ("buf", "off", "M", "C", "ld")

    def __init__(self, buf, off, M, C, ld):
        self.buf, self.off, self.M, self.C, self.ld = buf, int(off), int(M), int(C), int(ld)

    @property
    def ptr(self):
        return self.buf.t.data_ptr() + 2 * self.off

    def ptr_key(self):
        """Identity of the first element (valid before the buffers are materialised, unlike .ptr)."""
        return (id(self.buf), self.off, self.ld)

    def cols(self, c0, c):
        return Rows(self.buf, self.off + c0, self.M, c, self.ld)

    def rows(self, r0, n):
        return Rows(self.buf, self.off + r0 * self.ld, n, self.C, self.ld)


class Plan:
    """Ordered launch list + the buffers it touches."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs = {}
        self.ops = []
        self.tags = []  # one label per op (kind + shape): tools/opprof.py aggregates per-op timings by it
        self.keep = []  # tensors that must outlive the plan (packed weights etc.)
        self.n_launch = 0
        self.op_weights = {}   # op index -> weight tensor of a GEMM / conv op (tools/prefetch_bound.py)

    def scratch(self, name, nbytes):
        b = self.bufs.get(name)
        if b is None:
            b = self.bufs[name] = Buf(name, nbytes)
        elif nbytes > b.nbytes:
            assert b.t is None, "scratch grown after materialize"
            b.nbytes = int(nbytes)
        return b

    def new(self, name, nbytes):
        assert name not in self.bufs, name
        b = self.bufs[name] = Buf(name, nbytes)
        return b

    def rows(self, name, M, C, ld=None, unique=False):
        ld = ld or C
        buf = (self.new if unique else self.scratch)(name, M * ld * 2)
        return Rows(buf, 0, M, C, ld)

    def materialize(self):
        for b in self.bufs.values():
            if b.t is None:
                b.t = torch.zeros(max(b.nbytes, 256), dtype=torch.uint8, device=self.device)

    def total_bytes(self):
        return sum(b.nbytes for b in self.bufs.values())

    def add(self, fn, tag="misc"):
        if _DROP and tag.split()[0] in _DROP:   # timing experiments only (RCDM_DROP_OPS): the plan computes garbage
            return
        self.ops.append(fn)
        self.tags.append(tag)

    def run(self, ops=None):
        for op in (self.ops if ops is None else ops):
            op()


# upper bound of a fusion before it is built: RCDM_DROP_OPS=layernorm,temporal_attn,... leaves every op of those kinds out of
# the launch plan (wrong results; tools/ab_env.sh RCDM_DROP_OPS "" layernorm gives what removing all of them could buy at most)
_DROP = frozenset(k for k in os.environ.get("RCDM_DROP_OPS", "").split(",") if k)
if _DROP:
    import sys
    print(f"[rcdms_amd] WARNING: RCDM_DROP_OPS={','.join(sorted(_DROP))} — these op kinds are LEFT OUT of every launch plan; "
          "results are garbage (timing experiments only)", file=sys.stderr, flush=True)

# ------------------------------------------------------------------------------------------------
# single-kernel emitters

def _gn_handoff(plan, out, N, gn, ok_fn, d):
    """gn = (samples, rows_per_sample, groups) of the GroupNorm that reads `out` NEXT (or None).  Where the launch is split-K
    and the library takes the pair (rcdm_*_gnstat_ok), its reduce pass also leaves that norm's partial statistics in the
    shared "gn_ws" scratch: returns (GroupNormDesc, workspace Buf) and the caller records plan.gn_ready after adding its
    op; emit_groupnorm, if it is the very next op and reads exactly these rows, then launches finalize + apply only."""
    if gn is None or not GN_PRESTAT:
        return None
    samples, rps, groups = gn
    if samples * rps != out.M or N % groups:
        return None
    gnd = hip.GroupNormDesc(samples, rps, N, groups, out.ld, out.ld, 1e-5, 0)
    if not ok_fn(d, gnd):
        return None
    return gnd, plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(gnd))


def emit_gemm(plan, A, Wt, N, K, out, bias=None, rowvec=None, residual=None, geglu=False, scale=1.0, split_k=0,
              gelu=False, dup_rows=0, stat=False, lnx=None, gn=None):
    """out[M][N or N/2] = epi(A[M][K] W[N][K]^T); rowvec = (tensor, elem_offset, ldt, rows_per_sample).
    Deferred LayerNorm (rcdm_gemm_lnx): stat=True — also write the row statistics of the stored rows and RETURN their handle
    (None when this shape has no statistics-producing launch: the caller then emits the stand-alone LayerNorm);
    lnx=(handle, S) — A holds the RAW rows whose LayerNorm this GEMM consumes, Wt / bias carry gamma / beta (Packer.lnx_*)."""
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
    if stat and LNX and not geglu and not hip.gemm_lnx_workspace_bytes(d, producer=True, consumer=lnx is not None):
        parts = hip.gemm_stat_parts(d)
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
        assert lnx[0].C == K and lnx[0].M >= A.M
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
            x.stat_out = handle.buf.ptr if handle is not None else 0
            x.stat_in = lnx[0].buf.ptr if lnx is not None else 0
            hip.gemm_lnx(d, x, A.ptr, Wt.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr,
                         ws.ptr, ws.nbytes)
            return
        hip.gemm(d, A.ptr, Wt.data_ptr(), bptr, rvp, residual.ptr if residual is not None else 0, out.ptr, ws.ptr, ws.nbytes)
    n_before = len(plan.ops)
    plan.add(op, f"gemm M={A.M} N={N} K={K} epi={epi}" + (" lnx" if lnx is not None else "") + (" stat" if handle is not None else "")
             + (" gnstat" if hand is not None else ""))
    plan.keep += [Wt, bias, rv_t, x, lnx[1] if lnx else None]
    if len(plan.ops) > n_before:
        plan.op_weights[len(plan.ops) - 1] = Wt
    plan.n_launch += 2 if wsb else 1
    if len(plan.ops) == n_before:   # the op was left out (RCDM_DROP_OPS)
        return None
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
    n_before = len(plan.ops)
    plan.add(op, f"conv3x3 {n_img}x{H}x{W} {cin}->{cout} s={stride} up={up} epi={epi}" + (f" add1x1={x2.C}" if x2 is not None else "")
             + (" gnstat" if hand is not None else ""))
    if len(plan.ops) > n_before:
        plan.op_weights[len(plan.ops) - 1] = Wt
        if hand is not None:
            plan.gn_ready = dict(n_ops=len(plan.ops), key=out.ptr_key(), M=out.M, C=cout, gn=gn)
    plan.keep += [Wt, bias, rv_t]
    plan.n_launch += 2 if wsb else 1


# ResnetBlock3D's conv_shortcut folded into conv2's implicit GEMM (rcdm_conv3x3_add1x1); RCDM_SC_FOLD=0 keeps the separate
# 1x1 GEMM + residual read (same-process A/B)
SC_FOLD = os.environ.get("RCDM_SC_FOLD", "1") != "0"
# Upsample3D's nearest-2x + conv3x3 as four 2x2 phase convolutions over the source grid (rcdm_conv3x3 upsample = 2: 4/9 of
# the multiply-adds) wherever the library takes the shape; RCDM_UP2=0 keeps the upsample = 1 form (same-process A/B)
UP2 = os.environ.get("RCDM_UP2", "1") != "0"


def emit_upsample_conv(plan, pk, wkey, x, n_img, H, W, c, out, bias):
    """Upsample3D.forward (src/models/resnet.py:60-79): F.interpolate(scale 2, nearest) + conv3x3, c -> c channels."""
    d2 = hip.ConvDesc(n_img, H, W, c, c, 1, 2, x.ld, out.ld, 0, hip.EPI_BIAS if bias is not None else 0, 1, 0, 1.0, 0, 0, 0)
    if UP2 and hip.conv3x3_up2_supported(d2):
        emit_conv3x3(plan, x, n_img, H, W, pk.conv3x3_up2(wkey), c, c, out, up=2, bias=bias)
    else:
        emit_conv3x3(plan, x, n_img, H, W, pk.conv3x3(wkey), c, c, out, up=1, bias=bias)


# A split-K producer whose reduce pass leaves the statistics of the GroupNorm behind it (rcdm_*_gnstat; _gn_handoff): the
# norm then runs finalize + apply only.  RCDM_GN_PRESTAT=0: every norm takes its own statistics pass (same-process A/B).
GN_PRESTAT = os.environ.get("RCDM_GN_PRESTAT", "1") != "0"


def emit_groupnorm(plan, x, samples, rows_per_sample, gamma, beta, eps, silu, out, groups=32):
    d = hip.GroupNormDesc(samples, rows_per_sample, x.C, groups, x.ld, out.ld, eps, int(silu))
    ws = plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(d))
    rdy = getattr(plan, "gn_ready", None)
    plan.gn_ready = None
    pre = (rdy is not None and rdy["n_ops"] == len(plan.ops) and rdy["key"] == x.ptr_key() and rdy["M"] == x.M and
           rdy["C"] == x.C and rdy["gn"] == (samples, rows_per_sample, groups) and hip.groupnorm_prestat_ok(d))
    assert rdy is None or rdy["n_ops"] != len(plan.ops) or pre, "a producer left GroupNorm statistics that nobody consumes"

    def op():
        if pre:   # the partial statistics are in ws already (the producer's reduce pass)
            hip.groupnorm_silu_prestat(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), out.ptr, ws.ptr, ws.nbytes)
        else:
            hip.groupnorm_silu(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), out.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"groupnorm S={samples} R={rows_per_sample} C={x.C} silu={int(silu)}" + (" prestat" if pre else ""))
    plan.keep += [gamma, beta]
    plan.n_launch += 2 if pre else 3


def emit_groupnorm_stats(plan, x, samples, rows_per_sample, gamma, beta, eps, groups=32):
    """(mean, rstd) of a GroupNorm only — the consumer (emit_rowchain's `gn`) applies it while loading its rows.
    Returns the `gn` tuple emit_rowchain takes."""
    d = hip.GroupNormDesc(samples, rows_per_sample, x.C, groups, x.ld, x.ld, eps, 0)
    ws = plan.scratch("gn_ws", hip.groupnorm_workspace_bytes(d))
    stat = plan.scratch("gn_stat", samples * groups * 2 * 4)

    def op():
        hip.groupnorm_stats(d, x.ptr, stat.ptr, ws.ptr, ws.nbytes)
    plan.add(op, f"groupnorm_stats S={samples} R={rows_per_sample} C={x.C}")
    plan.keep += [gamma, beta]
    plan.n_launch += 2
    return (stat, gamma, beta, groups, rows_per_sample)


def emit_layernorm(plan, x, gamma, beta, out, pe=None, rows_per_frame=1, frames=1):
    d = hip.LayerNormDesc(x.M, x.C, x.ld, out.ld, 1e-5, rows_per_frame, frames)

    def op():
        hip.layernorm(d, x.ptr, gamma.data_ptr(), beta.data_ptr(), pe.data_ptr() if pe is not None else 0, out.ptr)
    plan.add(op, f"layernorm M={x.M} C={x.C} pe={int(pe is not None)}")
    plan.keep += [gamma, beta, pe]
    plan.n_launch += 1


def emit_flash_attn(plan, q, k, v, batch, heads, Lq, Lk, d_head, out, wide=False):
    """wide: the caller has no bound |scaled score| < 2^15 for this site (rcdm.h, rcdm_flash_attn): the fp32-argument softmax
    kernel is used where the d = 40 kernel would take its softmax argument from the matrix pipe (attn_score_bound)."""
    d = hip.AttnDesc(batch, heads, Lq, Lk, d_head, q.ld, k.ld, v.ld, out.ld, d_head ** -0.5, hip.ATTN_WIDE_RANGE if wide else 0)

    def op():
        hip.flash_attn(d, q.ptr, k.ptr, v.ptr, out.ptr)
    plan.add(op, f"flash_attn B={batch} H={heads} Lq={Lq} Lk={Lk} d={d_head}")
    plan.n_launch += 1


# Deferred LayerNorm (rcdm_gemm_lnx) wherever the row-stationary chains are not used (the 32x32 / 16x16 / 8x8 levels): the
# GEMM in front of a LayerNorm emits row statistics, the GEMM behind it takes the raw rows with gamma / beta folded into its
# weights — no LayerNorm launch, no normalised tensor in HBM.  RCDM_LNX=0 keeps the stand-alone launches (same-process A/B).
LNX = os.environ.get("RCDM_LNX", "1") != "0"
# proj_out folded into the feed-forward's second GEMM below the chain kernels' row count (Packer.ffz): the token rows live in
# the last C columns of a [M][5C] buffer whose first 4C columns the GEGLU projection fills, so one K = 5C GEMM replaces
# ff.net.2 (+ residual) and proj_out (+ residual).  RCDM_FFZ=0: the two GEMMs (same-process A/B)
FFZ = os.environ.get("RCDM_FFZ", "1") != "0"
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


# ------------------------------------------------------------------------------------------------
# weight packing (fp32 reference layout -> f16 kernel layout), once

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
        if not FFZ:
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
        if not LNX:
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
        if not LNX:
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
        if not (FF_FUSE and w1.shape == (8 * Cc, Cc) and w2.shape == (Cc, 4 * Cc) and hip.ff_fused_supported(Cc)):
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
        if not ROW_CHAIN or any(not self.has(k) for k in (wa_key, *wt_keys, *(ff_keys or ()))):
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


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def pack_resnet(pk, p):
    w = _NS(cin=pk.sd[p + "conv1.weight"].shape[1], cout=pk.sd[p + "conv1.weight"].shape[0])
    w.g1, w.b1 = pk.vec(p + "norm1.weight"), pk.vec(p + "norm1.bias")
    w.g2, w.b2 = pk.vec(p + "norm2.weight"), pk.vec(p + "norm2.bias")
    w.conv1, w.cb1 = pk.conv3x3(p + "conv1.weight"), pk.vec(p + "conv1.bias")
    w.conv2, w.cb2 = pk.conv3x3(p + "conv2.weight"), pk.vec(p + "conv2.bias")
    w.shortcut = w.conv2sc = None
    if pk.has(p + "conv_shortcut.weight"):
        w.shortcut, w.sb = pk.mat_f16(p + "conv_shortcut.weight"), pk.vec(p + "conv_shortcut.bias")
        if SC_FOLD and w.cin % 64 == 0 and w.cout % 64 == 0:
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
    LayerNorm in front (nothing bounds the rows).  Evaluated once per block at pack time, in fp64."""
    if ln is None or not pk.has(a + "to_q.weight"):
        return float("inf")
    gamma, beta = ln[0].double(), ln[1].double()
    C = gamma.numel()
    out = []
    for n in "qk":
        W = pk.sd[a + f"to_{n}.weight"].detach().to(gamma.device).double()
        bias = pk.sd[a + f"to_{n}.bias"].detach().to(gamma.device).double() if pk.has(a + f"to_{n}.bias") else None
        d = W.shape[0] // heads
        Wh = (W * gamma[None, :]).reshape(heads, d, C)
        off = (W @ beta + (bias if bias is not None else 0)).reshape(heads, d)
        out.append(math.sqrt(C) * Wh.pow(2).sum(dim=(1, 2)).sqrt() + off.pow(2).sum(dim=1).sqrt())
    d = pk.sd[a + "to_q.weight"].shape[0] // heads
    return float((out[0] * out[1]).max().item() * d ** -0.5 * 1.4426950408889634)


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
    if CHAIN_PROJ and w.ch_o2_ff is not None:
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
            if CHAIN_PROJ:   # proj_out + the module's residual behind the feed-forward (zero-initialised proj_out included)
                w.chain_ffz = pk.chain(a1 + "to_out.0.weight", 2, wt_keys=(p + "proj_out.weight",),
                                       ff_keys=(b + "ff.net.0.proj.weight", b + "ff.net.0.proj.bias", b + "ff.net.2.weight"))
    return w


# ------------------------------------------------------------------------------------------------
# block emitters

class Geo:
    """b samples x f frames of H x W latent pixels."""

    def __init__(self, b, f, H, W):
        self.b, self.f, self.H, self.W = b, f, H, W
        self.n_img = b * f
        self.hw = H * W
        self.M = self.n_img * self.hw


def emit_resnet(plan, w, x, geo, temb, out, eps=1e-5, groups=32, out_scale=1.0, dup_rows=0, out_gn=None):
    """ResnetBlock3D.forward (src/models/resnet.py:182-212).  temb = (tensor [b][ldt] fp32, col offset, ldt)
    = this block's slice of the batched time_emb_proj(silu(emb)) table.  x may be a concat view.
    out_gn = (samples, rows_per_sample, groups) of the GroupNorm that reads `out` as the very next op (the norm in front of
    the transformer / motion module / ResNet block behind this one), or None: a split-K conv2 then leaves its statistics."""
    g = geo
    a1 = plan.rows("norm", g.M, x.C)
    emit_groupnorm(plan, x, g.b, g.f * g.hw, w.g1, w.b1, eps, True, a1, groups)   # (first: x may carry a producer's statistics)
    res = x
    fold_sc = w.conv2sc is not None
    assert not fold_sc or x.C == w.cin
    if w.shortcut is not None:
        res = plan.rows("res_sc", g.M, w.cout)
        emit_gemm(plan, x, w.shortcut, w.cout, w.cin, res, bias=w.sb)
    h1 = plan.rows("res_h1", g.M, w.cout)
    emit_conv3x3(plan, a1, g.n_img, g.H, g.W, w.conv1, w.cin, w.cout, h1, bias=w.cb1,
                 rowvec=(temb[0], temb[1], temb[2], g.f * g.hw), gn=(g.b, g.f * g.hw, groups))
    a2 = plan.rows("norm", g.M, w.cout)
    emit_groupnorm(plan, h1, g.b, g.f * g.hw, w.g2, w.b2, eps, True, a2, groups)
    if fold_sc:   # (resnet.py:205-212 with a conv_shortcut: its 1x1 convolution of x rides in conv2's accumulators)
        emit_conv3x3(plan, a2, g.n_img, g.H, g.W, w.conv2sc, w.cout, w.cout, out, bias=w.cb2sc, scale=out_scale,
                     dup_rows=dup_rows, gn=out_gn, x2=x)
        return
    emit_conv3x3(plan, a2, g.n_img, g.H, g.W, w.conv2, w.cout, w.cout, out, bias=w.cb2, residual=res, scale=out_scale,
                 dup_rows=dup_rows, gn=out_gn)


# rcdm_ff_fused (rowff.hip): LayerNorm -> GEGLU feed-forward -> + residual as ONE row-stationary launch, for the channel
# counts the library has a kernel for.  RCDM_FF_FUSE=0 keeps the three-launch chain (same-process A/B).
FF_FUSE = os.environ.get("RCDM_FF_FUSE", "1") != "0"
# rcdm_rowchain (rowff.hip): [C x C projection (+ residual) -> LayerNorm (+ pe) -> q | qkv projection or feed-forward] as one
# row-stationary launch.  RCDM_ROWCHAIN=0 keeps the separate launches (same-process A/B).
ROW_CHAIN = os.environ.get("RCDM_ROWCHAIN", "1") != "0"
# a chain launch is one block of 160 rows per CU: below ~3/4 of a chip's worth of rows (the 256x256 configuration has
# 10240 token rows at this width = 64 blocks) the separate tile-parallel launches are faster
class _ChainMinRows:
    """Rows from which the row-stationary chain launches are used: 3/4 of a chip's worth of 160-row blocks, one block per
    CU — read from the device (hipDeviceAttributeMultiprocessorCount through torch) the first time a plan compares against
    it, 256 CUs (MI355X: 160 * 192 = 30720 rows) when no device is visible; RCDM_CHAIN_MIN_ROWS overrides."""

    def __init__(self):
        self._v = None

    def value(self):
        if self._v is None:
            env = os.environ.get("RCDM_CHAIN_MIN_ROWS")
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
CHAIN_PROJ = os.environ.get("RCDM_CHAIN_PROJ", "1") != "0"   # proj_out + residual as the trailing stage of the feed-forward chain
CHAIN_GN = os.environ.get("RCDM_CHAIN_GN", "1") != "0"   # GroupNorm apply in the prologue of the proj_in chain


def emit_rowchain(plan, a_in, res, tok, a_bias, ln, pe, stream, tail, out, rows_per_frame=1, frames=1, b2=None, gn=None,
                  z=None):
    """tok = a_in W_a^T + a_bias (+ res);  y = LayerNorm(tok) (+ pe);  tail 1 / 3: out = y W_t^T;  tail 0: out = tok + FF(y).
    ln = (gamma, beta); stream = Packer.chain(...); gn = emit_groupnorm_stats(...): a_in is the RAW input of that
    GroupNorm and the kernel applies it while loading its rows (res must be None).  tail 2, z = (z_res rows, z_bias):
    out = z_res + (tok + FF(y)) W_z^T + z_bias, the feed-forward's own output rows are not stored."""
    ws, b1p = stream
    M, C = a_in.M, a_in.C
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
                     tok_stat=None, z=None, out_gn=None):
    """BasicTransformerBlock.forward (src/models/attention.py:479-526) in place on tok [n_seq*Lq][C]:
    h += attn1(LN1(h)); h += attn2(LN2(h), ctx); h += FF(LN3(h)).  ctx_kv: Rows [n_seq*L][2C] = [K | V] of the context.
    shared_half: the two halves of tok (the CFG halves of a denoising step) hold IDENTICAL rows on entry — everything up
    to the query projection of the cross-attention is then computed on the first half only and stored to both.
    tok_stat: row statistics of tok from the GEMM that wrote it (emit_gemm(stat=True)); with them, and below the chain
    kernels' row count, the three LayerNorms are deferred into the epilogues of the GEMMs behind them (rcdm_gemm_lnx)."""
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
                    wide=w.score_bound >= MSUB_SCORE_LIMIT)
    chain_q = w.has_cross and w.ch_o1_q is not None and not shared_half and big
    # (below CHAIN_MIN_ROWS only: with more rows the N = C producers run on the ping-pong kernel, which has no statistics
    # epilogue — the b = 8 configuration measured 59.5 ms per step without and 60.0 with the deferred form at 40960 rows)
    want_ff = LNX and not big and w.geglu and w.lnx_ff is not None          # statistics for norm3 -> GEGLU
    # statistics for norm2 -> attn2.to_q (also for the shared half of a big batch, which runs the separate launches on Ms rows)
    want_q2 = LNX and (not big or (shared_half and Ms < CHAIN_MIN_ROWS)) and w.has_cross and w.lnx_q2 is not None
    st = None
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
        if ctx_img is not None:   # short context: the per-context fragment image (emit_ctx_kv), scores in registers
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


def emit_transformer(plan, w, x, geo, ctx_kv, L, heads, out, groups=32, shared_half=False, ctx_img=None, out_gn=None):
    """Transformer3DModel.forward + BasicTransformerBlock.forward (src/models/attention.py:318-365,479-526).
    ctx_kv: Rows [n_img*L][2C] = [K | V] projections of the context for this site (computed per context).
    shared_half: see emit_basic_block (x holds identical halves; both halves of `out` are still written in full)."""
    g, C = geo, w.C
    n_s, M_s = (g.n_img // 2, g.M // 2) if shared_half else (g.n_img, g.M)
    a = plan.rows("norm", g.M, C)
    z, tok = ffz_rows(plan, w, g.M, C, x, out)
    pre, tok_stat = None, None
    if w.ch_in_qkv is not None and not shared_half and g.M >= CHAIN_MIN_ROWS:
        if CHAIN_GN and g.hw % 16 == 0 and g.hw >= 160:   # the norm's apply rides too: only its statistics are launched
            pre = (x, w.proj_in_b, emit_groupnorm_stats(plan, x, n_s, g.hw, w.gn_g, w.gn_b, 1e-6, groups))
        else:
            emit_groupnorm(plan, x, n_s, g.hw, w.gn_g, w.gn_b, 1e-6, False, a, groups)
            pre = (a, w.proj_in_b, None)       # proj_in rides with norm1 + qkv (emit_basic_block)
    else:
        emit_groupnorm(plan, x.rows(0, M_s), n_s, g.hw, w.gn_g, w.gn_b, 1e-6, False, a.rows(0, M_s), groups)
        tok_stat = emit_gemm(plan, a.rows(0, M_s), w.proj_in, C, C, tok.rows(0, M_s), bias=w.proj_in_b,
                             stat=LNX and M_s < CHAIN_MIN_ROWS)   # (M_s: the shared half of a big batch qualifies too)
    post = (w.ch_o2_ffz, x, w.proj_out_b, out) if (getattr(w, "ch_o2_ffz", None) is not None and w.has_cross and
                                                   g.M >= CHAIN_MIN_ROWS) else None
    emit_basic_block(plan, w, tok, g.n_img, g.hw, heads, a, ctx_kv, L, shared_half=shared_half, ctx_img=ctx_img, pre=pre,
                     post=post, tok_stat=tok_stat, z=z, out_gn=out_gn)
    if post is None and z is None:
        emit_gemm(plan, tok, w.proj_out, C, C, out, bias=w.proj_out_b, residual=x, gn=out_gn)


def emit_ctx_kv(plan, w, ctx16, ctx_kv, n_seq=0, L=0, heads=0):
    """[K | V] = ctx [to_k; to_v]^T  (CrossAttention.forward attention.py:139-141) — context only.  With n_seq / L /
    heads given and L <= XATTN_MAX_KEYS, also the fragment image rcdm_xattn reads (returned; else None)."""
    emit_gemm(plan, ctx16, w.kv2, 2 * w.C, w.ctx_dim, ctx_kv, bias=getattr(w, "kv2_b", None))
    if heads and 0 < L <= XATTN_MAX_KEYS and os.environ.get("RCDM_XATTN", "1") != "0":
        return emit_xattn_pack(plan, ctx_kv.cols(0, w.C), ctx_kv.cols(w.C, w.C), n_seq, heads, L, w.C // heads)
    return None


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
    elif chained and CHAIN_GN and g.hw % 16 == 0 and g.hw >= 160:
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
    want_stat = LNX and g.M < CHAIN_MIN_ROWS
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


# ------------------------------------------------------------------------------------------------
# the whole UNet

CIN_PAD = 64   # conv_in reads its 9 channels from a 64-wide zero-padded row (one BK step per tap)
COUT_PAD = 8   # conv_out writes 4 channels + 4 zero columns (16-byte rows)


class UNetProgram:
    """Static launch plan of one UNet3DConditionModel.forward for fixed (b, f, H, W, L)."""

    def __init__(self, cfg, sd, b, frames, H, W, L, device, shared_prefix=False):
        """shared_prefix: the caller guarantees that samples [0, b/2) and [b/2, b) of the input are IDENTICAL and differ
        only in their context rows (the two CFG halves of a denoising step: RCDMs_pipeline.py:481 duplicates the latents,
        mask and masked latents).  conv_in, the first ResNet block and the first transformer up to the cross-attention
        query are then evaluated once and stored to both halves — bit-identical to evaluating the half batch twice."""
        if shared_prefix and (b % 2 or cfg["down_block_types"][0] != "CrossAttnDownBlock3D"):
            raise hip.RcdmError("shared_prefix needs an even batch and a cross-attention first block")
        self.shared_prefix = bool(shared_prefix)
        if H % 8 or W % 8:
            raise hip.RcdmError(f"latent size {H}x{W} must be a multiple of 8 on the HIP path")
        self.cfg, self.b, self.f, self.H, self.W, self.L = cfg, b, frames, H, W, L
        self.device = torch.device(device)
        self.plan = plan = Plan(device)
        self.ctx_plan_ops = []
        self.stream = torch.cuda.Stream(device=self.device)
        self.graph = None
        self.calls = 0
        self.ctx_key = None
        boc = list(cfg["block_out_channels"])
        lpb = cfg["layers_per_block"]
        heads, groups, eps = cfg["attention_head_dim"], cfg["norm_num_groups"], cfg["norm_eps"]
        mheads, n_attn = cfg["motion_num_attention_heads"], cfg["motion_attention_blocks"]
        nlev = len(boc)
        geos = [Geo(b, frames, H >> l, W >> l) for l in range(nlev)]
        self.geos = geos
        pk = Packer(sd, device)
        ted = boc[0] * 4

        def has_motion(res):
            return cfg["use_motion_module"] and res in cfg["motion_module_resolutions"]

        # ---- time embedding chain (unet.py:381-389) + all time_emb_proj batched (resnet.py:191) ----
        self.t_dev = torch.zeros(b, dtype=torch.float32, device=self.device)
        temb0 = torch.zeros(b, boc[0], dtype=torch.float32, device=self.device)
        temb1 = torch.zeros(b, ted, dtype=torch.float32, device=self.device)
        emb = torch.zeros(b, ted, dtype=torch.float32, device=self.device)
        te_w1, te_b1 = pk.mat_f16("time_embedding.linear_1.weight"), pk.vec("time_embedding.linear_1.bias")
        te_w2, te_b2 = pk.mat_f16("time_embedding.linear_2.weight"), pk.vec("time_embedding.linear_2.bias")
        resnet_prefixes = []
        for i, kind in enumerate(cfg["down_block_types"]):
            resnet_prefixes += [f"down_blocks.{i}.resnets.{j}." for j in range(lpb)]
        resnet_prefixes += ["mid_block.resnets.0.", "mid_block.resnets.1."]
        for i, kind in enumerate(cfg["up_block_types"]):
            resnet_prefixes += [f"up_blocks.{i}.resnets.{j}." for j in range(lpb + 1)]
        tp_off, off = {}, 0
        for p in resnet_prefixes:
            tp_off[p] = off
            off += sd[p + "time_emb_proj.weight"].shape[0]
        tp_total = off
        tp_w = pk.mat_f16(*[p + "time_emb_proj.weight" for p in resnet_prefixes])
        tp_b = torch.cat([pk.vec(p + "time_emb_proj.bias") for p in resnet_prefixes]).contiguous()
        tproj = torch.zeros(b, tp_total, dtype=torch.float32, device=self.device)
        plan.keep += [self.t_dev, temb0, temb1, emb, tproj, te_w1, te_b1, te_w2, te_b2, tp_w, tp_b]

        def small_linear_rows(x, K, Wm, bias, N, si, so, outt):
            for r0 in range(0, b, 8):
                r = min(8, b - r0)
                plan.add(lambda x=x, r0=r0, r=r: hip.small_linear(
                    x.data_ptr() + 4 * r0 * K, r, K, Wm.data_ptr(), bias.data_ptr(), N, si, so,
                    outt.data_ptr() + 4 * r0 * N))
                plan.n_launch += 1

        plan.add(lambda: hip.timestep_embed(self.t_dev.data_ptr(), b, boc[0], temb0.data_ptr()))
        plan.n_launch += 1
        small_linear_rows(temb0, boc[0], te_w1, te_b1, ted, 0, 1, temb1)
        small_linear_rows(temb1, ted, te_w2, te_b2, ted, 0, 0, emb)
        small_linear_rows(emb, ted, tp_w, tp_b, tp_total, 1, 0, tproj)
        # everything above depends on the timestep only: a sampling loop evaluates it for all T steps up front
        # (time_table) and replays the body from here with one table-row copy in front (sampler.DenoiseLoop)
        self.n_time_ops = len(plan.ops)
        self.tproj = tproj

        def temb_of(p):
            return (tproj, tp_off[p], tp_total)

        # ---- skip/concat layout: simulate the up path to learn each concat buffer's width ----------
        skip_specs = [(boc[0], 0)]
        for i in range(nlev):
            skip_specs += [(boc[i], i)] * lpb
            if i != nlev - 1:
                skip_specs.append((boc[i], i + 1))
        n_skip = len(skip_specs)
        rev = list(reversed(boc))
        h_ch, k = boc[-1], n_skip
        cat_hch = {}
        for i in range(nlev):
            for j in range(lpb + 1):
                k -= 1
                cat_hch[k] = h_ch
                h_ch = rev[i]
        assert k == 0
        cats = {}
        for k, (c, lvl) in enumerate(skip_specs):
            width = cat_hch[k] + c
            cats[k] = plan.rows(f"cat{k}", geos[lvl].M, width, unique=True)

        def skip_view(k):
            return cats[k].cols(cat_hch[k], skip_specs[k][0])

        def h_view(k):
            return cats[k].cols(0, cat_hch[k])

        # ---- input / conv_in ------------------------------------------------------------------------
        g0 = geos[0]
        self.x_in = plan.rows("x_in", g0.M, CIN_PAD, unique=True)
        conv_in_w = pk.conv3x3("conv_in.weight", cin_pad=CIN_PAD)
        self.in_channels = sd["conv_in.weight"].shape[1]
        g0h = Geo(b // 2, frames, H, W) if shared_prefix else None
        if shared_prefix:
            emit_conv3x3(plan, self.x_in.rows(0, g0h.M), g0h.n_img, g0.H, g0.W, conv_in_w, CIN_PAD, boc[0],
                         skip_view(0).rows(0, g0h.M), bias=pk.vec("conv_in.bias"), dup_rows=g0h.M)
        else:
            emit_conv3x3(plan, self.x_in, g0.n_img, g0.H, g0.W, conv_in_w, CIN_PAD, boc[0], skip_view(0),
                         bias=pk.vec("conv_in.bias"))

        # ---- cross-attention context: per-site [K|V] buffers, filled by the context plan -------------
        ctx_dim = cfg["cross_attention_dim"]
        self.ctx16 = plan.rows("ctx16", g0.n_img * L, ctx_dim, unique=True)
        ctx_plan = Plan(device)
        ctx_plan.bufs = plan.bufs  # share buffers (split-K workspace) and materialisation
        self._ctx_plan = ctx_plan
        site = [0]

        def transformer(p, x, geo, out, shared_half=False, out_gn=None):
            small = geo.M < CHAIN_MIN_ROWS   # below the chain kernels' row count: deferred LayerNorms + composed proj_out
            w = pack_transformer(pk, p, lnx=small or shared_half, ffz=small)
            kv = plan.rows(f"ctx_kv{site[0]}", geo.n_img * L, 2 * w.C, unique=True)
            site[0] += 1
            img = emit_ctx_kv(ctx_plan, w, self.ctx16, kv, geo.n_img, L, heads)
            emit_transformer(plan, w, x, geo, kv, L, heads, out, groups, shared_half=shared_half, ctx_img=img, out_gn=out_gn)

        def motion(p, x, geo, out, out_gn=None):
            emit_motion(plan, pack_motion(pk, p, n_attn, lnx=geo.M < CHAIN_MIN_ROWS), x, geo, mheads, out, groups, out_gn=out_gn)

        def layer(pb, j, kind_attn, res, x, geo, final_out, shared=False, next_resnet=False):
            """resnet -> [transformer] -> [motion]; the LAST op writes final_out, the others ping-pong.
            shared: the first layer under shared_prefix — the ResNet block runs on the first half of the batch (its
            GroupNorm statistics are per sample, its time-embedding row per sample: nothing crosses the halves).
            next_resnet: final_out is, as it stands, the input of another ResNet block of this geometry (down path: the next
            layer of the block) — its norm1 is then the op emitted right after this layer's last one."""
            stages = ["r"] + (["t"] if kind_attn else []) + (["m"] if has_motion(res) else [])
            cur = x
            cout = sd[pb + f"resnets.{j}.conv1.weight"].shape[0]
            for si, st in enumerate(stages):
                dst = final_out if si == len(stages) - 1 else plan.rows(f"blk{si % 2}", geo.M, cout)
                # the GroupNorm that reads dst as the very next op: per frame in front of a transformer / motion module
                # (attention.py:328, motion_module.py:162), across the frames in front of a ResNet block (resnet.py:185)
                if si + 1 < len(stages):
                    nxt = (geo.n_img, geo.hw, groups)
                else:
                    nxt = (geo.b, geo.f * geo.hw, groups) if next_resnet else None
                if si + 1 < len(stages) and (shared or geo.M >= CHAIN_MIN_ROWS):
                    nxt = None   # (the shared-prefix transformer norms half the rows; at the chain kernels' row count the
                                 #  norm in front of a transformer / motion module is a statistics-only launch)
                if st == "r":
                    pr = pb + f"resnets.{j}."
                    if shared:
                        emit_resnet(plan, pack_resnet(pk, pr), cur.rows(0, g0h.M), g0h, temb_of(pr), dst.rows(0, g0h.M),
                                    eps, groups, dup_rows=g0h.M)
                    else:
                        emit_resnet(plan, pack_resnet(pk, pr), cur, geo, temb_of(pr), dst, eps, groups, out_gn=nxt)
                elif st == "t":
                    transformer(pb + f"attentions.{j}.", cur, geo, dst, shared_half=shared, out_gn=nxt)
                else:
                    motion(pb + f"motion_modules.{j}.", cur, geo, dst, out_gn=nxt)
                cur = dst
            return cur

        # ---- down path ------------------------------------------------------------------------------
        cur, k = skip_view(0), 1
        for i, kind in enumerate(cfg["down_block_types"]):
            pb = f"down_blocks.{i}."
            for j in range(lpb):
                cur = layer(pb, j, kind == "CrossAttnDownBlock3D", 2 ** i, cur, geos[i], skip_view(k),
                            shared=shared_prefix and i == 0 and j == 0, next_resnet=j + 1 < lpb)
                k += 1
            if i != nlev - 1:
                dsw = pk.conv3x3(pb + "downsamplers.0.conv.weight")
                emit_conv3x3(plan, cur, geos[i].n_img, geos[i].H, geos[i].W, dsw, boc[i], boc[i], skip_view(k),
                             stride=2, bias=pk.vec(pb + "downsamplers.0.conv.bias"))
                cur = skip_view(k)
                k += 1
        assert k == n_skip

        # ---- mid block (unet_blocks.py:272-280) ------------------------------------------------------
        gm = geos[-1]
        m0 = plan.rows("blk0", gm.M, boc[-1])
        emit_resnet(plan, pack_resnet(pk, "mid_block.resnets.0."), cur, gm, temb_of("mid_block.resnets.0."), m0,
                    eps, groups, 1.0 / cfg.get("mid_block_scale_factor", 1), out_gn=(gm.n_img, gm.hw, groups))
        m1 = plan.rows("blk1", gm.M, boc[-1])
        transformer("mid_block.attentions.0.", m0, gm, m1)
        cur = m1
        if cfg["use_motion_module"] and cfg["motion_module_mid_block"]:
            m2 = plan.rows("blk0", gm.M, boc[-1])
            motion("mid_block.motion_modules.0.", m1, gm, m2)
            cur = m2
        k = n_skip - 1
        emit_resnet(plan, pack_resnet(pk, "mid_block.resnets.1."), cur, gm, temb_of("mid_block.resnets.1."),
                    h_view(k), eps, groups, 1.0 / cfg.get("mid_block_scale_factor", 1))

        # ---- up path ---------------------------------------------------------------------------------
        final = plan.rows("final", g0.M, boc[0], unique=True)
        for i, kind in enumerate(cfg["up_block_types"]):
            pb = f"up_blocks.{i}."
            lvl = nlev - 1 - i
            geo = geos[lvl]
            last_block = i == nlev - 1
            for j in range(lpb + 1):
                x = cats[k]  # [h | skip] full-width view
                last_layer = j == lpb
                if not last_layer:
                    dst = h_view(k - 1)
                elif last_block:
                    dst = final
                else:
                    dst = plan.rows("up_tmp", geo.M, rev[i])
                cur = layer(pb, j, kind == "CrossAttnUpBlock3D", 2 ** (nlev - 1 - i), x, geo, dst)
                k -= 1
            if not last_block:
                emit_upsample_conv(plan, pk, pb + "upsamplers.0.conv.weight", cur, geo.n_img, geo.H, geo.W, rev[i], h_view(k),
                                   pk.vec(pb + "upsamplers.0.conv.bias"))
        assert k == -1

        # ---- output head (unet.py:455-457) -------------------------------------------------------------
        a = plan.rows("norm", g0.M, boc[0])
        emit_groupnorm(plan, final, b, frames * g0.hw, pk.vec("conv_norm_out.weight"), pk.vec("conv_norm_out.bias"),
                       eps, True, a, groups)
        self.out_channels = sd["conv_out.weight"].shape[0]
        co_w = pk.conv3x3("conv_out.weight", cout_pad=COUT_PAD)
        co_b = torch.cat([pk.vec("conv_out.bias"),
                          torch.zeros(COUT_PAD - self.out_channels, device=self.device)]).contiguous()
        self.eps_out = plan.rows("eps_out", g0.M, COUT_PAD, unique=True)
        emit_conv3x3(plan, a, g0.n_img, g0.H, g0.W, co_w, boc[0], COUT_PAD, self.eps_out, bias=co_b)

        pk.done()
        plan.materialize()
        self.n_sites = site[0]

    # ---- context ---------------------------------------------------------------------------------
    def set_context(self, ctx, force=False):
        """ctx (b*f, L, D) any float dtype/device.  Recomputes the 16 [K|V] projections only if the
        context changed (the reference recomputes them every step, attention.py:139-141)."""
        # The cache key is the caller's tensor OBJECT (held strongly, so its storage cannot be recycled for another
        # context while it is the key) plus its version counter; an address/_version pair alone identifies a transient
        # allocation, not its contents.  Tensors without a version counter (inference mode) are never cached.
        try:
            ver = ctx._version
        except RuntimeError:
            ver = None
        if not force and ver is not None and self.ctx_key is not None and self.ctx_key[0] is ctx and self.ctx_key[1] == ver:
            return
        n_img = self.b * self.f
        if tuple(ctx.shape) != (n_img, self.L, self.cfg["cross_attention_dim"]):
            raise hip.RcdmError(f"encoder_hidden_states shape {tuple(ctx.shape)} != "
                                f"{(n_img, self.L, self.cfg['cross_attention_dim'])}")
        src = ctx.detach().to(self.device, torch.float32).contiguous()
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            hip.pack_f16(src.data_ptr(), self.ctx16.ptr, src.numel())
            self._ctx_plan.run()
        cur.wait_stream(self.stream)
        src.record_stream(self.stream)
        self.ctx_key = (ctx, ver) if ver is not None else None

    # ---- execution -------------------------------------------------------------------------------
    def run_body(self, skip_time=False):
        """Enqueue the UNet body on torch's current stream (inputs: x_in rows, t_dev; output: eps_out).  skip_time: the
        time-embedding chain is left out — `tproj` already holds this step's time_emb_proj rows."""
        self.plan.run(self.plan.ops[self.n_time_ops:] if skip_time else None)

    def time_table(self, timesteps):
        """[T][b * tp_total] fp32: the time_emb_proj rows of all resnets (unet.py:381-389, resnet.py:191) for each of
        the given timesteps, every batch row at the same timestep (what the sampling loop feeds, RCDMs_pipeline.py:483)."""
        rows = []
        with torch.cuda.stream(self.stream):
            for t in timesteps:
                self.t_dev.fill_(float(t))
                self.plan.run(self.plan.ops[:self.n_time_ops])
                rows.append(self.tproj.reshape(-1).clone())
            table = torch.stack(rows).contiguous()
        self.stream.synchronize()
        return table

    def capture(self, pre=None, post=None, skip_time=False):
        """Capture [pre ops] + body + [post ops] into a hipGraph on the program's stream."""
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(self.stream):
            g = hip.Graph()
            g.begin()
            try:
                for op in (pre or []):
                    op()
                self.run_body(skip_time)
                for op in (post or []):
                    op()
            finally:
                g.end()
        torch.cuda.synchronize(self.device)
        return g

    def forward(self, sample, timestep, ctx, use_graph=True):
        """UNet3DConditionModel.forward semantics: sample (b,Cin,f,H,W) -> (b,Cout,f,H,W) fp32."""
        b, f, H, W = self.b, self.f, self.H, self.W
        self.set_context(ctx)
        x = sample.detach().to(self.device, torch.float32).contiguous()
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        out = torch.empty(b, self.out_channels, f, H, W, dtype=torch.float32, device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.t_dev.copy_(t.expand(b))
            hip.ncfhw_to_rows(x.data_ptr(), b, self.in_channels, f, H, W, self.x_in.ptr, self.x_in.ld, CIN_PAD)
            if use_graph and self.calls >= 1:
                if self.graph is None:
                    # capture() synchronises; the first (eager) call has already warmed every kernel up
                    self.graph = self.capture()
                self.graph.launch()
            else:
                self.plan.run()
            self.calls += 1
            hip.rows_to_ncfhw(self.eps_out.ptr, self.eps_out.ld, b, self.out_channels, f, H, W, out.data_ptr())
        cur.wait_stream(self.stream)
        x.record_stream(self.stream)
        out.record_stream(self.stream)
        return out


# ------------------------------------------------------------------------------------------------
# block-level runners (module.forward of the mirrored classes): pack, plan, run eagerly.

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
        w = pack_resnet(pk, "")
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
