"""UNetProgram: the static launch plan of ONE UNet3DConditionModel.forward (src/models/unet.py:322-463, block wiring of
src/models/unet_blocks.py) for a fixed geometry (b, f, H, W, L) — time-embedding chain, conv_in, the down / mid / up
paths with the skip tensors written by their producers into the concat buffers their consumers read, the output head —
plus the per-context plan (the 16 [K | V] projections of the cross-attention context, computed once per context where
the reference recomputes them every step, attention.py:139-141), graph capture and the forward() entry."""
import torch

from . import hip
from . import switches as SW
from .emit_blocks import (CHAIN_MIN_ROWS, emit_ctx_kv, emit_motion, emit_rank1_ctx, emit_resnet, emit_transformer, wino_level,
                          full_rank_runs)
from .emit_ops import XATTN_MAX_KEYS, emit_conv3x3, emit_gemm, emit_groupnorm, emit_upsample_conv
from .packer import Packer, pack_motion, pack_resnet, pack_transformer
from .plan import Geo, Plan

CIN_PAD = 64   # conv_in reads its 9 channels from a 64-wide zero-padded row (one BK step per tap)
COUT_PAD = 8   # conv_out writes 4 channels + 4 zero columns (16-byte rows)


class UNetProgram:
    """Static launch plan of one UNet3DConditionModel.forward for fixed (b, f, H, W, L)."""

    def __init__(self, cfg, sd, b, frames, H, W, L, device, shared_prefix=False, rank1_runs=None):
        """shared_prefix: the caller guarantees that samples [0, b/2) and [b/2, b) of the input are IDENTICAL and differ
        only in their context rows (the two CFG halves of a denoising step: RCDMs_pipeline.py:481 duplicates the latents,
        mask and masked latents).  conv_in, the first ResNet block and the first transformer up to the cross-attention
        query are then evaluated once and stored to both halves — bit-identical to evaluating the half batch twice."""
        # rank1_runs: None, or the maximal runs ((i0, i1), ...) of images whose context rows are NOT all equal (full_rank_runs);
        # the caller guarantees — and set_context checks — that every image outside them has L identical context rows (SURVEY
        # F6: the unseen frames' rows, RCDMs_pipeline.py:447-450): their cross-attention is query-independent and collapses to
        # one row per (image, site), see emit_basic_block.  Selected per context the way shared_prefix is selected per story.
        self.rank1_runs = None
        if rank1_runs is not None:
            runs = tuple((int(i0), int(i1)) for i0, i1 in rank1_runs)
            if any(not (0 <= i0 < i1 <= b * frames) for i0, i1 in runs) or any(a[1] >= c[0] for a, c in zip(runs, runs[1:])):
                raise hip.RcdmError(f"rank1_runs {runs}: not disjoint ascending runs of the {b * frames} images")
            if sum(i1 - i0 for i0, i1 in runs) < b * frames and 0 < L <= XATTN_MAX_KEYS:
                self.rank1_runs = runs      # (every image full rank, or a context too long for rcdm_xattn: the general plan)
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

        def resnet_w(p, geo):
            shp = sd[p + "conv1.weight"].shape
            return pack_resnet(pk, p, wino=wino_level(geo, shp[1], shp[0]))

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
            r1 = None
            if self.rank1_runs is not None and not (shared_half and small):
                # (below the chain kernels' row count the shared-prefix transformer stores one half's rows to both halves:
                #  a per-image row cannot ride in that epilogue — this one site then keeps the general form)
                emit_ctx_kv(ctx_plan, w, self.ctx16, kv)
                r1 = emit_rank1_ctx(ctx_plan, plan, w, kv, geo, L, heads, self.rank1_runs, site[0] - 1)
                img = None
            else:
                img = emit_ctx_kv(ctx_plan, w, self.ctx16, kv, geo.n_img, L, heads)
            emit_transformer(plan, w, x, geo, kv, L, heads, out, groups, shared_half=shared_half, ctx_img=img, out_gn=out_gn,
                             rank1=r1)

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
                        emit_resnet(plan, resnet_w(pr, geo), cur, geo, temb_of(pr), dst, eps, groups, out_gn=nxt)
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
        # (the statistics hand-off to the norm behind a producer is gated as in layer(): at the chain kernels' row count the
        #  norm in front of a transformer / motion module is a statistics-only launch that takes no hand-off)
        small_mid = gm.M < CHAIN_MIN_ROWS
        mid_motion = bool(cfg["use_motion_module"] and cfg["motion_module_mid_block"])
        per_frame, cross_frame = (gm.n_img, gm.hw, groups), (gm.b, gm.f * gm.hw, groups)
        emit_resnet(plan, resnet_w("mid_block.resnets.0.", gm), cur, gm, temb_of("mid_block.resnets.0."), m0,
                    eps, groups, 1.0 / cfg.get("mid_block_scale_factor", 1), out_gn=per_frame if small_mid else None)
        m1 = plan.rows("blk1", gm.M, boc[-1])
        transformer("mid_block.attentions.0.", m0, gm, m1,
                    out_gn=(per_frame if mid_motion else cross_frame) if small_mid else None)
        cur = m1
        if mid_motion:
            m2 = plan.rows("blk0", gm.M, boc[-1])
            motion("mid_block.motion_modules.0.", m1, gm, m2, out_gn=cross_frame if small_mid else None)
            cur = m2
        k = n_skip - 1
        emit_resnet(plan, resnet_w("mid_block.resnets.1.", gm), cur, gm, temb_of("mid_block.resnets.1."),
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
        co_b = torch.cat([pk.vec("conv_out.bias"),
                          torch.zeros(COUT_PAD - self.out_channels, device=self.device)]).contiguous()
        self.eps_out = plan.rows("eps_out", g0.M, COUT_PAD, unique=True)
        if SW.OUT_TAPS and boc[0] % 8 == 0 and g0.M * 9 * COUT_PAD * 2 < (1 << 31):
            # 4 (padded 8) output channels waste 7/8 of the narrowest conv tile: nine tap planes of the pixels themselves as ONE
            # GEMM with N = 72, then the gather that sums every pixel's nine neighbours' planes (rcdm_conv_taps_gather)
            co_w9 = pk.conv3x3_taps("conv_out.weight", cout_pad=COUT_PAD)
            P = plan.rows("out_taps", g0.M, 9 * COUT_PAD)
            emit_gemm(plan, a, co_w9, 9 * COUT_PAD, boc[0], P)
            eps_out = self.eps_out

            def op_gather():
                hip.conv_taps_gather(P.ptr, P.ld, g0.n_img, g0.H, g0.W, COUT_PAD, 0, co_b.data_ptr(), eps_out.ptr, eps_out.ld)
            plan.add(op_gather, f"conv_gather {g0.n_img}x{g0.H}x{g0.W} C={COUT_PAD}")
            plan.keep += [co_b]
            plan.n_launch += 1
        else:
            co_w = pk.conv3x3("conv_out.weight", cout_pad=COUT_PAD)
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
        if self.rank1_runs is not None:
            inside = lambda i: any(i0 <= i < i1 for i0, i1 in self.rank1_runs)
            bad = [i for i0, i1 in full_rank_runs(src) for i in range(i0, i1) if not inside(i)]
            if bad:
                raise hip.RcdmError(f"this launch plan was built for contexts whose images outside {self.rank1_runs} have L "
                                    f"identical rows; images {bad} of this context do not (select the plan with full_rank_runs)")
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
