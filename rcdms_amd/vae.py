"""VAE of the stage-2 pipeline on the HIP path: decoder (`AutoencoderKLDecoder`) and, with the encoder
added, the whole module (`AutoencoderKL`).

Replaces `self.vae.decode(latents[i:i+1]).sample`, called once per frame at src/pipelines/RCDMs_pipeline.py:274-287 on
the SD-1.5 `AutoencoderKL` (diffusers==0.24.0, third party — restated from the published architecture, PARITY UNPINNED:
see oracle/vae_oracle.py).  `AutoencoderKLDecoder` holds the decoder-side parameters under the diffusers 0.24 key names
(`post_quant_conv.*`, `decoder.*`), so `load_state_dict(vae_state_dict, strict=False)` takes a real checkpoint, and
decodes all frames in one batch on the stage-2 kernels: conv3x3 implicit GEMM (nearest-2x upsampling folded into the
conv's input indexing), per-image GroupNorm(+SiLU), 1x1 shortcuts as GEMMs.  The mid-block attention has ONE head of
512 channels — too wide for the flash kernel — so it runs as scores = Q K^T (GEMM, scaled in the epilogue), row softmax
(rcdm_softmax_rows), out = P V (GEMM against V^T, which a GEMM with swapped operands produces directly); the value
bias is folded into the output projection (softmax rows sum to one).

`AutoencoderKL` adds `encoder.*` / `quant_conv.*` and `encode(x).latent_dist` (RCDMs_pipeline.py:429, one call per story on
the masked source frames): the same kernels, with diffusers' Downsample2D(padding=0) — F.pad (0,1,0,1) then a stride-2
conv — as the conv kernel's pad-after-only form (rcdm_conv3x3_desc.pad_after_only).  No CPU path."""
import torch
from torch import nn

from . import hip
from .engine import (CIN_PAD, COUT_PAD, Geo, Packer, Plan, Rows, _NS, emit_conv3x3, emit_gemm, emit_groupnorm,
                     emit_upsample_conv)

SD15_VAE = dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, out_channels=3,
                in_channels=3, norm_num_groups=32, scaling_factor=0.18215)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1, self.conv1 = nn.GroupNorm(groups, cin, eps=1e-6), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = nn.GroupNorm(groups, cout, eps=1e-6), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)


class _Attn(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])
        self.attentions = nn.ModuleList([_Attn(c, groups)])


class _Up(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(n)])
        if upsample:
            up = nn.Module()
            up.conv = nn.Conv2d(cout, cout, 3, padding=1)
            self.upsamplers = nn.ModuleList([up])


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = list(cfg["block_out_channels"]), cfg["norm_num_groups"]
        rev = boc[::-1]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], g)
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_Up(prev, c, cfg["layers_per_block"] + 1, g, i < len(rev) - 1))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cfg["out_channels"], 3, padding=1)


class _Down(nn.Module):
    def __init__(self, cin, cout, n, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(n)])
        if downsample:
            dn = nn.Module()
            dn.conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)
            self.downsamplers = nn.ModuleList([dn])


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = list(cfg["block_out_channels"]), cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        downs, prev = [], boc[0]
        for i, c in enumerate(boc):
            downs.append(_Down(prev, c, cfg["layers_per_block"], g, i < len(boc) - 1))
            prev = c
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _Mid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg["latent_channels"], 3, padding=1)


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKLDecoder(nn.Module):
    """Decoder half of diffusers' AutoencoderKL: `decode(z) -> .sample` for z (n, 4, h, w), any n (frames batched)."""

    def __init__(self, **cfg):
        super().__init__()
        self.cfg = dict(SD15_VAE)
        self.cfg.update(cfg)
        self.post_quant_conv = nn.Conv2d(self.cfg["latent_channels"], self.cfg["latent_channels"], 1)
        self.decoder = _Decoder(self.cfg)
        self._programs = {}

    @property
    def device(self):
        return self.post_quant_conv.weight.device

    @property
    def dtype(self):
        return self.post_quant_conv.weight.dtype

    def encode(self, x):
        raise NotImplementedError("AutoencoderKLDecoder holds the decoder half only; keep the torch AutoencoderKL for "
                                  "`encode` (one call per story, RCDMs_pipeline.py:429)")

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        if self.device.type != "cuda":
            raise hip.RcdmError(f"AutoencoderKLDecoder runs on the HIP path only (module is on {self.device})")
        n, c, h, w = z.shape
        key = (n, h, w, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        prog = self._programs.get((n, h, w))
        if prog is None or prog[0] != key:
            prog = (key, VaeDecodeProgram(self.cfg, self.state_dict(), n, h, w, self.device))
            self._programs[(n, h, w)] = prog
        out = prog[1].forward(z).to(z.dtype)
        return _Sample(out) if return_dict else (out,)


class DiagonalGaussianDistribution:
    """`encode(x).latent_dist`: mean / logvar (clamped to [-30, 20]) of the posterior, as diffusers' class of this name."""

    def __init__(self, mean, logvar):
        self.mean, self.logvar = mean, logvar
        self.std = torch.exp(0.5 * logvar)
        self.var = torch.exp(logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class _Posterior:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKL(AutoencoderKLDecoder):
    """Both halves under the diffusers 0.24 key names (248 tensors at the SD-1.5 shape): a drop-in for the pipeline's
    `vae` argument — `encode(x).latent_dist.sample(generator)`, `decode(z).sample`, `config.block_out_channels` /
    `config.scaling_factor`."""

    def __init__(self, **cfg):
        super().__init__(**cfg)
        lc = self.cfg["latent_channels"]
        self.encoder = _Encoder(self.cfg)
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.config = _NS(**self.cfg)
        self._enc_programs = {}

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        if self.device.type != "cuda":
            raise hip.RcdmError(f"AutoencoderKL runs on the HIP path only (module is on {self.device})")
        n, c, h, w = x.shape
        key = (n, h, w, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        prog = self._enc_programs.get((n, h, w))
        if prog is None or prog[0] != key:
            prog = (key, VaeEncodeProgram(self.cfg, self.state_dict(), n, h, w, self.device))
            self._enc_programs[(n, h, w)] = prog
        mean, logvar = prog[1].forward(x)
        dist = DiagonalGaussianDistribution(mean.to(x.dtype), logvar.to(x.dtype))
        return _Posterior(dist) if return_dict else (dist,)


def _pack_resnet(pk, p):
    w = _NS(cin=pk.sd[p + "conv1.weight"].shape[1], cout=pk.sd[p + "conv1.weight"].shape[0])
    w.g1, w.b1 = pk.vec(p + "norm1.weight"), pk.vec(p + "norm1.bias")
    w.g2, w.b2 = pk.vec(p + "norm2.weight"), pk.vec(p + "norm2.bias")
    w.conv1, w.cb1 = pk.conv3x3(p + "conv1.weight"), pk.vec(p + "conv1.bias")
    w.conv2, w.cb2 = pk.conv3x3(p + "conv2.weight"), pk.vec(p + "conv2.bias")
    w.shortcut = None
    if pk.has(p + "conv_shortcut.weight"):
        w.shortcut, w.sb = pk.mat_f16(p + "conv_shortcut.weight"), pk.vec(p + "conv_shortcut.bias")
    return w


def _emit_resnet2d(plan, w, x, geo, out, groups):
    """diffusers ResnetBlock2D without time embedding: GroupNorm statistics per image, eps 1e-6."""
    g = geo
    a1 = plan.rows("norm", g.M, w.cin)
    emit_groupnorm(plan, x, g.n_img, g.hw, w.g1, w.b1, 1e-6, True, a1, groups)
    h1 = plan.rows("res_h1", g.M, w.cout)
    emit_conv3x3(plan, a1, g.n_img, g.H, g.W, w.conv1, w.cin, w.cout, h1, bias=w.cb1)
    a2 = plan.rows("norm", g.M, w.cout)
    emit_groupnorm(plan, h1, g.n_img, g.hw, w.g2, w.b2, 1e-6, True, a2, groups)
    res = x
    if w.shortcut is not None:
        res = plan.rows("res_sc", g.M, w.cout)
        emit_gemm(plan, x, w.shortcut, w.cout, w.cin, res, bias=w.sb)
    emit_conv3x3(plan, a2, g.n_img, g.H, g.W, w.conv2, w.cout, w.cout, out, bias=w.cb2, residual=res)


class VaeDecodeProgram:
    """Static launch plan of AutoencoderKL.decode for n images of h x w latents."""

    def __init__(self, cfg, sd, n, h, w, device):
        hip.load()
        self.cfg, self.n, self.h, self.w = cfg, n, h, w
        self.device = torch.device(device)
        groups = cfg["norm_num_groups"]
        lc = cfg["latent_channels"]
        rev = list(cfg["block_out_channels"])[::-1]
        if lc > 8 or cfg["out_channels"] > COUT_PAD:
            raise NotImplementedError("latent_channels <= 8 and out_channels <= 8")
        pk = Packer(sd, self.device)
        self.plan = plan = Plan(self.device)
        g0 = Geo(n, 1, h, w)
        # post_quant_conv (1x1, lc -> lc) as an 8 x 8 GEMM whose output lands in the zero-padded 64-channel row conv_in reads
        pq_w = torch.zeros(8, 8, device=self.device)
        pq_w[:lc, :lc] = pk.f32("post_quant_conv.weight").reshape(lc, lc)
        self.pq_w = pq_w.to(torch.float16).contiguous()
        self.pq_b = torch.cat([pk.vec("post_quant_conv.bias"), torch.zeros(8 - lc, device=self.device)]).contiguous()
        self.z_rows = plan.rows("vae_z", g0.M, 8, unique=True)
        self.x_in = plan.rows("vae_x_in", g0.M, CIN_PAD, unique=True)
        emit_gemm(plan, self.z_rows, self.pq_w, 8, 8, self.x_in.cols(0, 8), bias=self.pq_b)
        top = rev[0]
        cur = plan.rows("vae_a", g0.M, top, unique=True)
        emit_conv3x3(plan, self.x_in, n, h, w, pk.conv3x3("decoder.conv_in.weight", cin_pad=CIN_PAD), CIN_PAD, top, cur,
                     bias=pk.vec("decoder.conv_in.bias"))
        nxt = plan.rows("vae_b", g0.M, top, unique=True)
        cur = _emit_mid_block(plan, pk, "decoder.mid_block.", cur, nxt, g0, groups)
        geo = g0
        idx = 0
        for i, c in enumerate(rev):
            for j in range(cfg["layers_per_block"] + 1):
                out = plan.rows(f"vae_s{idx}", geo.M, c, unique=True)
                idx += 1
                _emit_resnet2d(plan, _pack_resnet(pk, f"decoder.up_blocks.{i}.resnets.{j}."), cur, geo, out, groups)
                cur = out
            if i < len(rev) - 1:
                up = Geo(n, 1, geo.H * 2, geo.W * 2)
                out = plan.rows(f"vae_s{idx}", up.M, c, unique=True)
                idx += 1
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv."
                emit_upsample_conv(plan, pk, p + "weight", cur, n, geo.H, geo.W, c, out, pk.vec(p + "bias"))
                cur, geo = out, up
        a = plan.rows("norm", geo.M, rev[-1])
        emit_groupnorm(plan, cur, n, geo.hw, pk.vec("decoder.conv_norm_out.weight"), pk.vec("decoder.conv_norm_out.bias"),
                       1e-6, True, a, groups)
        oc = cfg["out_channels"]
        co_w = pk.conv3x3("decoder.conv_out.weight", cout_pad=COUT_PAD)
        co_b = torch.cat([pk.vec("decoder.conv_out.bias"), torch.zeros(COUT_PAD - oc, device=self.device)]).contiguous()
        self.out_rows = plan.rows("vae_out", geo.M, COUT_PAD, unique=True)
        emit_conv3x3(plan, a, n, geo.H, geo.W, co_w, rev[-1], COUT_PAD, self.out_rows, bias=co_b)
        self.out_geo = geo
        pk.done()
        plan.materialize()
        self.stream = torch.cuda.Stream(device=self.device)

    @torch.no_grad()
    def forward(self, z):
        n, c, h, w = z.shape
        assert (n, h, w) == (self.n, self.h, self.w)
        cur = torch.cuda.current_stream(self.device)
        z32 = z.detach().to(self.device, torch.float32).contiguous()
        g = self.out_geo
        out = torch.empty(n, self.cfg["out_channels"], 1, g.H, g.W, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            hip.ncfhw_to_rows(z32.data_ptr(), n, c, 1, h, w, self.z_rows.ptr, self.z_rows.ld, 8)
            self.plan.run()
            hip.rows_to_ncfhw(self.out_rows.ptr, self.out_rows.ld, n, self.cfg["out_channels"], 1, g.H, g.W, out.data_ptr())
        cur.wait_stream(self.stream)
        z32.record_stream(self.stream)
        return out[:, :, 0]


class VaeEncodeProgram:
    """Static launch plan of AutoencoderKL.encode for n images of H x W pixels: -> posterior mean and clamped logvar."""

    def __init__(self, cfg, sd, n, H, W, device):
        hip.load()
        self.cfg, self.n, self.H, self.W = cfg, n, H, W
        self.device = torch.device(device)
        groups, lc = cfg["norm_num_groups"], cfg["latent_channels"]
        boc = list(cfg["block_out_channels"])
        down = 2 ** (len(boc) - 1)
        if 2 * lc != COUT_PAD or cfg["in_channels"] > 8:
            raise NotImplementedError("latent_channels == 4 and in_channels <= 8")
        if H % down or W % down:
            raise ValueError(f"image size {H}x{W} must be a multiple of {down}")
        pk = Packer(sd, self.device)
        self.plan = plan = Plan(self.device)
        geo = Geo(n, 1, H, W)
        self.x_in = plan.rows("vae_px", geo.M, CIN_PAD, unique=True)      # pixels, channels zero-padded to 64
        cur = plan.rows("vae_e0", geo.M, boc[0], unique=True)
        emit_conv3x3(plan, self.x_in, n, H, W, pk.conv3x3("encoder.conv_in.weight", cin_pad=CIN_PAD), CIN_PAD, boc[0], cur,
                     bias=pk.vec("encoder.conv_in.bias"))
        idx = 1
        for i, c in enumerate(boc):
            for j in range(cfg["layers_per_block"]):
                out = plan.rows(f"vae_e{idx}", geo.M, c, unique=True)
                idx += 1
                _emit_resnet2d(plan, _pack_resnet(pk, f"encoder.down_blocks.{i}.resnets.{j}."), cur, geo, out, groups)
                cur = out
            if i < len(boc) - 1:
                dn = Geo(n, 1, geo.H // 2, geo.W // 2)
                out = plan.rows(f"vae_e{idx}", dn.M, c, unique=True)
                idx += 1
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv."
                emit_conv3x3(plan, cur, n, geo.H, geo.W, pk.conv3x3(p + "weight"), c, c, out, stride=2, pad_after_only=1,
                             bias=pk.vec(p + "bias"))
                cur, geo = out, dn
        nxt = plan.rows("vae_eb", geo.M, boc[-1], unique=True)
        cur = _emit_mid_block(plan, pk, "encoder.mid_block.", cur, nxt, geo, groups)
        a = plan.rows("norm", geo.M, boc[-1])
        emit_groupnorm(plan, cur, n, geo.hw, pk.vec("encoder.conv_norm_out.weight"), pk.vec("encoder.conv_norm_out.bias"),
                       1e-6, True, a, groups)
        mom = plan.rows("vae_mom", geo.M, COUT_PAD, unique=True)
        emit_conv3x3(plan, a, n, geo.H, geo.W, pk.conv3x3("encoder.conv_out.weight"), boc[-1], COUT_PAD, mom,
                     bias=pk.vec("encoder.conv_out.bias"))
        self.q_w = pk.f32("quant_conv.weight").reshape(2 * lc, 2 * lc).to(torch.float16).contiguous()
        self.out_rows = plan.rows("vae_moments", geo.M, COUT_PAD, unique=True)
        emit_gemm(plan, mom, self.q_w, COUT_PAD, COUT_PAD, self.out_rows, bias=pk.vec("quant_conv.bias"))
        self.out_geo = geo
        pk.done()
        plan.materialize()
        self.stream = torch.cuda.Stream(device=self.device)

    @torch.no_grad()
    def forward(self, x):
        n, c, H, W = x.shape
        assert (n, H, W) == (self.n, self.H, self.W) and c == self.cfg["in_channels"]
        cur = torch.cuda.current_stream(self.device)
        x32 = x.detach().to(self.device, torch.float32).contiguous()
        g, lc = self.out_geo, self.cfg["latent_channels"]
        out = torch.empty(n, 2 * lc, 1, g.H, g.W, dtype=torch.float32, device=self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            hip.ncfhw_to_rows(x32.data_ptr(), n, c, 1, H, W, self.x_in.ptr, self.x_in.ld, CIN_PAD)
            self.plan.run()
            hip.rows_to_ncfhw(self.out_rows.ptr, self.out_rows.ld, n, 2 * lc, 1, g.H, g.W, out.data_ptr())
        cur.wait_stream(self.stream)
        x32.record_stream(self.stream)
        mean, logvar = out[:, :lc, 0], out[:, lc:, 0]
        return mean, logvar.clamp(-30.0, 20.0)


def _emit_mid_block(plan, pk, m, cur, nxt, geo, groups):
    """UNetMidBlock2D: resnet, one-head attention, resnet; ping-pongs between the two buffers, returns the result."""
    _emit_resnet2d(plan, _pack_resnet(pk, m + "resnets.0."), cur, geo, nxt, groups)
    cur, nxt = nxt, cur
    _emit_mid_attention(plan, pk, m + "attentions.0.", cur, geo, nxt, groups)
    cur, nxt = nxt, cur
    _emit_resnet2d(plan, _pack_resnet(pk, m + "resnets.1."), cur, geo, nxt, groups)
    return nxt


def _emit_mid_attention(plan, pk, p, x, geo, out, groups):
    C, n, hw = x.C, geo.n_img, geo.hw
    if hw > 4096 or hw % 8:
        raise NotImplementedError(f"mid-block attention over {hw} tokens: rcdm_softmax_rows holds rows of <= 4096")
    a = plan.rows("norm", geo.M, C)
    emit_groupnorm(plan, x, n, hw, pk.vec(p + "group_norm.weight"), pk.vec(p + "group_norm.bias"), 1e-6, False, a, groups)
    wq, bq = pk.mat_f16(p + "to_q.weight"), pk.vec(p + "to_q.bias")
    wk, bk = pk.mat_f16(p + "to_k.weight"), pk.vec(p + "to_k.bias")
    wv = pk.mat_f16(p + "to_v.weight")
    wo = pk.mat_f16(p + "to_out.0.weight")
    # softmax rows sum to one, so P (V0 + 1 bv^T) = P V0 + bv: the value bias moves into the output projection
    bo = (pk.vec(p + "to_out.0.bias") + hip.matmul_f32(pk.f32(p + "to_out.0.weight"), pk.vec(p + "to_v.bias"))).contiguous()
    q = plan.rows("vae_q", geo.M, C, unique=True)
    k = plan.rows("vae_k", geo.M, C, unique=True)
    emit_gemm(plan, a, wq, C, C, q, bias=bq)
    emit_gemm(plan, a, wk, C, C, k, bias=bk)
    vt = plan.rows("vae_vt", C, hw, unique=True)           # V^T of ONE image: [C][hw]
    sc = plan.rows("vae_scores", hw, hw, unique=True)      # scores / probabilities of one image
    ao = plan.rows("vae_ao", geo.M, C, unique=True)
    wv_rows = Rows(_Holder16(wv), 0, C, C, C)
    scale = float(C) ** -0.5
    for i in range(n):
        img = lambda r: Rows(r.buf, r.off + i * hw * r.ld, hw, r.C, r.ld)
        a_i = _TensorLike(img(a))
        # V^T = Wv a_i^T : a GEMM whose "activation" rows are the weight rows and whose "weights" are the image's tokens
        emit_gemm(plan, wv_rows, a_i, hw, C, vt)
        emit_gemm(plan, img(q), _TensorLike(img(k)), hw, C, sc, scale=scale)
        plan.add(lambda sc=sc: hip.softmax_rows(hw, hw, sc.ld, sc.ld, 1.0, sc.ptr, sc.ptr), f"softmax_rows M={hw} N={hw}")
        emit_gemm(plan, sc, _TensorLike(vt), C, hw, img(ao))
    emit_gemm(plan, ao, wo, C, C, out, bias=bo, residual=x)
    plan.keep += [wv]


class _Holder16:
    """Adapter: a packed f16 weight tensor seen as a plan buffer (so it can be the A operand of emit_gemm)."""

    def __init__(self, t):
        self.t = t


class _TensorLike:
    """Adapter: rows of a plan buffer seen as a weight tensor (data_ptr()), for GEMMs between two activations."""

    def __init__(self, rows):
        self.rows = rows
        if rows.ld != rows.C:
            raise ValueError("a GEMM weight operand must be dense rows")

    def data_ptr(self):
        return self.rows.ptr
