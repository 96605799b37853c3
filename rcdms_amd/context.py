"""Context builders of the stage-2 pipeline on the HIP path.

Mirrors the two classes the reference driver defines next to `inference()` — `fine_stack` (text x CLIP patch tokens,
vis_dim 1664, 257 keys) and `semantic_stack` (text x projected image embedding, vis_dim 1280, 1 key):
stage2_batchtest_rcdms_model.py:117-149, instantiated :198-199, loaded from the checkpoint's `seen_module.` /
`unseen_module.` prefixes :227-242, called once per story at src/pipelines/RCDMs_pipeline.py:447-448.  Same
constructor, same parameter names (so `load_state_dict` takes the reference tensors), same forward signature
`(vis_f, text_f) -> (k, L, 768)`.

Execution: four launches on librcdm_hip.so.  The affine maps in front of the attention are composed offline in fp32,
  Q  = text (Wq Wt)^T  + (Wq bt + bq)                 one GEMM  K = text_dim
  KV = vis  ([Wk;Wv] Wv')^T + ([Wk;Wv] bv' + [bk;bv]) one GEMM  K = vis_dim, N = 2 x 768
then rcdm_flash_attn (8 heads, d = 96, keys = vision tokens) and the out_proj GEMM.  No CPU path: a module that is not
on a GPU raises RcdmError, as the UNet does."""
import torch
from torch import nn

from . import hip


class _ContextStack(nn.Module):
    def __init__(self, text_dim, vis_dim, hidden_dim=768, num_heads=8):
        super().__init__()
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.text_fc = nn.Linear(text_dim, hidden_dim)
        self.vis_fc = nn.Linear(vis_dim, hidden_dim)
        self.multihead_attn = nn.MultiheadAttention(embed_dim=hidden_dim, num_heads=num_heads)  # parameter holder
        self._packed = None

    # ---- ModelMixin-ish surface the pipeline touches ----------------------------------------------------------
    @property
    def device(self):
        return self.text_fc.weight.device

    @property
    def dtype(self):
        return self.text_fc.weight.dtype

    def _pack(self):
        """Compose and convert the weights once per parameter version (fp32 math, f16 storage)."""
        ps = [self.text_fc.weight, self.text_fc.bias, self.vis_fc.weight, self.vis_fc.bias,
              self.multihead_attn.in_proj_weight, self.multihead_attn.in_proj_bias,
              self.multihead_attn.out_proj.weight, self.multihead_attn.out_proj.bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        E = self.hidden_dim
        with torch.no_grad():
            wt, bt, wv, bv, win, bin_, wo, bo = [p.detach().float() for p in ps]
            mm = hip.matmul_f32 if wt.is_cuda else torch.matmul   # (the library's own fp32 kernel on the GPU path)
            wq = mm(win[:E], wt).contiguous()
            bq = (mm(win[:E], bt) + bin_[:E]).contiguous()
            wkv = mm(win[E:], wv).contiguous()
            bkv = (mm(win[E:], bv) + bin_[E:]).contiguous()
            h = lambda w: w.to(torch.float16).contiguous()
            packed = dict(wq=h(wq), bq=bq, wkv=h(wkv), bkv=bkv, wo=h(wo), bo=bo.contiguous())
        self._packed = (key, packed)
        return packed

    @torch.no_grad()
    def forward(self, vis_f, text_f):
        dev = self.device
        if dev.type != "cuda":
            raise hip.RcdmError("rcdms_amd.context: the context builders run on the HIP path only (module is on "
                                f"{dev}); there is no CPU fallback")
        hip.load()
        E, H = self.hidden_dim, self.num_heads
        if vis_f.dim() != 3 or text_f.dim() != 3 or vis_f.shape[0] != text_f.shape[0]:
            raise ValueError(f"expected vis_f (k, Lv, vis_dim) and text_f (k, L, text_dim), got {tuple(vis_f.shape)} "
                             f"and {tuple(text_f.shape)}")
        n, L, Dt = text_f.shape
        Lv, Dv = vis_f.shape[1], vis_f.shape[2]
        if Dt != self.text_fc.in_features or Dv != self.vis_fc.in_features:
            raise ValueError(f"feature dims {Dt}/{Dv} != text_dim {self.text_fc.in_features} / vis_dim "
                             f"{self.vis_fc.in_features}")
        w = self._pack()
        out_dtype = text_f.dtype
        text32 = text_f.detach().to(dev, torch.float32).contiguous()
        vis32 = vis_f.detach().to(dev, torch.float32).contiguous()
        f16 = dict(dtype=torch.float16, device=dev)
        text16, vis16 = torch.empty(n * L, Dt, **f16), torch.empty(n * Lv, Dv, **f16)
        q, kv = torch.empty(n * L, E, **f16), torch.empty(n * Lv, 2 * E, **f16)
        ao, out = torch.empty(n * L, E, **f16), torch.empty(n * L, E, **f16)
        hip.pack_f16(text32.data_ptr(), text16.data_ptr(), text32.numel())
        hip.pack_f16(vis32.data_ptr(), vis16.data_ptr(), vis32.numel())

        def gemm(a, wt, bias, o, M, N, K):
            d = hip.GemmDesc(M, N, K, K, N, 0, hip.EPI_BIAS, 1, 0, 1.0, 1)  # split_k = 1: no workspace
            hip.gemm(d, a.data_ptr(), wt.data_ptr(), bias.data_ptr(), 0, 0, o.data_ptr(), 0, 0)

        gemm(text16, w["wq"], w["bq"], q, n * L, E, Dt)
        gemm(vis16, w["wkv"], w["bkv"], kv, n * Lv, 2 * E, Dv)
        dh = E // H
        ad = hip.AttnDesc(n, H, L, Lv, dh, E, 2 * E, 2 * E, E, dh ** -0.5)
        hip.flash_attn(ad, q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * E, ao.data_ptr())
        gemm(ao, w["wo"], w["bo"], out, n * L, E, E)
        return out.view(n, L, E).to(out_dtype)


class fine_stack(_ContextStack):
    """stage2_batchtest_rcdms_model.py:134-149 — `local_module` of the pipeline (seen frames)."""


class semantic_stack(_ContextStack):
    """stage2_batchtest_rcdms_model.py:117-132 — `global_module` of the pipeline (frames to generate)."""
