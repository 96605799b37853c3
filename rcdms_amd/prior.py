"""Stage-1 frame-prior transformer on the HIP path.

Executes `MyPriorTransformer.forward` (reference src/models/myprior_transformer.py:275-411) as a static launch plan
over librcdm_hip.so, reusing the stage-2 kernels: per layer
  BasicTransformerBlock (attention.py:479-526, no cross-attention, attention biases, gelu FF):
      LayerNorm -> fused [q;k;v] GEMM (+bias) -> rcdm_flash_attn_masked (causal + text padding) -> out GEMM (+bias,
      +residual) -> LayerNorm -> GEMM (+bias, GELU epilogue) -> GEMM (+bias, +residual)
  motion module with prior_state (motion_module.py:147-182): LayerNorm `prior_norm` -> proj_in -> 2 x (LayerNorm + PE ->
      [q;k;v] GEMM -> rcdm_temporal_attn over the 5 frames of every token -> out GEMM + residual) -> GEGLU FF -> proj_out
      + residual.
Token rows are (b f) x L, L = num_embeddings + additional_embeddings (97).  Everything that does not depend on the
denoising step — the projected text states, the three projected conditioning vectors, the `prd` token, the positional
embedding — is assembled once per story by set_context(); a step only rewrites the time-embedding row and the
noisy-embedding row.  No CPU path: RcdmError without a GPU."""
import torch

from . import hip
from .engine import (Geo, Packer, Plan, Rows, _NS, emit_flash_attn_masked, emit_gemm, emit_layernorm, emit_motion,
                     pack_motion)


def _pack_block(pk, p):
    """BasicTransformerBlock of the prior: keys attn1.{to_q,to_k,to_v}.{weight,bias}, attn1.to_out.0, norm1, norm3, ff."""
    w = _NS()
    w.ln1 = (pk.vec(p + "norm1.weight"), pk.vec(p + "norm1.bias"))
    w.ln3 = (pk.vec(p + "norm3.weight"), pk.vec(p + "norm3.bias"))
    w.qkv = pk.mat_f16(p + "attn1.to_q.weight", p + "attn1.to_k.weight", p + "attn1.to_v.weight")
    w.qkv_b = torch.cat([pk.vec(p + f"attn1.{n}.bias") for n in ("to_q", "to_k", "to_v")]).contiguous()
    w.o, w.o_b = pk.mat_f16(p + "attn1.to_out.0.weight"), pk.vec(p + "attn1.to_out.0.bias")
    w.ff1, w.ff1_b = pk.mat_f16(p + "ff.net.0.proj.weight"), pk.vec(p + "ff.net.0.proj.bias")
    w.ff2, w.ff2_b = pk.mat_f16(p + "ff.net.2.weight"), pk.vec(p + "ff.net.2.bias")
    return w


class PriorProgram:
    """Launch plan of one MyPriorTransformer.forward for a fixed batch B = (cfg reps) x 5 frames and T text tokens."""

    FRAMES = 5  # `video_length = 5` is a literal in the reference (motion_module.py:150)

    def __init__(self, cfg, sd, B, T, device):
        hip.load()
        if B % self.FRAMES:
            raise ValueError(f"batch {B} is not a multiple of the 5 frames of a story")
        self.cfg, self.B, self.T = cfg, B, T
        self.device = torch.device(device)
        self.heads, self.dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
        self.C = C = self.heads * self.dh
        self.L = L = T + cfg["additional_embeddings"]
        self.E = cfg["embedding_dim"]
        self.clip_dim = cfg.get("clip_embed_dim") or self.E
        if sd["positional_embedding"].shape[1] != L:
            raise ValueError(f"positional_embedding holds {sd['positional_embedding'].shape[1]} positions, the sequence has "
                             f"{L} (num_embeddings + additional_embeddings; myprior_transformer.py:428-429)")
        mk = cfg["motion_module_kwargs"]
        self.m_heads, self.m_attn = mk["num_attention_heads"], len(mk["attention_block_types"])
        M = B * L
        pk = Packer(sd, self.device)
        self.pk_keep = []
        f32 = pk.f32
        self.pos = f32("positional_embedding")[0].contiguous()                 # (L, C)
        self.prd = f32("prd_embedding")[0, 0].contiguous()
        # per-story projections (set_context) and the per-step ones
        self.w_enc = (pk.mat_f16("encoder_hidden_states_proj.weight"), pk.vec("encoder_hidden_states_proj.bias"))
        self.w_emb = [(pk.mat_f16(f"{n}.weight"), pk.vec(f"{n}.bias")) for n in
                      ("embedding_proj", "embedding_proj1", "embedding_proj2")]
        self.w_in = pk.mat_f16("proj_in.weight")
        self.b_in = (pk.vec("proj_in.bias") + self.pos[L - 2]).contiguous()    # + positional embedding of its row
        self.w_t1, self.b_t1 = pk.mat_f16("time_embedding.linear_1.weight"), pk.vec("time_embedding.linear_1.bias")
        self.w_t2 = pk.mat_f16("time_embedding.linear_2.weight")
        self.b_t2 = (pk.vec("time_embedding.linear_2.bias") + self.pos[L - 3]).contiguous()
        self.w_clip, self.b_clip = pk.mat_f16("proj_to_clip_embeddings.weight"), pk.vec("proj_to_clip_embeddings.bias")
        ln_out = (pk.vec("norm_out.weight"), pk.vec("norm_out.bias"))
        blocks = [(_pack_block(pk, f"transformer_blocks.{2 * i}."),
                   pack_motion(pk, f"transformer_blocks.{2 * i + 1}.", self.m_attn)) for i in range(cfg["num_layers"])]
        pk.done()

        # ---- buffers and the body plan -------------------------------------------------------------------------
        self.plan = plan = Plan(self.device)
        self.tok = plan.rows("prior_tok", M, C, unique=True)        # residual stream, (b f) x L rows
        self.base = torch.zeros(M, C, dtype=torch.float16, device=self.device)   # step-independent rows (+ pos)
        self.kvalid = torch.ones(B, L, dtype=torch.uint8, device=self.device)
        self.t_dev = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.temb0 = torch.zeros(8, C, dtype=torch.float32, device=self.device)
        self.temb1 = torch.zeros(8, C, dtype=torch.float32, device=self.device)
        self.temb2 = torch.zeros(8, C, dtype=torch.float32, device=self.device)
        self.x16 = plan.rows("prior_x16", B, self.E, unique=True)
        self.out16 = plan.rows("prior_out", B, self.clip_dim, unique=True)
        tok = self.tok
        a = plan.rows("norm", M, C)
        geo = Geo(B // self.FRAMES, self.FRAMES, L, 1)
        for blk, mot in blocks:
            emit_layernorm(plan, tok, blk.ln1[0], blk.ln1[1], a)
            qkv = plan.rows("qkv", M, 3 * C)
            emit_gemm(plan, a, blk.qkv, 3 * C, C, qkv, bias=blk.qkv_b)
            ao = plan.rows("attn_out", M, C)
            emit_flash_attn_masked(plan, qkv.cols(0, C), qkv.cols(C, C), qkv.cols(2 * C, C), B, self.heads, L, L, self.dh,
                                   ao, self.kvalid, lambda: self.causal)
            emit_gemm(plan, ao, blk.o, C, C, tok, bias=blk.o_b, residual=tok)
            emit_layernorm(plan, tok, blk.ln3[0], blk.ln3[1], a)
            hid = plan.rows("geglu", M, 4 * C)
            emit_gemm(plan, a, blk.ff1, 4 * C, C, hid, bias=blk.ff1_b, gelu=True)
            emit_gemm(plan, hid, blk.ff2, C, 4 * C, tok, bias=blk.ff2_b, residual=tok)
            emit_motion(plan, mot, tok, geo, self.m_heads, tok, prior_state=True)
        # norm_out on the last token of every sample only (hidden_states[:, -1], :404-406), then the CLIP projection
        last = Rows(tok.buf, tok.off + (L - 1) * C, B, C, L * C)
        fin = plan.rows("prior_fin", B, C, unique=True)
        emit_layernorm(plan, last, ln_out[0], ln_out[1], fin)
        emit_gemm(plan, fin, self.w_clip, self.clip_dim, C, self.out16, bias=self.b_clip)
        plan.materialize()
        self._d_in = hip.GemmDesc(B, C, self.E, self.E, L * C, 0, hip.EPI_BIAS, 1, 0, 1.0, 1)
        self.ctx_key = None
        self.causal = True

    # ---- per story ------------------------------------------------------------------------------------------------
    def _linear(self, x32, w16, bias):
        """(rows, K) fp32 -> (rows, N) fp32 through rcdm_gemm (f16 operands, fp32 accumulate)."""
        rows, K = x32.shape
        N = w16.shape[0]
        x16 = torch.empty(rows, K, dtype=torch.float16, device=self.device)
        hip.pack_f16(x32.data_ptr(), x16.data_ptr(), x32.numel())
        out = torch.empty(rows, N, dtype=torch.float16, device=self.device)
        d = hip.GemmDesc(rows, N, K, K, N, 0, hip.EPI_BIAS, 1, 0, 1.0, 1)
        hip.gemm(d, x16.data_ptr(), w16.data_ptr(), bias.data_ptr(), 0, 0, out.data_ptr(), 0, 0)
        return out.float()

    def set_context(self, proj_embedding, encoder_hidden_states, proj_embedding1, mask_label, attention_mask):
        B, T, L, C = self.B, self.T, self.L, self.C
        dev = self.device
        to = lambda t: t.detach().to(dev, torch.float32).contiguous()
        enc = to(encoder_hidden_states)
        if tuple(enc.shape) != (B, T, self.E):
            raise hip.RcdmError(f"encoder_hidden_states shape {tuple(enc.shape)} != {(B, T, self.E)}")
        seq = torch.empty(B, L, C, dtype=torch.float32, device=dev)
        seq[:, :T] = self._linear(enc.reshape(B * T, self.E), *self.w_enc).reshape(B, T, C)
        for i, x in enumerate((proj_embedding, proj_embedding1, mask_label)):
            seq[:, T + i] = self._linear(to(x).reshape(B, self.E), *self.w_emb[i])
        seq[:, T + 3:T + 5] = 0.0                       # time / noisy-embedding rows: written every step
        seq[:, T + 5] = self.prd
        seq += self.pos
        seq[:, T + 3:T + 5] = 0.0
        self.base.copy_(seq.reshape(B * L, C).to(torch.float16))
        self.kvalid.fill_(1)
        # the reference adds the causal mask only together with a padding mask (myprior_transformer.py:389-393); with
        # attention_mask=None its blocks attend bidirectionally.  The flag is read at launch time, so a captured graph
        # must be re-captured when it changes (PriorLoop checks `causal` against what it captured).
        self.causal = attention_mask is not None
        if attention_mask is not None:
            self.kvalid[:, :T] = (to(attention_mask).reshape(B, T) != 0).to(torch.uint8)
        self.ctx_key = True

    # ---- per step --------------------------------------------------------------------------------------------------
    def step_ops(self, latents, n_lat):
        """The launches of one forward on the current stream, reading the timestep from self.t_dev[0] (device) and the
        noisy embeddings from `latents` (device fp32 [n_lat][E], sample b uses row b % n_lat): time embedding ->
        sequence assembly -> proj_in into its row -> the transformer body.  No host-side value is baked in, so the
        list can be captured in a hipGraph and replayed."""
        B, L, C, E = self.B, self.L, self.C, self.E
        return [
            lambda: hip.timestep_embed(self.t_dev.data_ptr(), 1, C, self.temb0.data_ptr()),
            lambda: hip.small_linear(self.temb0.data_ptr(), 1, C, self.w_t1.data_ptr(), self.b_t1.data_ptr(), C, 0, 1,
                                     self.temb1.data_ptr()),
            lambda: hip.small_linear(self.temb1.data_ptr(), 1, C, self.w_t2.data_ptr(), self.b_t2.data_ptr(), C, 0, 0,
                                     self.temb2.data_ptr()),
            lambda: hip.prior_assemble(self.base.data_ptr(), self.temb2.data_ptr(), latents.data_ptr(), n_lat, self.tok.ptr,
                                       self.x16.ptr, B, L, C, E, L - 3),
            lambda: hip.gemm(self._d_in, self.x16.ptr, self.w_in.data_ptr(), self.b_in.data_ptr(), 0, 0,
                             self.tok.ptr + (L - 2) * C * 2, 0, 0),
            self.plan.run,
        ]

    @torch.no_grad()
    def forward(self, hidden_states, timestep):
        """MyPriorTransformer.forward for one (hidden_states (B, E), scalar timestep) -> (B, clip_dim) fp32."""
        if self.ctx_key is None:
            raise hip.RcdmError("PriorProgram.forward before set_context")
        B = self.B
        t = torch.as_tensor(timestep, dtype=torch.float32, device=self.device).reshape(-1)
        if t.numel() != 1 and not bool((t == t[0]).all()):
            raise NotImplementedError("per-sample timesteps: the reference pipeline passes one scalar per step "
                                      "(prior_pipeline.py:317)")
        self.t_dev[:1] = t[:1]
        x32 = hidden_states.detach().to(self.device, torch.float32).contiguous()
        if tuple(x32.shape) != (B, self.E):
            raise hip.RcdmError(f"hidden_states shape {tuple(x32.shape)} != {(B, self.E)}")
        for op in self.step_ops(x32, B):
            op()
        torch.cuda.current_stream(self.device).synchronize()   # x32 must outlive the launches
        return self.out_rows().float()

    def out_rows(self):
        """(B, clip_dim) f16 view of the predicted embeddings."""
        return self.out16.buf.t[:self.B * self.clip_dim * 2].view(torch.float16).view(self.B, self.clip_dim)
