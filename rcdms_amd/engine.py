"""Host-side launch planner for the stage-2 denoiser on MI355X — the public face of the planner package.

Turns the reference's module graph (UNet3DConditionModel.forward, src/models/unet.py:322-463 and the block wiring of
src/models/unet_blocks.py) into a flat, static list of librcdm_hip.so launches over pre-allocated HBM buffers, so that
one denoising step can be captured once into a hipGraph and replayed (RCDMs_pipeline.py:480-503 calls it T times per
story).  The parts (round 6: one module each, this file only re-exports them):

  plan.py          Buf / Rows / Plan / Geo — buffers, strided f16 row views, the ordered launch list
  switches.py      every environment switch the Python side reads (all select between two correct implementations)
  packer.py        fp32 reference state dict -> f16 kernel layouts + the pack-time weight algebra
  emit_ops.py      one emitter per C-ABI entry point (GEMM, conv3x3, norms, attention kernels)
  emit_blocks.py   ResnetBlock3D / Transformer3DModel / BasicTransformerBlock / motion module as launch sequences
  unet_program.py  UNetProgram: the whole UNet for one geometry, context plan, graph capture, forward()
  eager.py         run_tokens / run_block: the mirrored leaf classes' own forward()
  numerics.py      numerics_report: activation / score / LayerNorm headroom of a loaded checkpoint on the f16 path

Data layout in HBM: every activation is channels-last f16 rows X[(b f y x)][C] with an explicit row
stride; torch is used for device memory, streams, host<->device copies and the one-time pack-time weight algebra —
nothing torch computes is on the per-step path."""
from . import switches as SW                                                                         # noqa: F401
from .plan import Buf, Geo, Plan, Rows, _NS                                                          # noqa: F401
from .emit_ops import (LNX_MAX_PARTS, XATTN_MAX_KEYS, emit_conv3x3, emit_flash_attn, emit_flash_attn_masked,   # noqa: F401
                       emit_gemm, emit_groupnorm, emit_groupnorm_stats, emit_layernorm, emit_temporal_attn,
                       emit_upsample_conv, emit_xattn, emit_xattn_pack, gemm_lnx_ok)
from .packer import (MSUB_SCORE_LIMIT, Packer, attn_score_bound, pack_attention, pack_basic_block, pack_motion,   # noqa: F401
                     pack_resnet, pack_transformer)
from .emit_blocks import (CHAIN_MIN_ROWS, emit_basic_block, emit_ctx_kv, emit_ff, emit_motion, emit_rank1_ctx,   # noqa: F401
                          emit_resnet, emit_rowchain, emit_transformer, ffz_rows, full_rank_runs)
from .unet_program import CIN_PAD, COUT_PAD, UNetProgram                                             # noqa: F401
from .eager import ncfhw_from_rows, rows_from_ncfhw, run_block, run_tokens                           # noqa: F401
