"""Every environment switch the Python side of rcdms_amd reads, in ONE place (INTEGRATION.md §6 lists them with the
switches librcdm_hip.so reads itself).  Rule: a switch selects between two CORRECT implementations of the same
arithmetic (same-process / same-box A/B of a design decision); nothing here can make a launch plan compute anything
but UNet3DConditionModel.forward.  Read once at import."""
import os


def _on(name, default="1"):
    return os.environ.get(name, default) != "0"


# name -> what "0" selects instead of the default                                               (reference lines it concerns)
SC_FOLD = _on("RCDM_SC_FOLD")        # 0: conv_shortcut as its own 1x1 GEMM + residual read instead of a tenth tap of conv2 (resnet.py:205-212)
UP2 = _on("RCDM_UP2")                # 0: Upsample3D as nearest-2x indexing + nine taps instead of four 2x2 phase convolutions (resnet.py:60-79)
GN_PRESTAT = _on("RCDM_GN_PRESTAT")  # 0: every GroupNorm takes its own statistics pass (no statistics-carrying split-K reduce)
LNX = _on("RCDM_LNX")                # 0: stand-alone LayerNorm launches below the 64x64 level instead of the deferred form (attention.py:482-514)
FFZ = _on("RCDM_FFZ")                # 0: ff.net.2 and proj_out as two GEMMs instead of the composed K = 5C one (attention.py:361,514)
FF_FUSE = _on("RCDM_FF_FUSE")        # 0: LayerNorm -> GEGLU -> ff.net.2 as three launches instead of rcdm_ff_fused
ROW_CHAIN = _on("RCDM_ROWCHAIN")     # 0: separate launches instead of the row-stationary chains of the 64x64 level
RANK1_CTX = _on("RCDM_RANK1_CTX")    # 0: cross-attention evaluated in full even for images whose context rows are all equal
# which latent sizes run the stride-1 3x3 convolutions of their ResNet blocks (>= 640 channels) in the Winograd F(2x2, 3x3) form
# (rcdm_conv3x3_wino, with the GroupNorm apply + SiLU in its input transform) instead of the nine-tap implicit GEMM
# (resnet.py:182-212): "le<N>" = every level whose (even) sides are in [8, N], a comma list = exactly those sides, "0" = none
_w = os.environ.get("RCDM_WINO", "le32").strip()
WINO_MAX_SIDE = int(_w[2:]) if _w.startswith("le") else 0
WINO_MIN_SIDE = 8     # (below: 4 tiles per image, unmeasured)
WINO_MIN_C = int(os.environ.get("RCDM_WINO_MIN_C", "640"))   # channels from which a conv takes the form (k-loops of >= 10 steps per position GEMM; 320: measured no gain)
WINO = tuple(int(v) for v in _w.split(",") if v.strip().isdigit() and int(v) > 0) if not _w.startswith("le") else ()


def wino_side(H, W):
    """Whether a level of H x W latents is one of the Winograd levels."""
    if H % 2 or W % 2:
        return False
    return (min(H, W) >= WINO_MIN_SIDE and max(H, W) <= WINO_MAX_SIDE) if WINO_MAX_SIDE else (H == W and H in WINO)


UP9 = _on("RCDM_UP9")                # 0: Upsample3D as four 2x2 phase convolutions (RCDM_UP2) instead of one 9-tap-plane GEMM + gather (resnet.py:60-79)
OUT_TAPS = _on("RCDM_OUT_TAPS")      # 0: conv_out (320 -> 4 channels, unet.py:457) as the nine-tap implicit GEMM instead of a 72-wide tap-plane GEMM + gather
CHAIN_MIN_ROWS = os.environ.get("RCDM_CHAIN_MIN_ROWS")   # token rows from which the chains are used (default: 3/4 of a chip of 160-row blocks)

TABLE = {
    "RCDM_SC_FOLD": SC_FOLD, "RCDM_UP2": UP2, "RCDM_GN_PRESTAT": GN_PRESTAT, "RCDM_LNX": LNX, "RCDM_FFZ": FFZ,
    "RCDM_FF_FUSE": FF_FUSE, "RCDM_ROWCHAIN": ROW_CHAIN, "RCDM_RANK1_CTX": RANK1_CTX, "RCDM_WINO": _w, "RCDM_WINO_MIN_C": WINO_MIN_C, "RCDM_UP9": UP9, "RCDM_OUT_TAPS": OUT_TAPS, "RCDM_CHAIN_MIN_ROWS": CHAIN_MIN_ROWS,
}
