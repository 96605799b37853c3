"""ctypes binding of librcdm_hip.so (include/rcdm.h).  Thin: tensors stay owned by torch, only
`data_ptr()` integers and the current HIP stream handle cross the boundary.  There is NO CPU
fallback: if the library is missing or a call fails this module raises."""
import ctypes as C
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RCDM_LIB") or os.path.join(_HERE, "lib", "librcdm_hip.so")  # RCDM_LIB: kernel experiments

EPI_BIAS, EPI_ROWVEC, EPI_RESIDUAL, EPI_GEGLU, EPI_GELU = 1, 2, 4, 8, 16

_ERR = {-1: "RCDM_EINVAL", -2: "RCDM_ESHAPE", -3: "RCDM_ELAUNCH", -4: "RCDM_EWORKSPACE", -5: "RCDM_ECOMM"}


class RcdmError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
                ("epilogue", C.c_int32), ("rows_per_sample", C.c_int32), ("ldt", C.c_int32),
                ("out_scale", C.c_float), ("split_k", C.c_int32), ("dup_rows", C.c_int32)]


class LnFuse(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("pe", C.c_void_p), ("out", C.c_void_p), ("ld", C.c_int32),
                ("rows_per_frame", C.c_int32), ("frames", C.c_int32), ("eps", C.c_float)]


class Lnx(C.Structure):
    _fields_ = [("stat_out", C.c_void_p), ("stat_parts", C.c_int32), ("stat_out_rows", C.c_int32), ("stat_in", C.c_void_p),
                ("parts_in", C.c_int32), ("stat_in_rows", C.c_int32), ("colsum", C.c_void_p), ("eps", C.c_float), ("C", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("n_img", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
                ("c_in", C.c_int32), ("c_out", C.c_int32), ("stride", C.c_int32),
                ("upsample", C.c_int32), ("lda", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32),
                ("epilogue", C.c_int32), ("rows_per_sample", C.c_int32), ("ldt", C.c_int32),
                ("out_scale", C.c_float), ("split_k", C.c_int32), ("pad_after_only", C.c_int32),
                ("dup_rows", C.c_int32), ("c_in2", C.c_int32), ("lda2", C.c_int32)]


class GroupNormDesc(C.Structure):
    _fields_ = [("samples", C.c_int32), ("rows_per_sample", C.c_int32), ("C", C.c_int32),
                ("groups", C.c_int32), ("ldx", C.c_int32), ("ldy", C.c_int32),
                ("eps", C.c_float), ("silu", C.c_int32)]


class LayerNormDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32), ("ldy", C.c_int32),
                ("eps", C.c_float), ("rows_per_frame", C.c_int32), ("frames", C.c_int32)]


class FFDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32), ("ldo", C.c_int32), ("eps", C.c_float)]


class RowChainDesc(C.Structure):
    _fields_ = [("M", C.c_int32), ("C", C.c_int32), ("lda", C.c_int32), ("ldr", C.c_int32), ("ldt", C.c_int32),
                ("ldo", C.c_int32), ("tail", C.c_int32), ("rows_per_frame", C.c_int32), ("frames", C.c_int32),
                ("eps", C.c_float), ("gn_groups", C.c_int32), ("gn_rows", C.c_int32), ("ldz", C.c_int32)]


class AttnDesc(C.Structure):
    _fields_ = [("batch", C.c_int32), ("heads", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
                ("d", C.c_int32), ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32),
                ("ldo", C.c_int32), ("scale", C.c_float), ("flags", C.c_int32)]


ATTN_WIDE_RANGE = 1   # AttnDesc.flags: scores not bounded below 2^15 -> never the matrix-pipe-softmax (MSUB) kernel


class TemporalAttnDesc(C.Structure):
    _fields_ = [("samples", C.c_int32), ("frames", C.c_int32), ("pixels", C.c_int32),
                ("heads", C.c_int32), ("d", C.c_int32), ("ldqkv", C.c_int32), ("ldo", C.c_int32),
                ("scale", C.c_float)]


# every symbol include/rcdm.h declares: name -> (restype, argtypes)
_P, _I, _F, _SZ = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
SYMBOLS = {
    "rcdm_version": (C.c_int, []),
    "rcdm_last_hip_error": (C.c_int, []),
    "rcdm_last_hip_error_string": (C.c_char_p, []),
    "rcdm_gemm_ln": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(LnFuse), _P, _P, _P, _P, _P, _P]),
    "rcdm_gemm_stat_parts": (C.c_int, [C.POINTER(GemmDesc)]),
    "rcdm_gemm_lnx_stat_parts": (C.c_int, [C.POINTER(GemmDesc), _I]),
    "rcdm_gemm_lnx_parts_ok": (C.c_int, [C.POINTER(GemmDesc), _I, _I]),
    "rcdm_gemm_plan_query": (C.c_int, [C.POINTER(GemmDesc), _I, _I, C.POINTER(C.c_int32)]),
    "rcdm_conv3x3_plan_query": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    "rcdm_gemm_lnx_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc), _I, _I]),
    "rcdm_set_groupnorm_fold": (C.c_int, [_I]),
    "rcdm_gemm_lnx": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(Lnx), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_gemm_workspace_bytes": (_SZ, [C.POINTER(GemmDesc)]),
    "rcdm_set_igemm_variant": (C.c_int, [_I]),
    "rcdm_set_shape_rules": (C.c_int, [C.c_char_p]),
    "rcdm_set_igemm_pingpong": (C.c_int, [_I]),
    "rcdm_debug_set_igemm_trace": (C.c_int, [_P]),
    "rcdm_debug_set_attn_trace": (C.c_int, [_P]),
    "rcdm_debug_mfma_peak": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P]),
    "rcdm_gemm": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_conv3x3_workspace_bytes": (_SZ, [C.POINTER(ConvDesc)]),
    "rcdm_conv3x3": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_conv3x3_add1x1": (C.c_int, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_conv3x3_add1x1_gnstat": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "rcdm_groupnorm_workspace_bytes": (_SZ, [C.POINTER(GroupNormDesc)]),
    "rcdm_groupnorm_silu": (C.c_int, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_groupnorm_stats": (C.c_int, [C.POINTER(GroupNormDesc), _P, _P, _P, _SZ, _P]),
    "rcdm_groupnorm_stats_prestat": (C.c_int, [C.POINTER(GroupNormDesc), _P, _P, _SZ, _P]),
    "rcdm_groupnorm_prestat_ok": (C.c_int, [C.POINTER(GroupNormDesc)]),
    "rcdm_groupnorm_silu_prestat": (C.c_int, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _SZ, _P]),
    "rcdm_gemm_gnstat_ok": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GroupNormDesc)]),
    "rcdm_gemm_gnstat": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "rcdm_conv3x3_gnstat_ok": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(GroupNormDesc)]),
    "rcdm_conv3x3_gnstat": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P, _SZ, _P]),
    "rcdm_softmax_rows": (C.c_int, [_I, _I, _I, _I, C.c_float, _P, _P, _P]),
    "rcdm_layernorm": (C.c_int, [C.POINTER(LayerNormDesc), _P, _P, _P, _P, _P, _P]),
    "rcdm_flash_attn": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _P]),
    "rcdm_flash_attn_masked": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P, _I, _P, _P]),
    "rcdm_temporal_attn": (C.c_int, [C.POINTER(TemporalAttnDesc), _P, _P, _P]),
    "rcdm_ff_stream_bytes": (_SZ, [_I]),
    "rcdm_ff_fused_supported": (C.c_int, [_I]),
    "rcdm_pack_ff_stream": (C.c_int, [_P, _P, _P, _I, _P, _P, _P]),
    "rcdm_ff_fused": (C.c_int, [C.POINTER(FFDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "rcdm_rowchain_supported": (C.c_int, [_I]),
    "rcdm_rowchain_config_supported": (C.c_int, [_I, _I, _I]),
    "rcdm_rowchain_stream_bytes": (_SZ, [_I, _I]),
    "rcdm_pack_rowchain": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "rcdm_rowchain": (C.c_int, [C.POINTER(RowChainDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rcdm_timestep_embed": (C.c_int, [_P, _I, _I, _P, _P]),
    "rcdm_small_linear": (C.c_int, [_P, _I, _I, _P, _P, _I, _I, _I, _P, _P]),
    "rcdm_assemble_input": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "rcdm_ncfhw_to_rows": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "rcdm_rows_to_ncfhw": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "rcdm_cfg_ddim_step": (C.c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "rcdm_cfg_pndm_step": (C.c_int, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P]),
    "rcdm_prior_assemble": (C.c_int, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rcdm_cfg_unclip_step": (C.c_int, [_P, _I, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, _P]),
    "rcdm_load_timestep": (C.c_int, [_P, _P, _P, _I, _P]),
    "rcdm_xattn_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rcdm_xattn_pack_kv": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "rcdm_xattn": (C.c_int, [C.POINTER(AttnDesc), _P, _P, _P, _P]),
    "rcdm_advance_step": (C.c_int, [_P, _P]),
    "rcdm_load_table_row": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "rcdm_pack_f16": (C.c_int, [_P, _P, _SZ, _P]),
    "rcdm_mish": (C.c_int, [_P, _P, _SZ, _P]),
    "rcdm_pack_conv3x3": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "rcdm_pack_conv3x3_up2": (C.c_int, [_P, _I, _I, _P, _P]),
    "rcdm_matmul_f32": (C.c_int, [_P, _P, _P, _I, _I, _I, _P]),
    "rcdm_conv3x3_up2_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rcdm_conv3x3_wino_supported": (C.c_int, [C.POINTER(ConvDesc)]),
    "rcdm_conv3x3_wino_workspace_bytes": (_SZ, [C.POINTER(ConvDesc)]),
    "rcdm_conv3x3_wino_plan_query": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    "rcdm_pack_conv3x3_wino": (C.c_int, [_P, _I, _I, _P, _P]),
    "rcdm_set_wino_slab_f16": (C.c_int, [_I]),
    "rcdm_upsample_taps_gather": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "rcdm_conv_taps_gather": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "rcdm_set_splitk_slab_f16": (C.c_int, [_I]),
    "rcdm_conv3x3_wino": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ,
                                    C.POINTER(GroupNormDesc), _P, _P]),
    "rcdm_groupnorm_finalize": (C.c_int, [_I, _I, _I, C.c_float, _P, _P, _P]),
    "rcdm_groupnorm_apply": (C.c_int, [C.POINTER(GroupNormDesc), _P, _P, _P, _P, _P, _P]),
    "rcdm_pack_geglu_rows": (C.c_int, [_P, _P, _I, _I, _P, _P, _P]),
    "rcdm_graph_begin_capture": (C.c_int, [_P]),
    "rcdm_graph_end_capture": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "rcdm_graph_launch": (C.c_int, [_P, _P]),
    "rcdm_graph_destroy": (C.c_int, [_P]),
    "rcdm_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rcdm_event_record": (C.c_int, [_P, _P]),
    "rcdm_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "rcdm_event_destroy": (C.c_int, [_P]),
    "rcdm_stream_synchronize": (C.c_int, [_P]),
    "rcdm_comm_unique_id": (C.c_int, [_P]),
    "rcdm_comm_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "rcdm_comm_destroy": (C.c_int, [_P]),
    "rcdm_bcast": (C.c_int, [_P, _P, C.c_size_t, C.c_int32, _P]),
    "rcdm_allgather": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "rcdm_comm_last_error": (C.c_int, []),
}

_lib = None


def load():
    """Load the shared library (once) and type every declared symbol.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RcdmError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -m rcdms_amd.build`); "
            "rcdms_amd has no CPU fallback by design")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        lib = load()
        extra = ""
        if rc == -3:
            extra = f" (hip error {lib.rcdm_last_hip_error()}: {lib.rcdm_last_hip_error_string().decode()})"
        if rc == -5:
            extra = f" (ncclResult {lib.rcdm_comm_last_error()}; 0 = librccl could not be opened)"
        raise RcdmError(f"{what} failed: {_ERR.get(rc, rc)}{extra}")


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# thin typed wrappers (all enqueue on torch's current stream unless `stream` is given)

def set_shape_rules(rules):
    """Per-shape tile rules "taps,M,N,Cin,variant,split;..." ("off": no table; None: back to RCDM_SHAPE_RULES)."""
    _check(load().rcdm_set_shape_rules(None if rules is None else rules.encode()), "rcdm_set_shape_rules")


def set_igemm_variant(v):
    _check(load().rcdm_set_igemm_variant(v), "rcdm_set_igemm_variant")


def set_igemm_pingpong(on):
    _check(load().rcdm_set_igemm_pingpong(int(bool(on))), "rcdm_set_igemm_pingpong")


def gemm_workspace_bytes(desc):
    return load().rcdm_gemm_workspace_bytes(C.byref(desc))


def gemm(desc, A, W, bias, rowvec, residual, out, ws_ptr=0, ws_bytes=0, stream=None):
    _check(load().rcdm_gemm(C.byref(desc), A, W, bias, rowvec, residual, out, ws_ptr, ws_bytes,
                            stream_ptr() if stream is None else stream), "rcdm_gemm")


def conv3x3_workspace_bytes(desc):
    return load().rcdm_conv3x3_workspace_bytes(C.byref(desc))


def conv3x3(desc, x, W, bias, rowvec, residual, out, ws_ptr=0, ws_bytes=0, stream=None):
    _check(load().rcdm_conv3x3(C.byref(desc), x, W, bias, rowvec, residual, out, ws_ptr, ws_bytes,
                               stream_ptr() if stream is None else stream), "rcdm_conv3x3")


def conv3x3_add1x1(desc, x, x2, W, bias, rowvec, residual, out, ws_ptr=0, ws_bytes=0, stream=None):
    _check(load().rcdm_conv3x3_add1x1(C.byref(desc), x, x2, W, bias, rowvec, residual, out, ws_ptr, ws_bytes,
                                      stream_ptr() if stream is None else stream), "rcdm_conv3x3_add1x1")


def groupnorm_workspace_bytes(desc):
    return load().rcdm_groupnorm_workspace_bytes(C.byref(desc))


def groupnorm_silu(desc, x, gamma, beta, y, ws_ptr, ws_bytes, stream=None):
    _check(load().rcdm_groupnorm_silu(C.byref(desc), x, gamma, beta, y, ws_ptr, ws_bytes,
                                      stream_ptr() if stream is None else stream), "rcdm_groupnorm_silu")


def groupnorm_silu_prestat(desc, x, gamma, beta, y, ws_ptr, ws_bytes, stream=None):
    _check(load().rcdm_groupnorm_silu_prestat(C.byref(desc), x, gamma, beta, y, ws_ptr, ws_bytes,
                                              stream_ptr() if stream is None else stream), "rcdm_groupnorm_silu_prestat")


def groupnorm_prestat_ok(desc):
    return bool(load().rcdm_groupnorm_prestat_ok(C.byref(desc)))


def gemm_gnstat_ok(desc, gn):
    return bool(load().rcdm_gemm_gnstat_ok(C.byref(desc), C.byref(gn)))


def gemm_gnstat(desc, gn, A, W, bias, rowvec, residual, out, ws_ptr, ws_bytes, gn_ws_ptr, gn_ws_bytes, stream=None):
    _check(load().rcdm_gemm_gnstat(C.byref(desc), C.byref(gn), A, W, bias, rowvec, residual, out, ws_ptr, ws_bytes, gn_ws_ptr,
                                   gn_ws_bytes, stream_ptr() if stream is None else stream), "rcdm_gemm_gnstat")


def conv3x3_gnstat_ok(desc, gn):
    return bool(load().rcdm_conv3x3_gnstat_ok(C.byref(desc), C.byref(gn)))


def conv3x3_gnstat(desc, gn, x, W, bias, rowvec, residual, out, ws_ptr, ws_bytes, gn_ws_ptr, gn_ws_bytes, stream=None):
    _check(load().rcdm_conv3x3_gnstat(C.byref(desc), C.byref(gn), x, W, bias, rowvec, residual, out, ws_ptr, ws_bytes, gn_ws_ptr,
                                      gn_ws_bytes, stream_ptr() if stream is None else stream), "rcdm_conv3x3_gnstat")


def conv3x3_add1x1_gnstat(desc, gn, x, x2, W, bias, rowvec, residual, out, ws_ptr, ws_bytes, gn_ws_ptr, gn_ws_bytes, stream=None):
    _check(load().rcdm_conv3x3_add1x1_gnstat(C.byref(desc), C.byref(gn), x, x2, W, bias, rowvec, residual, out, ws_ptr, ws_bytes,
                                             gn_ws_ptr, gn_ws_bytes, stream_ptr() if stream is None else stream),
           "rcdm_conv3x3_add1x1_gnstat")


def groupnorm_stats(desc, x, stat, ws_ptr, ws_bytes, stream=None):
    _check(load().rcdm_groupnorm_stats(C.byref(desc), x, stat, ws_ptr, ws_bytes,
                                       stream_ptr() if stream is None else stream), "rcdm_groupnorm_stats")


def groupnorm_stats_prestat(desc, stat, ws_ptr, ws_bytes, stream=None):
    _check(load().rcdm_groupnorm_stats_prestat(C.byref(desc), stat, ws_ptr, ws_bytes,
                                               stream_ptr() if stream is None else stream), "rcdm_groupnorm_stats_prestat")


def layernorm(desc, x, gamma, beta, pe, y, stream=None):
    _check(load().rcdm_layernorm(C.byref(desc), x, gamma, beta, pe, y,
                                 stream_ptr() if stream is None else stream), "rcdm_layernorm")


def softmax_rows(M, N, ldx, ldy, scale, x, y, stream=None):
    _check(load().rcdm_softmax_rows(M, N, ldx, ldy, scale, x, y, stream_ptr() if stream is None else stream),
           "rcdm_softmax_rows")


def gemm_ln(desc, ln, a, w, bias, residual, out, stream=None):
    _check(load().rcdm_gemm_ln(C.byref(desc), C.byref(ln), a, w, bias, residual, out,
                               stream_ptr() if stream is None else stream), "rcdm_gemm_ln")


def set_groupnorm_fold(on):
    _check(load().rcdm_set_groupnorm_fold(int(on)), "rcdm_set_groupnorm_fold")


def gemm_stat_parts(desc, consumer=False):
    """Column-tile count (= statistics slots per row) of a statistics-producing launch of this shape; consumer: the launch
    also consumes a deferred LayerNorm (both flags steer the tile choice: rcdm_gemm_lnx_stat_parts)."""
    if consumer:
        return int(load().rcdm_gemm_lnx_stat_parts(C.byref(desc), 1))
    return int(load().rcdm_gemm_stat_parts(C.byref(desc)))


_PLAN_FIELDS = ("variant", "bm", "bn", "tiles_m", "tiles_n", "splits", "blocks_per_cu", "k_steps")


def gemm_plan_query(desc, producer=False, consumer=False):
    out = (C.c_int32 * 8)()
    _check(load().rcdm_gemm_plan_query(C.byref(desc), int(bool(producer)), int(bool(consumer)), out), "rcdm_gemm_plan_query")
    return dict(zip(_PLAN_FIELDS, out))


def conv3x3_plan_query(desc):
    out = (C.c_int32 * 8)()
    _check(load().rcdm_conv3x3_plan_query(C.byref(desc), out), "rcdm_conv3x3_plan_query")
    return dict(zip(_PLAN_FIELDS, out))


def gemm_lnx_parts_ok(desc, parts, consumer=False):
    return bool(load().rcdm_gemm_lnx_parts_ok(C.byref(desc), int(parts), int(bool(consumer))))


def gemm_lnx_workspace_bytes(desc, producer=False, consumer=False):
    return int(load().rcdm_gemm_lnx_workspace_bytes(C.byref(desc), int(bool(producer)), int(bool(consumer))))


def gemm_lnx(desc, lnx, a, w, bias, rowvec, residual, out, workspace, workspace_bytes, stream=None):
    _check(load().rcdm_gemm_lnx(C.byref(desc), C.byref(lnx), a, w, bias, rowvec, residual, out, workspace, workspace_bytes,
                                stream_ptr() if stream is None else stream), "rcdm_gemm_lnx")


def flash_attn(desc, q, k, v, out, stream=None):
    _check(load().rcdm_flash_attn(C.byref(desc), q, k, v, out,
                                  stream_ptr() if stream is None else stream), "rcdm_flash_attn")


def flash_attn_masked(desc, q, k, v, key_valid, causal, out, stream=None):
    _check(load().rcdm_flash_attn_masked(C.byref(desc), q, k, v, key_valid, int(causal), out,
                                         stream_ptr() if stream is None else stream), "rcdm_flash_attn_masked")


def temporal_attn(desc, qkv, out, stream=None):
    _check(load().rcdm_temporal_attn(C.byref(desc), qkv, out,
                                     stream_ptr() if stream is None else stream), "rcdm_temporal_attn")


def ff_stream_bytes(Cc):
    return load().rcdm_ff_stream_bytes(Cc)


def ff_fused_supported(Cc):
    return bool(load().rcdm_ff_fused_supported(Cc))


def rowchain_supported(Cc):
    return bool(load().rcdm_rowchain_supported(Cc))


def rowchain_config_supported(Cc, tail, pe_frames=0):
    return bool(load().rcdm_rowchain_config_supported(Cc, tail, pe_frames))


def rowchain_stream_bytes(Cc, tail):
    return load().rcdm_rowchain_stream_bytes(Cc, tail)


def pack_rowchain(wa, Cc, tail, wt, w1, b1, w2, wstream, b1p, stream=None):
    _check(load().rcdm_pack_rowchain(wa, Cc, tail, wt, w1, b1, w2, wstream, b1p,
                                     stream_ptr() if stream is None else stream), "rcdm_pack_rowchain")


def rowchain(desc, a_in, res, tok, a_bias, ln_g, ln_b, pe, wstream, b1p, b2, out, stream=None, gn_stat=None, gn_g=None,
             gn_b=None, z_res=None, z_bias=None):
    _check(load().rcdm_rowchain(C.byref(desc), a_in, res, tok, a_bias, ln_g, ln_b, pe, wstream, b1p, b2, out,
                                gn_stat, gn_g, gn_b, z_res, z_bias, stream_ptr() if stream is None else stream), "rcdm_rowchain")


def pack_ff_stream(w1, b1, w2, Cc, wstream, b1p, stream=None):
    _check(load().rcdm_pack_ff_stream(w1, b1, w2, Cc, wstream, b1p, stream_ptr() if stream is None else stream),
           "rcdm_pack_ff_stream")


def ff_fused(desc, x, ln_g, ln_b, wstream, b1p, b2, out, stream=None):
    _check(load().rcdm_ff_fused(C.byref(desc), x, ln_g, ln_b, wstream, b1p, b2, out,
                                stream_ptr() if stream is None else stream), "rcdm_ff_fused")


def timestep_embed(t, rows, dim, out, stream=None):
    _check(load().rcdm_timestep_embed(t, rows, dim, out, stream_ptr() if stream is None else stream),
           "rcdm_timestep_embed")


def small_linear(x, rows, K, W, bias, N, silu_in, silu_out, out, stream=None):
    _check(load().rcdm_small_linear(x, rows, K, W, bias, N, silu_in, silu_out, out,
                                    stream_ptr() if stream is None else stream), "rcdm_small_linear")


def prior_assemble(base, temb, latents, n_lat, tok, x16, B, L, Cc, E, time_row, stream=None):
    _check(load().rcdm_prior_assemble(base, temb, latents, n_lat, tok, x16, B, L, Cc, E, time_row,
                                      stream_ptr() if stream is None else stream), "rcdm_prior_assemble")


def cfg_unclip_step(pred, ld, latents, n, reps, E, guidance_scale, clip_range, coef, noise, step_counter, stream=None):
    _check(load().rcdm_cfg_unclip_step(pred, ld, latents, n, reps, E, guidance_scale, clip_range, coef, noise,
                                       step_counter, stream_ptr() if stream is None else stream), "rcdm_cfg_unclip_step")


def assemble_input(lat, mask, masked, S, reps, frames, H, W, out, ld, c_pad, stream=None):
    _check(load().rcdm_assemble_input(lat, mask, masked, S, reps, frames, H, W, out, ld, c_pad,
                                      stream_ptr() if stream is None else stream), "rcdm_assemble_input")


def ncfhw_to_rows(x, b, Cc, frames, H, W, out, ld, c_pad, stream=None):
    _check(load().rcdm_ncfhw_to_rows(x, b, Cc, frames, H, W, out, ld, c_pad,
                                     stream_ptr() if stream is None else stream), "rcdm_ncfhw_to_rows")


def rows_to_ncfhw(rows, ld, b, Cc, frames, H, W, out, stream=None):
    _check(load().rcdm_rows_to_ncfhw(rows, ld, b, Cc, frames, H, W, out,
                                     stream_ptr() if stream is None else stream), "rcdm_rows_to_ncfhw")


def cfg_pndm_step(eps, ld, lat, hist, S, reps, frames, H, W, gs, table, step, stream=None):
    _check(load().rcdm_cfg_pndm_step(eps, ld, lat, hist, S, reps, frames, H, W, gs, table, step,
                                     stream_ptr() if stream is None else stream), "rcdm_cfg_pndm_step")


def cfg_ddim_step(eps, ld, lat, S, reps, frames, H, W, gs, coef, step, stream=None):
    _check(load().rcdm_cfg_ddim_step(eps, ld, lat, S, reps, frames, H, W, gs, coef, step,
                                     stream_ptr() if stream is None else stream), "rcdm_cfg_ddim_step")


def load_timestep(ts, step, t_out, rows, stream=None):
    _check(load().rcdm_load_timestep(ts, step, t_out, rows, stream_ptr() if stream is None else stream),
           "rcdm_load_timestep")


def xattn_image_bytes(batch, heads, d):
    return load().rcdm_xattn_image_bytes(batch, heads, d)


def xattn_pack_kv(k, v, batch, Lk, heads, d, ldk, ldv, image, stream=None):
    _check(load().rcdm_xattn_pack_kv(k, v, batch, Lk, heads, d, ldk, ldv, image,
                                     stream_ptr() if stream is None else stream), "rcdm_xattn_pack_kv")


def xattn(desc, q, image, out, stream=None):
    _check(load().rcdm_xattn(C.byref(desc), q, image, out, stream_ptr() if stream is None else stream), "rcdm_xattn")


def load_table_row(table, step, dst, row_floats, stream=None):
    _check(load().rcdm_load_table_row(table, step, dst, row_floats, stream_ptr() if stream is None else stream),
           "rcdm_load_table_row")


def advance_step(step, stream=None):
    _check(load().rcdm_advance_step(step, stream_ptr() if stream is None else stream), "rcdm_advance_step")


def pack_f16(src, dst, n, stream=None):
    _check(load().rcdm_pack_f16(src, dst, n, stream_ptr() if stream is None else stream), "rcdm_pack_f16")


def mish(x, y, n, stream=None):
    _check(load().rcdm_mish(x, y, n, stream_ptr() if stream is None else stream), "rcdm_mish")


def matmul_f32(a, b):
    """a [n][k] @ b [k][m] (or b [k]) in fp32 on the library's own kernel (rcdm_matmul_f32): weight composition at pack time."""
    import torch
    a = a.contiguous().float()
    vec = b.dim() == 1
    b2 = (b[:, None] if vec else b).contiguous().float()
    n, k = a.shape
    m = b2.shape[1]
    assert b2.shape[0] == k and a.is_cuda and b2.is_cuda
    c = torch.empty(n, m, dtype=torch.float32, device=a.device)
    _check(load().rcdm_matmul_f32(a.data_ptr(), b2.data_ptr(), c.data_ptr(), n, k, m, stream_ptr()), "rcdm_matmul_f32")
    return c[:, 0].contiguous() if vec else c


def pack_conv3x3_up2(w, c_out, c_in, dst, stream=None):
    _check(load().rcdm_pack_conv3x3_up2(w, c_out, c_in, dst, stream_ptr() if stream is None else stream),
           "rcdm_pack_conv3x3_up2")


def conv3x3_up2_supported(desc):
    return bool(load().rcdm_conv3x3_up2_supported(C.byref(desc)))


def upsample_taps_gather(P, ldp, n_img, h, w, c_out, bias, out, ldc, stream=None):
    _check(load().rcdm_upsample_taps_gather(P, ldp, n_img, h, w, c_out, bias, out, ldc, stream_ptr() if stream is None else stream),
           "rcdm_upsample_taps_gather")


def conv_taps_gather(P, ldp, n_img, h, w, c_out, upsample, bias, out, ldc, stream=None):
    _check(load().rcdm_conv_taps_gather(P, ldp, n_img, h, w, c_out, upsample, bias, out, ldc, stream_ptr() if stream is None else stream),
           "rcdm_conv_taps_gather")


def set_splitk_slab_f16(on):
    _check(load().rcdm_set_splitk_slab_f16(on), "rcdm_set_splitk_slab_f16")


def set_wino_slab_f16(on):
    _check(load().rcdm_set_wino_slab_f16(on), "rcdm_set_wino_slab_f16")


def conv3x3_wino_supported(desc):
    return bool(load().rcdm_conv3x3_wino_supported(C.byref(desc)))


def conv3x3_wino_workspace_bytes(desc):
    return load().rcdm_conv3x3_wino_workspace_bytes(C.byref(desc))


def conv3x3_wino_plan_query(desc):
    out = (C.c_int32 * 8)()
    _check(load().rcdm_conv3x3_wino_plan_query(C.byref(desc), out), "rcdm_conv3x3_wino_plan_query")
    return list(out)


def pack_conv3x3_wino(w, c_out, c_in, dst, stream=None):
    _check(load().rcdm_pack_conv3x3_wino(w, c_out, c_in, dst, stream_ptr() if stream is None else stream),
           "rcdm_pack_conv3x3_wino")


def conv3x3_wino(desc, x, U, bias, rowvec, residual, out, ws_ptr, ws_bytes, x2=0, W2=0, gn=None, gn_stat=0, gn_gamma=0, gn_beta=0,
                 gn_out=None, gn_out_partial=0, stream=None):
    """rcdm_conv3x3_wino: gn = GroupNormDesc of the norm (+ SiLU) the input transform applies to x (None: x as it is);
    gn_out = GroupNormDesc of the norm that reads `out` next, its per-tile partial statistics go to gn_out_partial."""
    _check(load().rcdm_conv3x3_wino(C.byref(desc), C.byref(gn) if gn is not None else None, gn_stat, gn_gamma, gn_beta, x, x2, U, W2,
                                    bias, rowvec, residual, out, ws_ptr, ws_bytes, C.byref(gn_out) if gn_out is not None else None,
                                    gn_out_partial, stream_ptr() if stream is None else stream),
           "rcdm_conv3x3_wino")


def groupnorm_apply(desc, x, stat, gamma, beta, y, stream=None):
    _check(load().rcdm_groupnorm_apply(C.byref(desc), x, stat, gamma, beta, y, stream_ptr() if stream is None else stream),
           "rcdm_groupnorm_apply")


def groupnorm_finalize(samples, groups, splits, eps, partial, stat, stream=None):
    _check(load().rcdm_groupnorm_finalize(samples, groups, splits, eps, partial, stat, stream_ptr() if stream is None else stream),
           "rcdm_groupnorm_finalize")


def pack_conv3x3(w, c_out, c_in, cin_pad, dst, stream=None):
    _check(load().rcdm_pack_conv3x3(w, c_out, c_in, cin_pad, dst, stream_ptr() if stream is None else stream),
           "rcdm_pack_conv3x3")


def pack_geglu_rows(w, bias, n_out, K, w_dst, bias_dst, stream=None):
    _check(load().rcdm_pack_geglu_rows(w, bias, n_out, K, w_dst, bias_dst,
                                       stream_ptr() if stream is None else stream), "rcdm_pack_geglu_rows")


class Graph:
    """One captured hipGraph (rcdm_graph_*).  Capture on torch's current stream."""

    def __init__(self):
        self.exec = C.c_void_p(0)
        self.stream = None

    def begin(self):
        self.stream = stream_ptr()
        _check(load().rcdm_graph_begin_capture(self.stream), "rcdm_graph_begin_capture")

    def end(self):
        _check(load().rcdm_graph_end_capture(self.stream, C.byref(self.exec)), "rcdm_graph_end_capture")

    def launch(self, stream=None):
        _check(load().rcdm_graph_launch(self.exec, stream_ptr() if stream is None else stream), "rcdm_graph_launch")

    def __del__(self):
        try:
            if self.exec and _lib is not None:
                _lib.rcdm_graph_destroy(self.exec)
        except Exception:
            pass


class Comm:
    """An RCCL communicator behind the C-ABI (rcdm_comm_*): broadcast / all-gather of raw device bytes on the current
    (or a given) HIP stream.  `unique_id()` on one rank, ship the 128 bytes to the others, then every rank constructs
    Comm(id, nranks, rank) — a collective call."""

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(load().rcdm_comm_unique_id(buf), "rcdm_comm_unique_id")
        return buf.raw

    def __init__(self, unique_id, nranks, rank):
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        self.nranks, self.rank = int(nranks), int(rank)
        self.comm = C.c_void_p(0)
        _check(load().rcdm_comm_create(C.create_string_buffer(bytes(unique_id), 128), self.nranks, self.rank,
                                       C.byref(self.comm)), "rcdm_comm_create")

    def bcast(self, ptr, nbytes, root=0, stream=None):
        _check(load().rcdm_bcast(self.comm, ptr, nbytes, root, stream_ptr() if stream is None else stream), "rcdm_bcast")

    def allgather(self, send_ptr, recv_ptr, bytes_per_rank, stream=None):
        _check(load().rcdm_allgather(self.comm, send_ptr, recv_ptr, bytes_per_rank,
                                     stream_ptr() if stream is None else stream), "rcdm_allgather")

    def close(self):
        """Destroy the communicator (collective: every rank of it should close).  Call it explicitly, while the peers
        are alive and after every graph that captured a collective on it is gone."""
        if self.comm:
            load().rcdm_comm_destroy(self.comm)
            self.comm = C.c_void_p(0)

    def __del__(self):
        # never at interpreter teardown: ncclCommDestroy can block on a peer that has already exited
        try:
            if _lib is not None and not sys.is_finalizing():
                self.close()
        except Exception:
            pass


class Event:
    def __init__(self):
        self.ev = C.c_void_p(0)
        _check(load().rcdm_event_create(C.byref(self.ev)), "rcdm_event_create")

    def record(self, stream=None):
        _check(load().rcdm_event_record(self.ev, stream_ptr() if stream is None else stream), "rcdm_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        _check(load().rcdm_event_elapsed_ms(self.ev, stop.ev, C.byref(ms)), "rcdm_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self.ev and _lib is not None:
                _lib.rcdm_event_destroy(self.ev)
        except Exception:
            pass
