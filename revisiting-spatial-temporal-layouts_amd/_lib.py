"""ctypes binding of libstlt_hip.so (include/stlt_hip.h).  No fallback: if the library is missing the import of
any compute entry point raises, loudly."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STLT_HIP_LIB") or os.path.join(HERE, "libstlt_hip.so")  # env override: A/B experiments only

K_NAMES = ("embed", "gemm", "attn_spatial", "attn_temporal", "add_layernorm", "frames_embed", "gather_last", "ln_bwd", "attn_bwd",
           "gelu", "embed_bwd", "optim", "misc", "mhsa_fused", "mhsa_fused_spatial")
FLAG_CLS_ONLY_LAST_SPATIAL = 1
FLAG_LAST_ROW_ONLY_TEMPORAL = 2
FLAG_SKIP_PADDING = 4
FLAG_TRAIN_UPPER_ONLY = 8
FLAG_TRAIN_LOWER_ONLY = 16
FLAG_TRAIN_BACKBONE = 32
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2

_f = C.POINTER(C.c_float)
_i64 = C.POINTER(C.c_int64)
_u8 = C.POINTER(C.c_uint8)
_vp = C.c_void_p


class OptChunk(C.Structure):
    _fields_ = [("param", C.c_void_p), ("flat_offset", C.c_int64), ("n", C.c_int32), ("weight_decay", C.c_float)]


class WtEntry(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wt", C.c_void_p), ("n_out", C.c_int64), ("k_in", C.c_int64)]


class ProfLaunch(C.Structure):
    _fields_ = [("kid", C.c_int), ("kernels", C.c_int), ("us", C.c_float), ("reserved", C.c_float), ("flops", C.c_double), ("bytes", C.c_double),
                ("note", C.c_char * 160)]


class WgradItem(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("n_out", C.c_int64), ("x", C.c_void_p), ("k_in", C.c_int64), ("rows", C.c_int64), ("g_w", C.c_void_p)]


class LayerParams(C.Structure):
    _fields_ = [(n, _vp) for n in (
        "in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
        "norm1_w", "norm1_b", "norm2_w", "norm2_b")]


class Params(C.Structure):
    _fields_ = (
        [(n, C.c_int64) for n in ("d", "H", "n_categories", "n_spatial", "n_temporal", "n_classes", "n_positions")]
        + [("ln_eps", C.c_float)]
        + [(n, _vp) for n in ("cat_emb", "box_w", "box_b", "score_w", "score_b", "emb_ln_w", "emb_ln_b",
                              "pos_emb", "type_emb", "frames_ln_w", "frames_ln_b")]
        + [("spatial", C.POINTER(LayerParams)), ("temporal", C.POINTER(LayerParams))]
        + [(n, _vp) for n in ("fc1_w", "fc1_b", "head_ln_w", "head_ln_b", "fc2_w", "fc2_b")]
    )


class AttnBlockParams(C.Structure):
    _fields_ = [(n, _vp) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln_w", "ln_b")]


class FfnBlockParams(C.Structure):
    _fields_ = [(n, _vp) for n in ("lin1_w", "lin1_b", "lin2_w", "lin2_b", "ln_w", "ln_b")]


class CrossModalParams(C.Structure):
    _fields_ = [("cross_attn", AttnBlockParams), ("layout_attn", AttnBlockParams), ("appearance_attn", AttnBlockParams),
                ("appearance_ffn", AttnBlockParams), ("layout_ffn", FfnBlockParams)]


class HeadParams(C.Structure):
    _fields_ = [(n, _vp) for n in ("fc1_w", "fc1_b", "ln_w", "ln_b", "fc2_w", "fc2_b")]


class CafParams(C.Structure):
    _fields_ = [("layout", Params), ("feat_channels", C.c_int64), ("app_tokens", C.c_int64), ("proj_w", _vp), ("proj_b", _vp),
                ("cls_token", _vp), ("pos_embed", _vp), ("n_app_layers", C.c_int64), ("app_layers", C.POINTER(LayerParams)),
                ("n_fusion", C.c_int64), ("fusion", C.POINTER(CrossModalParams)), ("fusion_head", HeadParams),
                ("layout_head", HeadParams), ("appearance_head", HeadParams)]


class Inputs(C.Structure):
    _fields_ = [("B", C.c_int64), ("T", C.c_int64), ("N", C.c_int64)] + [
        (n, _vp) for n in ("categories", "boxes", "scores", "kpm_boxes", "frame_types", "kpm_frames", "lengths")] + [
        ("n_real_tokens", C.c_int64), ("n_real_frames", C.c_int64)]


# symbol -> (restype, argtypes); the not-gpu tests check that every one of these is exported
SIGNATURES = {
    "stlt_version": (C.c_int, []),
    "stlt_last_error": (C.c_char_p, []),
    "stlt_embed_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float,
                                 C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_linear_fwd": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                  C.c_int, _vp]),
    "stlt_linear_small_fwd": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, _vp]),
    "stlt_linear_small_choice": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "stlt_input_grad_small": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp]),
    "stlt_input_grad_small_choice": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "stlt_set_gemm_small_tiles": (C.c_int, [C.c_int]),
    "stlt_set_train_side_stream": (C.c_int, [C.c_int]),
    "stlt_get_train_side_stream": (C.c_int, []),
    "stlt_ctx_create": (C.c_int, [C.POINTER(_vp)]),
    "stlt_ctx_destroy": (C.c_int, [_vp]),
    "stlt_ctx_dw_defer": (C.c_int, [_vp, C.c_int]),
    "stlt_ctx_dw_pending": (C.c_int, [_vp]),
    "stlt_ctx_dw_flush": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "stlt_ctx_wt_refresh": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "stlt_ctx_wt_clear": (C.c_int, [_vp]),
    "stlt_ctx_wt_hits": (C.c_longlong, [_vp]),
    "stlt_gemm": (C.c_int, [C.c_int, C.c_int, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, C.c_int64,
                            C.c_int64, C.c_int64, C.c_int64, C.c_int, _vp]),
    "stlt_weight_grad_group": (C.c_int, [_vp, C.c_int, _vp]),
    "stlt_gemm_scratch_bytes": (C.c_size_t, []),
    "stlt_gemm_set_scratch": (C.c_int, [_vp, C.c_size_t]),
    "stlt_reduce_slabs": (C.c_int, [_vp, C.c_int64, C.c_int, _vp, C.c_int64, C.c_int, _vp]),
    "stlt_attn_core_fwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_mhsa_fused_fwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_fused_mhsa_active": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "stlt_fused_mhsa_used": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "stlt_mhsa_fused_fwd_ex": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_uint32,
                                         _vp, _vp, _vp]),
    "stlt_attn_cross_fwd": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_attn_ragged_fwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_add_layernorm_fwd": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_float, C.c_int64, C.c_int64,
                                         _vp, C.c_int64, _vp]),
    "stlt_frames_embed_fwd": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_int64, C.c_int64,
                                        C.c_int64, _vp, _vp]),
    "stlt_gather_last_fwd": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_collate_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp]),
    "stlt_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "stlt_backbone_forward": (C.c_int, [C.POINTER(Params), C.POINTER(Inputs), _vp, C.c_size_t, C.c_int, _vp, _vp]),
    "stlt_forward": (C.c_int, [C.POINTER(Params), C.POINTER(Inputs), _vp, C.c_size_t, C.c_int, _vp, _vp, _vp]),
    "stlt_caf_workspace_bytes": (C.c_size_t, [C.c_int64] * 7),
    "stlt_caf_forward": (C.c_int, [C.c_void_p, C.POINTER(Inputs), _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp]),
    "stlt_caf_forward_flags": (C.c_int, [C.c_void_p, C.POINTER(Inputs), _vp, _vp, C.c_size_t, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "stlt_train_tape_bytes": (C.c_size_t, [C.c_int64] * 6),
    "stlt_train_scratch_bytes": (C.c_size_t, [C.c_int64] * 5),
    "stlt_train_forward": (C.c_int, [C.POINTER(Params), C.POINTER(Inputs), _vp, C.c_size_t, _vp, C.c_float, C.c_uint64, C.c_int, _vp]),
    "stlt_train_backward": (C.c_int, [C.POINTER(Params), C.POINTER(Params), C.POINTER(Inputs), _vp, C.c_size_t, _vp,
                                      C.c_size_t, _vp, C.c_float, C.c_uint64, C.c_int, _vp, _vp]),
    "stlt_linear_bwd_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "stlt_linear_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "stlt_attn_fwd_dropout": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_float, C.c_uint64, C.c_uint32, _vp, _vp]),
    "stlt_attn_bwd": (C.c_int, [_vp, C.c_int64, _vp, _vp, C.c_int64, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                C.c_float, C.c_uint64, C.c_uint32, _vp, C.c_int64, _vp, _vp, C.c_int64, _vp]),
    "stlt_attn_core_bwd_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "stlt_attn_core_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, _vp, _vp, _vp,
                                     C.c_size_t, _vp]),
    "stlt_add_layernorm_bwd_scratch_bytes": (C.c_size_t, [C.c_int64]),
    "stlt_add_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_float, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "stlt_embed_fwd_train": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "stlt_embed_bwd_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64]),
    "stlt_embed_bwd": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "stlt_frames_embed_fwd_train": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "stlt_frames_embed_bwd_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "stlt_frames_embed_bwd": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, C.c_size_t, _vp]),
    "stlt_gelu_fwd": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "stlt_gelu_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "stlt_loss_fwd_bwd": (C.c_int, [_vp, _vp, C.c_int, C.c_int64, C.c_int64, C.c_float, _vp, _vp, _vp, _vp]),
    "stlt_dropout": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, _vp]),
    "stlt_relu_bwd": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "stlt_block_keep_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "stlt_block_work_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "stlt_attn_block_fwd_train": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_float, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int, C.c_int64, C.c_float,
                                            C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "stlt_attn_block_bwd_train": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_float, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int, C.c_int64, C.c_float,
                                            C.c_uint64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "stlt_ffn_block_fwd_train": (C.c_int, [_vp, C.c_int64, C.c_float, C.c_int, C.c_int, _vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, _vp, _vp, _vp,
                                           _vp, _vp, C.c_size_t, _vp]),
    "stlt_ffn_block_bwd_train": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_int, C.c_int, _vp, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "stlt_prof_take_gemm_flops": (C.c_double, []),
    "stlt_eval_topk": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, _vp]),
    "stlt_eval_max_clips": (C.c_int64, []),
    "stlt_eval_store_sigmoid": (C.c_int, [_vp, C.c_int64, _vp, C.c_int64, C.c_int64, _vp, _vp, C.c_int64, _vp]),
    "stlt_debug_buffer_bytes": (C.c_size_t, []),
    "stlt_set_gemm_split_bf16": (C.c_int, [C.c_int]),
    "stlt_eval_average_precision": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp]),
    "stlt_grad_norm": (C.c_int, [_vp, C.c_int64, C.c_float, _vp, _vp, _vp]),
    "stlt_adamw_step": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, _vp]),
    "stlt_prof_enable": (C.c_int, [C.c_int]),
    "stlt_prof_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "stlt_prof_launches": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int64)]),
    "stlt_debug_set_buffer": (C.c_int, [_vp]),
}

_lib = None


class StltHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises StltHipError when the .so has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StltHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
            "(or revisiting-spatial-temporal-layouts_amd/build.py). There is no CPU/PyTorch fallback for the STLT hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str):
    if code != 0:
        msg = load().stlt_last_error().decode("utf-8", "replace")
        raise StltHipError(f"{what} failed ({code}): {msg}")
