"""ctypes binding of libyolohip.so (the C ABI declared in include/yolohip.h).

The product path has no fallback: if the shared library is missing, or a call fails, a RuntimeError
is raised.  Nothing here touches torch; callers pass raw device addresses (tensor.data_ptr()).
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
import threading
import weakref
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolohip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "yolohip.h")

c_fp = C.c_void_p   # device pointers travel as integers
i32, i64, f32 = C.c_int, C.c_int64, C.c_float


class YhOp(C.Structure):
    """Mirror of `struct yh_op` (include/yolohip.h)."""
    _fields_ = [("kind", C.c_int32), ("lane", C.c_int32), ("i", C.c_int32 * 19), ("f", C.c_float * 4),
                ("p", C.c_void_p * 12), ("l", C.c_int64 * 2)]


# op kinds, same order as the enum in yolohip.h
(OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_PACK_WEIGHTS, OP_CONV_FWD, OP_CONV_BWD_DATA, OP_CONV_BWD_WEIGHT,
 OP_COLSUM, OP_BN_FINALIZE, OP_BN_EVAL_COEF, OP_BN_SILU_FWD, OP_BN_SILU_BWD_REDUCE, OP_BN_SILU_BWD_APPLY,
 OP_MAXPOOL5_FWD, OP_MAXPOOL5_BWD, OP_MEMSET, OP_ADD_INT64, OP_PACK_WEIGHTS_MULTI, OP_PACK_FOLD_MULTI,
 OP_CONV_FWD_FUSED, OP_CONV_BWD_DATA_PAIR, OP_FORK, OP_JOIN, OP_WINO_WEIGHTS_MULTI, OP_CONV_WINO_FWD,
 OP_CONV_WINO_BWD_DATA, OP_CONV_WINO_BWD_WEIGHT, OP_CONV_PW_BWD_WEIGHT, OP_PW_PACK_MULTI, OP_CONV_PW_FWD,
 OP_CONV_PW_BWD_DATA, OP_CONV_STEM_FWD, OP_PACK_WEIGHTS_S2M, OP_CONV_BWD_DATA_S2M, OP_CONV_PW_FWD2, OP_NOP,
 OP_BF16_PACK_MULTI, OP_BF16_CONV_FWD, OP_BF16_CONV_BWD_DATA, OP_BF16_CONV_BWD_WEIGHT, OP_BF16_COLSUM, OP_BF16_BN_SILU_FWD,
 OP_BF16_BN_SILU_BWD_REDUCE, OP_BF16_BN_SILU_BWD_APPLY, OP_BF16_MAXPOOL5_FWD, OP_BF16_MAXPOOL5_BWD,
 OP_FOLD_OIHW_MULTI, OP_CONV_WINO_FWD_FUSED, OP_CONV_PW_FWD_FUSED, OP_CONV_NARROW, OP_CONV_NARROW_DGRAD_S2, OP_CONV_NARROW_BWD_WEIGHT, OP_BF16_CONV_NARROW, OP_BF16_CONV_NARROW_DGRAD_S2,
 OP_BF16_CONV_NARROW_BWD_WEIGHT, OP_SPPF_POOL3, OP_CONV_LAT_FWD_FUSED, OP_LAT_PACK_MULTI, OP_CONV_S2_FWD) = range(1, 59)

_P3 = C.c_void_p * 3
_I3 = C.c_int * 3
_PP = C.POINTER(C.c_void_p)     # HOST array of device pointers (None -> NULL)
_IP = C.POINTER(C.c_int)

_SIGS = {
    "yh_version": (i32, []),
    "yh_last_error": (C.c_char_p, []),
    "yh_nchw_to_nhwc": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_u8hwc_to_nhwc": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_nhwc_to_nchw": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_pack_weights": (i32, [c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_pack_weights_multi": (i32, [c_fp, i32, c_fp]),
    "yh_conv_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_fwd_fused": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_pack_fold_multi": (i32, [c_fp, i32, c_fp]),
    "yh_fold_oihw_multi": (i32, [c_fp, i32, c_fp]),
    "yh_conv_wino_fwd_fused": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_pw_fwd_fused": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_s2_ok": (i32, [i32, i32, i32, i32, i32]),
    "yh_conv_s2_blocks": (i32, [i32, i32, i32, i32]),
    "yh_conv_s2_fwd_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_lat_ok": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "yh_conv_lat_fwd_fused": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_lat_pack_multi": (i32, [c_fp, i32, c_fp]),
    "yh_conv_fwd_blocks": (i32, [i32, i32, i32, i32, i32, i32]),
    "yh_conv_bwd_data": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_wino_weights": (i32, [c_fp, c_fp, i32, i32, i32, i32, c_fp]),
    "yh_wino_weights_multi": (i32, [c_fp, i32, c_fp]),
    "yh_conv_wino_blocks": (i32, [i32, i32, i32]),
    "yh_conv_wino_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_wino_fwd_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_wino_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_fwd_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_bwd_weight_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_bwd_weight_prologue_ok": (i32, [i32, i32, i32, i32, i32, i32, i32]),
    "yh_conv_pw_prologue_ok": (i32, [i64, i32, i32]),
    "yh_conv_pw_fwd_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i64, i32, i32, c_fp]),
    "yh_conv_pw_x6_blocks": (i32, [i64, i32, i32]),
    "yh_conv_pw_fwd_x6": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i64, i32, i32, c_fp]),
    "yh_conv_pw_fwd2_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i64, i32, c_fp]),
    "yh_conv_pw_bwd_weight_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i64, i32, i32, c_fp]),
    "yh_conv_narrow_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_narrow_bwd_weight_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_wino_bwd_weight_act": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_wino_bwd_weight_ws": (i64, [i32, i32, i32, i32, i32]),
    "yh_conv_fwd_fused_splitk": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32,
                                       c_fp]),
    "yh_conv_fwd_fused_ws": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "yh_conv_bwd_data_s2m": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_pack_weights_s2m": (i32, [c_fp, c_fp, i32, i32, i32, c_fp]),
    "yh_conv_narrow_ok": (i32, [i32, i32, i32, i32]),
    "yh_conv_narrow_blocks": (i32, [i32, i32, i32, i32, i32]),
    "yh_conv_narrow_dgrad_s2_ok": (i32, [i32, i32]),
    "yh_conv_narrow_bwd_weight_ok": (i32, [i32, i32, i32, i32, i32]),
    "yh_bf16_conv_narrow": (i32, [c_fp, i32, c_fp, i32, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_narrow_dgrad_s2": (i32, [c_fp, i32, c_fp, i32, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_narrow_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_narrow_bwd_weight_ws": (i64, [i32, i32, i32, i32, i32, i32]),
    "yh_conv_narrow_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_narrow_dgrad_s2": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_narrow": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_pw_pack_multi": (i32, [c_fp, i32, c_fp]),
    "yh_conv_pw_blocks": (i32, [i64, i32, i32]),
    "yh_conv_pw_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i64, i32, i32, c_fp]),
    "yh_conv_pw_fwd2": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i64, i32, c_fp]),
    "yh_conv_pw_bwd_data": (i32, [c_fp, i32, c_fp, i32, i32, c_fp, i32, c_fp, i32, i64, i32, i32, c_fp]),
    "yh_conv_pw_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i64, i32, i32, c_fp]),
    "yh_conv_pw_bwd_weight_ws": (i64, [i64, i32, i32]),
    "yh_conv_wino_bwd_data": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_bwd_data_pair": (i32, [c_fp, i32, c_fp, i32, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_conv_bwd_weight_ws": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "yh_colsum": (i32, [c_fp, i32, i64, i32, c_fp, c_fp, c_fp]),
    "yh_colsum_ws": (i64, [i64, i32]),
    "yh_bn_finalize": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, c_fp, f32, f32, c_fp, i32, c_fp, c_fp]),
    "yh_bn_finalize_x": (i32, [c_fp, i32, i64, c_fp, c_fp, c_fp, c_fp, f32, f32, c_fp, i32, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "yh_bn_silu_bwd_reduce_acc": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp]),
    "yh_bf16_bn_silu_bwd_reduce_acc": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp]),
    "yh_bn_silu_bwd_apply_acc": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32, i64, i32, i32, i32, i32, c_fp]),
    "yh_bf16_bn_silu_bwd_apply_acc": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32, i64, i32, i32, i32, i32, c_fp]),
    "yh_bn_silu_fwd_res": (i32, [c_fp, i32, c_fp, c_fp, i32, c_fp, i32, c_fp, i32, i64, i32, i32, i32, i32, c_fp]),
    "yh_bn_eval_coef": (i32, [c_fp, c_fp, c_fp, c_fp, f32, c_fp, i32, c_fp]),
    "yh_bn_silu_fwd": (i32, [c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i64, i32, i32, i32, i32, c_fp]),
    "yh_bn_silu_bwd_reduce": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp]),
    "yh_bn_bwd_blocks": (i32, [i64, i32]),
    "yh_bn_silu_bwd_apply": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32,
                                   i64, i32, i32, i32, i32, c_fp]),
    "yh_maxpool5_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, c_fp]),
    "yh_maxpool5_bwd": (i32, [c_fp, i32, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_sppf_pool3_ok": (i32, [i32, i32]),
    "yh_sppf_pool3_fwd": (i32, [c_fp, i32, c_fp, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_yolo_loss": (i32, [_PP, _PP, _PP, C.POINTER(f32), _IP, i32, i32, f32, C.POINTER(f32), C.POINTER(f32), c_fp, c_fp, c_fp]),
    "yh_loss_ws": (i64, [_IP, i32]),
    "yh_yolo_loss_ex": (i32, [_PP, _PP, _PP, i32, _IP, C.POINTER(f32), _IP, i32, i32, f32, C.POINTER(f32), C.POINTER(f32), c_fp, c_fp,
                              c_fp]),
    "yh_bf16_pack_multi": (i32, [c_fp, i32, c_fp]),
    "yh_bf16_conv_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_blocks": (i32, [i64]),
    "yh_bf16_conv_stream_blocks": (i32, [i32, i32, i32, i32, i32, i32]),
    "yh_bf16_conv_stream_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_stream_bwd_data": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_fwd_blocks": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "yh_bf16_conv_bwd_data": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_bwd_weight": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_conv_bwd_weight_ws": (i64, [i32, i32, i32, i32, i32, i32, i32]),
    "yh_bf16_nchw_to_nhwc": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_u8hwc_to_nhwc": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_nhwc_to_nchw": (i32, [c_fp, c_fp, i32, i32, i32, i32, i32, i32, c_fp]),
    "yh_bf16_colsum": (i32, [c_fp, i32, i64, i32, c_fp, c_fp, c_fp]),
    "yh_bf16_bn_silu_fwd": (i32, [c_fp, i32, c_fp, c_fp, i32, c_fp, i32, i64, i32, i32, i32, i32, c_fp]),
    "yh_bf16_bn_silu_bwd_reduce": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i64, i32, i32, i32, i32, c_fp]),
    "yh_bf16_bn_silu_bwd_apply": (i32, [c_fp, i32, c_fp, i32, c_fp, c_fp, i32, c_fp, c_fp, c_fp, c_fp, i32, c_fp, i32, i32,
                                        i64, i32, i32, i32, i32, c_fp]),
    "yh_bf16_maxpool5_fwd": (i32, [c_fp, i32, c_fp, i32, c_fp, i32, i32, i32, i32, c_fp]),
    "yh_bf16_maxpool5_bwd": (i32, [c_fp, i32, c_fp, c_fp, i32, i32, i32, i32, i32, c_fp]),
    "yh_eval_counts": (i32, [_PP, _PP, C.POINTER(f32), _IP, i32, i32, f32, f32, f32, c_fp, c_fp]),
    "yh_decode": (i32, [c_fp, c_fp, C.POINTER(f32), i32, i32, i32, i32, f32, c_fp]),
    "yh_decode_bwd": (i32, [c_fp, c_fp, c_fp, C.POINTER(f32), i32, i32, i32, i32, f32, c_fp]),
    "yh_ciou": (i32, [c_fp, c_fp, c_fp, i64, f32, f32, c_fp, c_fp, c_fp]),
    "yh_candidates": (i32, [_PP, C.POINTER(f32), _IP, i32, f32, f32, f32, f32, f32, c_fp, c_fp, c_fp, c_fp, i32, c_fp,
                            c_fp, c_fp]),
    "yh_candidates_ws": (i64, [_IP]),
    "yh_assign_targets": (i32, [c_fp, c_fp, i32, i32, C.POINTER(f32), _IP, i32, i32, _PP, c_fp]),
    "yh_nms": (i32, [c_fp, c_fp, c_fp, c_fp, i32, C.c_double, i32, c_fp, c_fp, c_fp, c_fp]),
    "yh_nms_ws": (i64, [i32]),
    "yh_gather_detections": (i32, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, i32, c_fp, c_fp]),
    "yh_grad_sqnorm": (i32, [c_fp, i64, f32, c_fp, c_fp, c_fp]),
    "yh_sqnorm_ws": (i64, [i64]),
    "yh_adam_step": (i32, [c_fp, c_fp, c_fp, c_fp, i64, C.c_double, C.c_double, C.c_double, C.c_double, i32, f32, c_fp, f32, c_fp]),
    "yh_memset": (i32, [c_fp, i32, i64, c_fp]),
    "yh_add_int64": (i32, [c_fp, i64, c_fp]),
    "yh_run": (i32, [c_fp, C.POINTER(YhOp), i32, c_fp, C.POINTER(i32)]),
    "yh_create": (i32, [C.POINTER(c_fp)]),
    "yh_destroy": (i32, [c_fp]),
    "yh_context_set_overlap": (i32, [c_fp, i32]),
    "yh_context_info": (i32, [c_fp, C.POINTER(i32), C.POINTER(i32), C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp)]),
}

_lib = None


def declared_symbols() -> list:
    """Every function name include/yolohip.h declares (used by the CPU-side ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yh_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load libyolohip.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension must be built first "
                "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().yh_last_error().decode(errors="replace")
        raise RuntimeError(f"libyolohip {what} failed (code {rc}): {msg}")


def ptr3(tensors):
    """HOST array of three device pointers (None -> NULL)."""
    return _P3(*[(t.data_ptr() if t is not None else None) for t in tensors])


def int3(vals):
    return _I3(*[int(v) for v in vals])


def floats(vals):
    arr = (f32 * len(vals))(*[float(v) for v in vals])
    return arr


class Context:
    """Owner of one `yh_context` (side stream + fork/join events of yh_run).  One per (device, host thread) is
    handed out by `context_for`; create your own for an extra concurrent stream."""

    def __init__(self):
        h = c_fp()
        check(lib().yh_create(C.byref(h)), "create")
        self.handle = h

    def set_overlap(self, enable: bool):
        check(lib().yh_context_set_overlap(self.handle, int(bool(enable))), "context_set_overlap")

    def info(self) -> dict:
        dev, ov = i32(-2), i32(-2)
        side, fork, join = c_fp(), c_fp(), c_fp()
        check(lib().yh_context_info(self.handle, C.byref(dev), C.byref(ov), C.byref(side), C.byref(fork), C.byref(join)),
              "context_info")
        return {"device": dev.value, "overlap": ov.value, "side_stream": side.value, "fork_event": fork.value,
                "join_event": join.value}

    def close(self):
        """Destroy the native context (waits for its side stream).  Skipped while the interpreter is shutting down: the HIP
        runtime may already be gone, and the process is about to release everything anyway."""
        h, self.handle = self.handle, None
        if h is not None and _lib is not None and not sys.is_finalizing():
            _lib.yh_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# The default context of a host thread lives in thread-local storage: it is released with its thread (no leak under thread
# churn, and a recycled thread ident can never inherit another thread's context); an atexit hook closes the contexts that are
# still alive before torch / HIP tear down.
_tls = threading.local()
_live = weakref.WeakSet()


def context_for(device_index: int) -> Context:
    """The default execution context of this host thread for one device."""
    per_dev = getattr(_tls, "contexts", None)
    if per_dev is None:
        per_dev = _tls.contexts = {}
    ctx = per_dev.get(int(device_index))
    if ctx is None:
        ctx = per_dev[int(device_index)] = Context()
        _live.add(ctx)
    return ctx


@atexit.register
def _close_contexts():
    for ctx in list(_live):
        try:
            ctx.close()
        except Exception:
            pass


def set_overlap(enable: bool, device_index: int = 0):
    """Lanes on/off for this thread's default context of a device (bench instrumentation, A/B tests)."""
    context_for(device_index).set_overlap(enable)


def run_ops(ops, n: int, stream: int, ctx: "Context" = None):
    """yh_run on `stream`; ctx = None runs every op in list order on that stream (no side lane)."""
    failed = i32(-1)
    rc = lib().yh_run(ctx.handle if ctx is not None else None, ops, n, stream, C.byref(failed))
    if rc != 0:
        msg = lib().yh_last_error().decode(errors="replace")
        raise RuntimeError(f"libyolohip op #{failed.value} (kind {ops[failed.value].kind}) failed (code {rc}): {msg}")
