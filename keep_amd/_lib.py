"""ctypes binding of libkeep_hip.so (the C ABI in include/keep_hip.h).

The north star asks for "a thin C-ABI cffi layer"; cffi is not installed in this image, so the same
ABI is bound with ctypes (argument-for-argument what a cffi ``ffi.cdef`` of keep_hip.h would give).
There is deliberately no fallback: if the shared object is missing or does not load, every entry
point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KEEP_HIP_LIB") or os.path.join(_HERE, "libkeep_hip.so")     # override: A/B builds only

KEEP_OK, KEEP_EINVAL, KEEP_ESTATE, KEEP_EKEY, KEEP_EHIP, KEEP_EUNSUPPORTED, KEEP_ENOMEM = 0, -1, -2, -3, -4, -5, -6
PIX_F32, PIX_F16, PIX_BF16, PIX_U8_HWC = 0, 1, 2, 3
SIM_RAW, SIM_ARGMAX, SIM_SOFTMAX, SIM_SOFTMAX_F16, SIM_TOP2SCORE = 0, 1, 2, 3, 4
PREC_FP16, PREC_STRICT, PREC_COMP = 0, 1, 2
ATTN_PLAIN, ATTN_SPLIT, ATTN_SPLIT_COMPQKV, ATTN_COMPQKV, ATTN_PROJ_CLS, ATTN_COMPQKV_PROJ_CLS = 0, 1, 2, 3, 4, 5        # keep_set_block_precision: attention side of a ViT block
MLP_PLAIN, MLP_SPLIT, MLP_COMP, MLP_COMP_W, MLP_CLS = 0, 1, 2, 3, 4          # ... and its MLP

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes); must list every symbol keep_hip.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "keep_version": (C.c_char_p, []),
    "keep_create": (_i32, [_i32, C.POINTER(_vp)]),
    "keep_destroy": (_i32, [_vp]),
    "keep_last_error": (C.c_char_p, [_vp]),
    "keep_load_warnings": (C.c_char_p, [_vp]),
    "keep_load_tensor": (_i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    "keep_finalize_weights": (_i32, [_vp]),
    "keep_vit_depth": (_i32, [_vp]),
    "keep_bert_layers": (_i32, [_vp]),
    "keep_set_option": (_i32, [_vp, C.c_char_p, C.c_double]),
    "keep_get_option": (C.c_double, [_vp, C.c_char_p]),
    "keep_set_block_precision": (_i32, [_vp, _i32, _i32, _i32]),
    "keep_get_block_precision": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "keep_calibrate_bias": (_i32, [_vp, _vp, _i32, _i64, _vp]),
    "keep_reserve": (_i32, [_vp, _i64, _i64, _i64]),
    "keep_workspace_bytes": (_i64, [_vp]),
    "keep_encode_image": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "keep_encode_text": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "keep_resize_crop_u8": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _i64, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp]),
    "keep_token_error": (_i32, [_vp, _vp]),
    "keep_token_error_async": (_i32, [_vp, _vp, _vp]),
    "keep_similarity": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp, _vp, _vp]),
    "keep_classify": (_i32, [_vp, _vp, _i32, _i64, _vp, _i64, _f32, _f32, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "keep_prompt_scores": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "keep_group_argmax": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "keep_retrieval_rank": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "keep_refine": (_i32, [_vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    "keep_profile_enable": (_i32, [_vp, C.c_char_p]),
    "keep_profile_read": (_i32, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "keep_profile_reset": (_i32, [_vp]),
    "keep_op_linear": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp]),
    "keep_op_mlp": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "keep_op_attention": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp]),
    "keep_op_layernorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp]),
    "keep_op_sgemm": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp, _vp]),
    "keep_op_l2norm": (_i32, [_vp, _vp, _i64, _i64, _vp]),
    "keep_clock_probe": (_i32, [_vp, _i32, _vp, _vp]),
    "keep_mfma_probe": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(C.c_double), _vp]),
    "keep_debug_read": (_i32, [_vp, _vp, _i64]),
}

_lib: Optional[C.CDLL] = None


class KeepHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libkeep_hip.so; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KeepHipError(f"{LIB_PATH} not found: build it with `python -m keep_amd.build` "
                           "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(handle, rc: int, what: str = "") -> None:
    if rc == KEEP_OK:
        return
    msg = load().keep_last_error(handle).decode() if handle else ""
    text = f"{what}: {msg}" if what else msg
    if rc in (KEEP_EINVAL, KEEP_EUNSUPPORTED):
        raise ValueError(text)
    if rc == KEEP_EKEY:
        raise KeyError(text)
    raise KeepHipError(f"{text} (code {rc})")
