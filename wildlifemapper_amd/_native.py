"""ctypes binding of libwm_hip.so (include/wm_hip.h).  No fallback: if the
library is missing or a call fails this module raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch  # noqa: F401  (must be imported first so both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# WM_HIP_LIB: another build of the same library (A/B runs of two builds in one gpurun call); still no fallback of any kind
LIB_PATH = os.environ.get("WM_HIP_LIB") or os.path.join(_HERE, "libwm_hip.so")

PREC_BF16, PREC_FP16, PREC_FP8 = 0, 1, 2
PREC_BY_NAME = {"bf16": PREC_BF16, "fp16": PREC_FP16, "f16": PREC_FP16, "fp8": PREC_FP8}
NUM_QUERIES, NUM_LOGITS = 51, 8
KCLASS_NAMES = ("gemm16", "attn_window", "attn_global", "layernorm", "other")
GEMM_VARIANTS = ("v1_128", "v2_160", "v2_128", "v3_lockstep", "v3_conv3x3", "v5_320", "v5_320_res", "v5_256", "v5_256_res",
                 "v5_320_lnf", "v5_256_lnf", "fp8_320", "fp8_256", "v5_320_foldp", "v5_256_foldp", "v5_320_split", "v5_256_split", "v3_patch_embed", "fp8_256_planes")
FLAG_CONF, FLAG_SCORE, FLAG_NMS, FLAG_MERGED = 1, 2, 4, 8
CFG_FUSE_LN = 1
CFG_FOLD_LN = 2
CFG_FOLD_LN_BF16 = 4
ABI_VERSION = 4            # include/wm_hip.h WM_ABI_VERSION this binding was written for
FP8_QKV, FP8_PROJ, FP8_MLP, FP8_ALL = 1, 2, 4, 7
GEMM_W_PACKED, GEMM_A_PACKED, GEMM_OUT_PACKED, LAYOUT_PACKED = 0x1000, 0x2000, 0x4000, 0x100
GEMM32_SPLIT = 0x100          # wm_op_gemm32: act | GEMM32_SPLIT = the fp16-split form the decoder runs
SAT_NAMES = ("layernorm_out", "qkv", "attention_out", "mlp_hidden", "last_block_16")


class WmConfig(C.Structure):
    _fields_ = [("embed_dim", C.c_int32), ("depth", C.c_int32), ("num_heads", C.c_int32),
                ("num_global", C.c_int32), ("global_attn_indexes", C.c_int32 * 8),
                ("max_batch", C.c_int32), ("precision", C.c_int32), ("flags", C.c_int32), ("fp8_gemms", C.c_int32), ("reserved", C.c_int32 * 2)]


class WmBoxRecord(C.Structure):
    _fields_ = [("box", C.c_float * 4), ("score", C.c_float), ("label", C.c_int32),
                ("flags", C.c_int32), ("nms_rank", C.c_int32)]


class WmKclassStat(C.Structure):
    _fields_ = [("launches", C.c_int64), ("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/wm_hip.h declares: name -> (restype, argtypes)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    "wm_last_error": (C.c_char_p, []),
    "wm_abi_version": (_I, []),
    "wm_create": (_I, [C.POINTER(WmConfig), _I, C.POINTER(_P)]),
    "wm_destroy": (_I, [_P]),
    "wm_load_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    "wm_finalize_weights": (_I, [_P]),
    "wm_preprocess_u8": (_I, [_P, _P, _I, _I, _I, _P]),
    "wm_preprocess_u8_resized": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_resized_size": (_I, [_I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "wm_debug_resize_coeffs": (_I, [_I, _I, C.POINTER(_I), C.POINTER(_I), _I, C.POINTER(_I)]),
    "wm_tile_frame_u8": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "wm_merge_tiles_nms": (_I, [_P, _P, _I, _F, _P, _P]),
    "wm_hfc_fft": (_I, [_P, _P, _P, _I, _P]),
    "wm_encoder_forward": (_I, [_P, _P, _P, _P, _I, _P]),
    "wm_decoder_forward": (_I, [_P, _P, _P, _P, _I, _P]),
    "wm_postprocess_nms": (_I, [_P, _P, _P, _P, _F, _F, _F, _P, _I, _P]),
    "wm_forward": (_I, [_P, _P, _P, _P, _P, _P, _I, _P]),
    "wm_set_tap": (_I, [_P, _I]),
    "wm_read_tap": (_I, [_P, _P, _I, _P]),
    "wm_profile_enable": (_I, [_P, _I]),
    "wm_profile_reset": (_I, [_P]),
    "wm_profile_read": (_I, [_P, C.POINTER(WmKclassStat)]),
    "wm_debug_gemm_variant_counts": (_I, [C.POINTER(_L), _I]),
    "wm_debug_reset_gemm_variant_counts": (_I, []),
    "wm_debug_saturation_enable": (_I, [_P, _I]),
    "wm_debug_saturation_read": (_I, [_P, C.POINTER(_L), _I, _I, _P]),
    "wm_op_cvt_f32_to_16": (_I, [_P, _P, _L, _I, _P]),
    "wm_op_cvt_16_to_f32": (_I, [_P, _P, _L, _I, _P]),
    "wm_op_gemm16": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_gemm16_takes_packed": (_I, [_I, _I, _I]),
    "wm_op_pack16": (_I, [_P, _P, _L, _I, _P]),
    "wm_op_ln_stats16": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "wm_op_ln_stats16_split": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P]),
    "wm_op_gemm16_split": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_stream_merge": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "wm_op_unpack16": (_I, [_P, _P, _L, _I, _P]),
    "wm_stream_overflow": (_I, [_P, _I]),
    "wm_op_fold_weight16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "wm_op_gemm16_folded": (_I, [_P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_gemm16_stats": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_gemm8": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_stream_rows": (_I, [_P, _P, _P, _L, _I, _I, _I, _P]),
    "wm_op_gemm8_planes": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "wm_op_layernorm_fp8_plane": (_I, [_P, _P, _P, _F, _P, _L, _I, _I, _P]),
    "wm_op_cvt_f32_to_fp8": (_I, [_P, _P, _L, _P]),
    "wm_op_conv3x3_16": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "wm_op_patch_embed16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "wm_op_gemm32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "wm_op_layernorm": (_I, [_P, _P, _P, _F, _P, _P, _L, _I, _I, _P]),
    "wm_op_encoder_attention": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_encoder_attention_qkv": (_I, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "wm_op_mha16": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "wm_op_mha32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load libwm_hip.so once.  Raises RuntimeError if it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)       # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        got = l.wm_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} reports ABI version {got}, this binding was written for {ABI_VERSION}: "
                               "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _lib = l
    return _lib


def gemm_variant_counts(reset: bool = False) -> dict:
    """{variant name: launches since the last reset} (wm_debug_gemm_variant_counts)."""
    arr = (C.c_int64 * len(GEMM_VARIANTS))()
    check(lib().wm_debug_gemm_variant_counts(arr, len(GEMM_VARIANTS)))
    if reset:
        lib().wm_debug_reset_gemm_variant_counts()
    return {n: int(arr[i]) for i, n in enumerate(GEMM_VARIANTS)}


def check(status: int) -> None:
    if status != 0:
        msg = lib().wm_last_error()
        raise RuntimeError("wm_hip: " + (msg.decode() if msg else f"status {status}"))


def ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device: Optional[torch.device] = None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; the HIP path needs a ROCm device tensor "
                           "(there is no CPU fallback in wildlifemapper_amd)")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f"{what}: expected a contiguous float32 tensor, got {t.dtype}, contiguous={t.is_contiguous()}")
