"""Input pipeline on the GPU: uint8 HWC frames -> the model's (B,3,1024,1024) fp32 input."""
from __future__ import annotations

import torch

from . import _native as N


def tiles_from_u8(images: torch.Tensor) -> torch.Tensor:
    """images: (B,h,w,3) uint8 on a ROCm device, h,w <= 1024.  Returns ToTensor + ImageNet-normalised tiles,
    top-left aligned on a zero 1024x1024 canvas (utils/misc.py:46-67), computed by wm_preprocess_u8."""
    if not images.is_cuda or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
        raise RuntimeError(f"tiles_from_u8: expected a (B,h,w,3) uint8 ROCm tensor, got {tuple(images.shape)} {images.dtype} on {images.device}")
    images = images.contiguous()
    B, h, w, _ = images.shape
    out = torch.empty((B, 3, 1024, 1024), device=images.device, dtype=torch.float32)
    N.check(N.lib().wm_preprocess_u8(N.ptr(images), N.ptr(out), B, h, w, N.stream_ptr(images.device)))
    return out
