"""Input pipeline on the GPU: uint8 HWC frames -> the model's (B,3,1024,1024) fp32 input."""
from __future__ import annotations

import torch

from . import _native as N


def resized_size(height: int, width: int, size: int = 768, max_size: int = 768):
    """(oh, ow) of the val transform's resize (augmentation.py:80-99), from the library (wm_resized_size; host-only call)."""
    import ctypes as C
    oh, ow = C.c_int(), C.c_int()
    N.check(N.lib().wm_resized_size(height, width, size, max_size, C.byref(oh), C.byref(ow)))
    return oh.value, ow.value


def tiles_from_u8(images: torch.Tensor, resize=None) -> torch.Tensor:
    """images: (B,h,w,3) uint8 on a ROCm device.  Returns ToTensor + ImageNet-normalised tiles, top-left aligned on a zero
    1024x1024 canvas (utils/misc.py:46-67).  resize=None: h, w <= 1024, no resampling (wm_preprocess_u8).
    resize=(size, max_size), e.g. (768, 768) as the val pipeline (dataloader_coco.py:288): frames of any size are first
    resampled with PIL's bilinear arithmetic (wm_preprocess_u8_resized); the content occupies resized_size(h, w, ...)."""
    if not images.is_cuda or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3:
        raise RuntimeError(f"tiles_from_u8: expected a (B,h,w,3) uint8 ROCm tensor, got {tuple(images.shape)} {images.dtype} on {images.device}")
    images = images.contiguous()
    B, h, w, _ = images.shape
    out = torch.empty((B, 3, 1024, 1024), device=images.device, dtype=torch.float32)
    with torch.cuda.device(images.device):
        if resize is None:
            N.check(N.lib().wm_preprocess_u8(N.ptr(images), N.ptr(out), B, h, w, N.stream_ptr(images.device)))
        else:
            size, max_size = resize
            N.check(N.lib().wm_preprocess_u8_resized(N.ptr(images), N.ptr(out), B, h, w, int(size), int(max_size or 0), N.stream_ptr(images.device)))
    return out
