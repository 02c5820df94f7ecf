"""CPU restatement (numpy, test infrastructure only) of the resize step of the reference's val transform:

    T.RandomResize([768], max_size=768)          dataloader_coco.py:286-292
      -> augmentation.resize(image, target, 768, 768)   segment_anything/utils/augmentation.py:77-133
      -> torchvision.transforms.functional.resize(PIL image, (oh, ow))  = PIL.Image.resize((ow, oh), BILINEAR)

The arithmetic lives in a third-party dependency that is not under /root/reference: Pillow (pinned 9.4.0 in
requirements.txt:17, 11.3.0 in uv.lock:558; 12.2.0 is importable in the build container), src/libImaging/Resample.c,
8-bit path: an antialiased separable triangle filter with support = scale, coefficients normalised in double and
converted to 22-bit fixed point, horizontal pass then vertical pass with an 8-bit intermediate image, each output
clip8((sum + 2^21) >> 22).  PINNED by tests/golden/resize_pil.npz, which oracle/gen_golden.py --only resize produced by
calling PIL itself (tests/test_oracle_small.py::test_pil_resize_restatement).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def get_size_with_aspect_ratio(image_size: Tuple[int, int], size: int, max_size: int | None = None) -> Tuple[int, int]:
    """augmentation.py:80-99.  image_size = (w, h); returns (oh, ow)."""
    w, h = image_size
    if max_size is not None:
        min_original_size = float(min((w, h)))
        max_original_size = float(max((w, h)))
        if max_original_size / min_original_size * size > max_size:
            size = int(round(max_size * min_original_size / max_original_size))
    if (w <= h and w == size) or (h <= w and h == size):
        return (h, w)
    if w < h:
        ow = size
        oh = int(size * h / w)
    else:
        oh = size
        ow = int(size * w / h)
    return (oh, ow)


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter over the whole
    axis.  Returns (bounds [out,2] int32 = (first input index, count), coeffs [out, ksize] int32, ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, dtype=np.float64)
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
        ww = w.sum() if xmax else 0.0        # Resample.c accumulates in order; a float64 sum of <= 2*support+1 terms
        # (sequential accumulation, as the C loop does)
        acc = 0.0
        for x in range(xmax):
            acc += w[x]
        ww = acc
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    """One separable pass over `axis` of an (H, W, C) uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out_size = bounds.shape[0]
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """(H, W, C) uint8 -> (oh, ow, C) uint8 as PIL.Image.resize((ow, oh), BILINEAR) (horizontal pass first; a pass whose
    size does not change is skipped, as ImagingResample does)."""
    h, w = img.shape[:2]
    out = img
    if ow != w:
        b, k, _ = precompute_coeffs(w, ow)
        out = _pass(out, b, k, 1)
    if oh != h:
        b, k, _ = precompute_coeffs(h, oh)
        out = _pass(out, b, k, 0)
    return np.ascontiguousarray(out)


def val_transform_u8(img: np.ndarray, size: int = 768, max_size: int = 768) -> np.ndarray:
    """The val pipeline's resize (dataloader_coco.py:288) on an (H, W, 3) uint8 frame."""
    oh, ow = get_size_with_aspect_ratio((img.shape[1], img.shape[0]), size, max_size)
    return resize_bilinear_u8(img, oh, ow)
