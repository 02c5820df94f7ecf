"""Helpers for the -m gpu tests: thin ctypes calls of the single-op C-ABI entry points."""
from __future__ import annotations

import ctypes as C

import torch

from wildlifemapper_amd import _native as N

PRECS = {"bf16": (N.PREC_BF16, torch.bfloat16), "fp16": (N.PREC_FP16, torch.float16)}


def dev():
    return torch.device("cuda:0")


def sp():
    return N.stream_ptr(dev())


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def max_rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def to16(x: torch.Tensor, prec: str) -> torch.Tensor:
    """fp32 cuda tensor -> 16-bit tensor through the library's own convert kernel."""
    code, dt = PRECS[prec]
    x = x.contiguous().float()
    out = torch.empty(x.shape, device=x.device, dtype=dt)
    N.check(N.lib().wm_op_cvt_f32_to_16(N.ptr(x), N.ptr(out), x.numel(), code, sp()))
    return out


def gemm16(a16, w16, bias=None, residual=None, res_mod=0, act=0, prec="bf16", want32=True, want16=False, layout=0):
    """layout: OR of N.GEMM_W_PACKED / GEMM_A_PACKED / GEMM_OUT_PACKED (operands in LDS-image order, see pack16)."""
    code, dt = PRECS[prec]
    M, K = a16.shape
    Nn = w16.shape[0]
    o32 = torch.empty((M, Nn), device=a16.device, dtype=torch.float32) if want32 else None
    o16 = torch.empty((M, Nn), device=a16.device, dtype=dt) if want16 else None
    N.check(N.lib().wm_op_gemm16(N.ptr(a16), N.ptr(w16), N.ptr(bias), N.ptr(residual), res_mod, N.ptr(o32), N.ptr(o16),
                                 M, Nn, K, act | layout, code, sp()))
    return o32, o16


def pack16(t: torch.Tensor) -> torch.Tensor:
    """[rows][K] 16-bit -> LDS-image order through the library's pack kernel (same shape, permuted content)."""
    out = torch.empty_like(t)
    N.check(N.lib().wm_op_pack16(N.ptr(t.contiguous()), N.ptr(out), t.shape[0], t.shape[1], sp()))
    return out


def _image_index(device):
    l = torch.arange(64, device=device)
    return l >> 2, (l & 3) ^ ((-(l >> 4)) & 3)


def pack16_torch(t: torch.Tensor) -> torch.Tensor:
    """The layout's definition restated with torch indexing (include/wm_hip.h): position l of a 16 x 32 block holds row
    l >> 2, chunk (l & 3) ^ ((-(l >> 4)) & 3)."""
    rows, K = t.shape
    row, ch = _image_index(t.device)
    return t.view(rows // 16, 16, K // 32, 4, 8)[:, row, :, ch, :].permute(1, 2, 0, 3).contiguous().view(rows, K)


def unpack16_torch(p: torch.Tensor) -> torch.Tensor:
    rows, K = p.shape
    row, ch = _image_index(p.device)
    out = torch.empty_like(p).view(rows // 16, 16, K // 32, 4, 8)
    out[:, row, :, ch, :] = p.view(rows // 16, K // 32, 64, 8).permute(2, 0, 1, 3)
    return out.view(rows, K)


def fold_bn(C):
    return 320 if C % 320 == 0 else 256


def ln_stats16(x, prec="fp16"):
    """Folded LayerNorm, standalone producer: (stats [rows, C / BN, 2], x16 in LDS-image order)."""
    code, dt = PRECS[prec]
    rows, Cc = x.shape
    stats = torch.empty((rows, Cc // fold_bn(Cc), 2), device=x.device, dtype=torch.float32)
    x16 = torch.empty((rows, Cc), device=x.device, dtype=dt)
    N.check(N.lib().wm_op_ln_stats16(N.ptr(x), N.ptr(stats), N.ptr(x16), rows, Cc, code, sp()))
    return stats, x16


def fold_weight16(w16, gamma, beta, bias, prec="fp16"):
    code, dt = PRECS[prec]
    Nn, K = w16.shape
    wf = torch.empty_like(w16)
    c1 = torch.empty(Nn, device=w16.device, dtype=torch.float32)
    c2 = torch.empty(Nn, device=w16.device, dtype=torch.float32)
    N.check(N.lib().wm_op_fold_weight16(N.ptr(w16), N.ptr(gamma), N.ptr(beta), N.ptr(bias), N.ptr(wf), N.ptr(c1), N.ptr(c2), Nn, K, code, sp()))
    return wf, c1, c2


def gemm16_folded(x16, wf, c1, c2, stats, eps, act=0, prec="fp16", out_packed=False):
    code, dt = PRECS[prec]
    M, K = x16.shape
    Nn = wf.shape[0]
    out = torch.empty((M, Nn), device=x16.device, dtype=dt)
    N.check(N.lib().wm_op_gemm16_folded(N.ptr(x16), N.ptr(wf), N.ptr(c1), N.ptr(c2), N.ptr(stats), eps, N.ptr(out), M, Nn, K,
                                        act | (N.GEMM_OUT_PACKED if out_packed else 0), code, sp()))
    return out


def gemm16_stats(a16, w16, bias, residual, prec="fp16", layout=0):
    """Folded LayerNorm, producing GEMM: (out32 = residual + a w^T + bias, x16 copy, stats)."""
    code, dt = PRECS[prec]
    M, K = a16.shape
    Nn = w16.shape[0]
    out = residual.clone()
    x16 = torch.empty((M, Nn), device=a16.device, dtype=dt)
    stats = torch.empty((M, Nn // fold_bn(Nn), 2), device=a16.device, dtype=torch.float32)
    N.check(N.lib().wm_op_gemm16_stats(N.ptr(a16), N.ptr(w16), N.ptr(bias), N.ptr(out), N.ptr(out), N.ptr(x16), N.ptr(stats), M, Nn, K, layout, code, sp()))
    return out, x16, stats


def ln_stats16_split(x, prec="fp16", rewrite=True):
    """Split stream, standalone producer: (stats, hi, lo in LDS-image order, x rounded to float(hi) + float(lo))."""
    code, dt = PRECS[prec]
    rows, Cc = x.shape
    stats = torch.empty((rows, Cc // fold_bn(Cc), 2), device=x.device, dtype=torch.float32)
    hi = torch.empty((rows, Cc), device=x.device, dtype=dt)
    lo = torch.empty((rows, Cc), device=x.device, dtype=torch.float16)
    xr = x.clone() if rewrite else None
    N.check(N.lib().wm_op_ln_stats16_split(N.ptr(x), N.ptr(stats), N.ptr(hi), N.ptr(lo), N.ptr(xr), rows, Cc, code, sp()))
    return stats, hi, lo, xr


def gemm16_split(a16, w16, bias, hi, lo, prec="fp16", layout=0):
    """Split stream, producing GEMM, in place on clones: (hi', lo', stats) with hi' + lo' = (hi + lo) + a w^T + bias."""
    code, dt = PRECS[prec]
    M, K = a16.shape
    Nn = w16.shape[0]
    hi2, lo2 = hi.clone(), lo.clone()
    stats = torch.empty((M, Nn // fold_bn(Nn), 2), device=a16.device, dtype=torch.float32)
    N.check(N.lib().wm_op_gemm16_split(N.ptr(a16), N.ptr(w16), N.ptr(bias), N.ptr(hi2), N.ptr(lo2), N.ptr(stats), M, Nn, K, layout, code, sp()))
    return hi2, lo2, stats


def stream_merge(hi, lo, prec="fp16"):
    code, dt = PRECS[prec]
    rows, Cc = hi.shape
    out = torch.empty((rows, Cc), device=hi.device, dtype=torch.float32)
    N.check(N.lib().wm_op_stream_merge(N.ptr(hi), N.ptr(lo), N.ptr(out), rows, Cc, code, sp()))
    return out


def stream_rows_split(x, prec="bf16"):
    """fp32 rows -> planes of rows (hi of type prec, lo fp16; columns at plane_pos): the fp8 blocks' stream."""
    code, dt = PRECS[prec]
    hi = torch.empty(x.shape, device=x.device, dtype=dt)
    lo = torch.empty(x.shape, device=x.device, dtype=torch.float16)
    N.check(N.lib().wm_op_stream_rows(N.ptr(x), N.ptr(hi), N.ptr(lo), x.shape[0], x.shape[1], code, 0, sp()))
    return hi, lo


def plane_pos(C, device):
    """Position of column c inside a row of the fp8 blocks' stream planes (include/wm_hip.h wm_op_stream_rows)."""
    c = torch.arange(C, device=device)
    return (c & ~255) + ((c >> 5) & 1) * 128 + ((c >> 6) & 3) * 32 + (c & 31)


def stream_rows_merge(hi, lo, prec="bf16"):
    code, dt = PRECS[prec]
    out = torch.empty(hi.shape, device=hi.device, dtype=torch.float32)
    N.check(N.lib().wm_op_stream_rows(N.ptr(out), N.ptr(hi), N.ptr(lo), hi.shape[0], hi.shape[1], code, 1, sp()))
    return out


def gemm8_planes(a8, w8, wscale, bias, hi, lo, prec="bf16"):
    """(hi, lo) += (a w^T) * wscale + bias on clones."""
    code, dt = PRECS[prec]
    M, K = a8.shape
    hi2, lo2 = hi.clone(), lo.clone()
    N.check(N.lib().wm_op_gemm8_planes(N.ptr(a8), N.ptr(w8), N.ptr(wscale), N.ptr(bias), N.ptr(hi2), N.ptr(lo2), M, w8.shape[0], K, code, sp()))
    return hi2, lo2


def layernorm_fp8(x, g, b, eps):
    """The blocks' LayerNorm with e4m3 output (fp32 rows in)."""
    rows, Cc = x.shape
    out = torch.empty((rows, Cc), device=x.device, dtype=torch.uint8)
    N.check(N.lib().wm_op_layernorm(N.ptr(x), N.ptr(g), N.ptr(b), eps, None, N.ptr(out), rows, Cc, N.PREC_FP8, sp()))
    return out


def layernorm_fp8_plane(hi, g, b, eps, prec="bf16"):
    """e4m3 LayerNorm of the hi plane, output in plane order."""
    code, dt = PRECS[prec]
    rows, Cc = hi.shape
    out = torch.empty((rows, Cc), device=hi.device, dtype=torch.uint8)
    N.check(N.lib().wm_op_layernorm_fp8_plane(N.ptr(hi), N.ptr(g), N.ptr(b), eps, N.ptr(out), rows, Cc, code, sp()))
    return out


def unpack16(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(t)
    N.check(N.lib().wm_op_unpack16(N.ptr(t.contiguous()), N.ptr(out), t.shape[0], t.shape[1], sp()))
    return out


def patch_embed16(img16, w16, bias, prec="fp16", want32=True):
    """16 x 16 / stride-16 patch embed as an implicit GEMM: img16 (B, Cin, 1024, 1024) 16-bit, w16 (N, Cin * 256) -> (B * 4096, N)."""
    code, dt = PRECS[prec]
    B, Cin = img16.shape[0], img16.shape[1]
    Nn = w16.shape[0]
    o32 = torch.empty((B * 4096, Nn), device=img16.device, dtype=torch.float32) if want32 else None
    o16 = torch.empty((B * 4096, Nn), device=img16.device, dtype=dt)
    N.check(N.lib().wm_op_patch_embed16(N.ptr(img16.contiguous()), N.ptr(w16), N.ptr(bias), N.ptr(o32), N.ptr(o16), B, Nn, Cin, code, sp()))
    return o32, o16


def gemm32(a, w, bias=None, residual=None, act=0, split=False):
    """split: the fp16-split form on the 16-bit matrix pipe (gemm32.h gemm32x3_kernel), else the fp32 MFMA."""
    M, K = a.shape
    Nn = w.shape[0]
    out = torch.empty((M, Nn), device=a.device, dtype=torch.float32)
    N.check(N.lib().wm_op_gemm32(N.ptr(a), N.ptr(w), N.ptr(bias), N.ptr(residual), N.ptr(out), M, Nn, K, act | (N.GEMM32_SPLIT if split else 0), sp()))
    return out


def layernorm(x, g, b, eps, prec="bf16", want32=True, want16=False, packed=False):
    code, dt = PRECS[prec]
    if packed:
        code |= N.LAYOUT_PACKED
    rows, Cc = x.shape
    o32 = torch.empty_like(x) if want32 else None
    o16 = torch.empty(x.shape, device=x.device, dtype=dt) if want16 else None
    N.check(N.lib().wm_op_layernorm(N.ptr(x), N.ptr(g), N.ptr(b), eps, N.ptr(o32), N.ptr(o16), rows, Cc, code, sp()))
    return o32, o16


def encoder_attention(qkv16, qkv_bias, rel_h, rel_w, batch, heads, hd, window, prec="bf16"):
    code, dt = PRECS[prec]
    out = torch.empty((batch * 4096, heads * hd), device=qkv16.device, dtype=dt)
    N.check(N.lib().wm_op_encoder_attention(N.ptr(qkv16), N.ptr(qkv_bias), N.ptr(rel_h), N.ptr(rel_w), N.ptr(out),
                                            batch, heads, hd, window, code, sp()))
    return out


def mha16(q, k, v, batch, heads, hd, nq, nk, prec="bf16"):
    code, dt = PRECS[prec]
    out = torch.empty((batch * nq, heads * hd), device=q.device, dtype=dt)
    N.check(N.lib().wm_op_mha16(N.ptr(q), q.shape[-1], N.ptr(k), k.shape[-1], N.ptr(v), v.shape[-1], N.ptr(out),
                                heads * hd, batch, heads, hd, nq, nk, code, sp()))
    return out


def mha32(q, k, v, heads):
    B, nq, Cc = q.shape
    nk = k.shape[1]
    out = torch.empty_like(q)
    N.check(N.lib().wm_op_mha32(N.ptr(q), N.ptr(k), N.ptr(v), N.ptr(out), B, heads, Cc // heads, nq, nk, sp()))
    return out


# ---- fp8 (OCP e4m3) helpers for the WM_PREC_FP8 tests ----
def e4m3_lut(device):
    """256-entry decode table of OCP e4m3fn (bias 7, 0x7f / 0xff = NaN, no infinity)."""
    import math
    vals = []
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = float("nan")
        elif e == 0:
            v = m * 2.0 ** -9
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        vals.append(-v if s else v)
    return torch.tensor(vals, dtype=torch.float32, device=device)


def to_fp8(x: torch.Tensor) -> torch.Tensor:
    """fp32 cuda tensor -> e4m3 bytes (uint8) through the library's convert kernel (unit scale, RNE, saturating)."""
    x = x.contiguous().float()
    out = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    N.check(N.lib().wm_op_cvt_f32_to_fp8(N.ptr(x), N.ptr(out), x.numel(), sp()))
    return out


def from_fp8(b: torch.Tensor) -> torch.Tensor:
    return e4m3_lut(b.device)[b.long()]


def quant_weight_fp8(w: torch.Tensor):
    """Per-output-channel e4m3 quantisation as the weight packer does it: (bytes, scale[N])."""
    sc = w.abs().amax(dim=1) / 448.0
    sc = torch.where(sc > 0, sc, torch.ones_like(sc))
    return to_fp8(w / sc[:, None]), sc.contiguous()


def gemm8(a8, w8, wscale, bias=None, residual=None, act=0, mode="f32", prec="bf16"):
    """mode: 'f32' (needs residual; returns fp32), '16' (16-bit out), '8' (e4m3 out)."""
    code, dt = PRECS[prec]
    M, K = a8.shape
    Nn = w8.shape[0]
    o32 = residual.clone() if mode == "f32" else None
    o16 = torch.empty((M, Nn), device=a8.device, dtype=dt) if mode == "16" else None
    o8 = torch.empty((M, Nn), device=a8.device, dtype=torch.uint8) if mode == "8" else None
    N.check(N.lib().wm_op_gemm8(N.ptr(a8), N.ptr(w8), N.ptr(wscale), N.ptr(bias), N.ptr(o32) if mode == "f32" else None, N.ptr(o32),
                                N.ptr(o16), N.ptr(o8), M, Nn, K, act, code, sp()))
    return {"f32": o32, "16": o16, "8": o8}[mode]
