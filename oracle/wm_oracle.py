"""CPU oracle for the WildlifeMapper inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch CPU ops, fp32) of the
reference algorithm on the path SURVEY.md §8a lists (A1-A20).  It is imported
only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as
the checker -- never by the product path in wildlifemapper_amd/, which must
fail loudly when the HIP library is missing.

Parity status: PINNED for everything that torch provides (encoder, HFC
adaptor, decoder, heads, FFT) by golden vectors generated here from the
reference's own `modeling` modules (oracle/gen_golden.py -> tests/golden/),
and (round 3) for PostProcess / box_cxcywh_to_xyxy by running the reference's own class / function bodies, taken from
build_sam.py:212-258 and utils/box_ops.py:9-13 by definition node (tests/golden/postprocess_ref.npz, bit-exact).
UNPINNED for the two torchvision functions on the path, because torchvision is
not installed in this image: `Grayscale` (network.py:41) and `ops.nms`
(visualize_prediction.py:154).  Those two are restated from their published
definitions and only checked against hand-worked cases.

Every function cites the reference file:line it follows (paths relative to
/root/reference/wildlifemapper/segment_anything unless noted).

`rnd` hook: every function that feeds a matrix product takes its operands
through `cfg.rnd` (identity by default).  Tests pass a bf16 round-trip there to
get a "bf16-operand emulation" of the HIP kernels, which separates kernel bugs
from precision effects.  With the default it is the plain fp32 reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _ident(t: Tensor) -> Tensor:
    return t


def bf16_round(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def fp16_round(t: Tensor) -> Tensor:
    return t.clamp(-65504.0, 65504.0).to(torch.float16).to(torch.float32)


def e4m3_round(t: Tensor) -> Tensor:
    """OCP e4m3fn, round to nearest even, saturating at +-448 (what v_cvt_pk_fp8_f32 behind a clamp does on gfx950)."""
    return t.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


@dataclass
class OracleCfg:
    embed_dim: int = 1280
    depth: int = 32
    num_heads: int = 16
    global_attn_indexes: Tuple[int, ...] = (7, 15, 23, 31)
    grid: int = 64
    patch: int = 16
    window: int = 14
    hfc_dim: int = 1024
    hfc_heads: int = 8
    dec_heads: int = 8
    dec_depth: int = 2
    num_queries: int = 51
    rnd: Callable[[Tensor], Tensor] = field(default=_ident)
    # WM_PREC_FP8 emulation (BASELINE.json configs[4]): the blocks' four projections take e4m3 activations (unit scale)
    # and e4m3 weights with one fp32 scale per output channel; attention runs on bf16 operands; set `rnd` to fp16_round
    # for the stem / HFC adaptor / neck, which always use fp16 operands on the GPU.
    block_fp8: bool = False

    @staticmethod
    def from_model_type(model_type: str, rnd: Callable[[Tensor], Tensor] = _ident) -> "OracleCfg":
        table = {  # build_sam.py:19-52
            "vit_h": (1280, 32, 16, (7, 15, 23, 31)),
            "vit_l": (1024, 24, 16, (5, 11, 17, 23)),
            "vit_b": (768, 12, 12, (2, 5, 8, 11)),
        }
        table["default"] = table["vit_h"]
        d, depth, heads, gidx = table[model_type]
        return OracleCfg(embed_dim=d, depth=depth, num_heads=heads, global_attn_indexes=gidx, rnd=rnd)


def as_torch_weights(sd: Dict[str, "np.ndarray | Tensor"]) -> Dict[str, Tensor]:
    out = {}
    for k, v in sd.items():
        out[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))
    return out


# ----------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------
def linear(x: Tensor, w: Tensor, b: Optional[Tensor], cfg: OracleCfg) -> Tensor:
    """y = x W^T + b with operands passed through cfg.rnd (fp32 accumulate)."""
    y = cfg.rnd(x) @ cfg.rnd(w).t()
    return y if b is None else y + b


def linear8(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """y = (e4m3(x) e4m3(W / s)^T) * s + b with s[n] = max_k |W[n, k]| / 448 (gemm8.h; packer in wm_api.hip)."""
    sw = w.abs().amax(dim=1, keepdim=True) / 448.0
    sw = torch.where(sw > 0, sw, torch.ones_like(sw))
    y = (e4m3_round(x) @ e4m3_round(w / sw).t()) * sw.t()
    return y if b is None else y + b


def block_linear(x: Tensor, w: Tensor, b: Optional[Tensor], cfg: OracleCfg) -> Tensor:
    return linear8(x, w, b) if cfg.block_fp8 else linear(x, w, b, cfg)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance (image_encoder.py:173,183)."""
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (common.py:26 via build_sam act default)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def gelu_logistic(x: Tensor) -> Tensor:
    """The fp8 emulation's GELU (cfg.block_fp8 only; NOT the reference's): x / (1 + 2^(-x (a + b x^2))), the form gemm8.h's e4m3
    epilogue evaluates (wm_common.h gelu_e4m3_fast2; within 2.7e-4 of gelu_erf everywhere, far below an e4m3 step)."""
    return x / (1.0 + torch.exp2(-x * (2.3087653 + 0.10012561 * x * x)))


# ----------------------------------------------------------------------------
# A2  MedSAM.fft  (network.py:36-57)
# ----------------------------------------------------------------------------
def grayscale(x: Tensor) -> Tensor:
    """torchvision Grayscale on a (B,3,H,W) tensor: 0.2989 R + 0.587 G + 0.114 B
    (network.py:41; torchvision 0.14.1/0.23.0 `rgb_to_grayscale`).  UNPINNED."""
    r, g, b = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    return 0.2989 * r + 0.587 * g + 0.114 * b


def highpass_band(n: int, rate: float = 0.125) -> Tuple[int, int]:
    """Signed-frequency interval [lo, hi] zeroed per axis by network.py:43-45.

    `line = int((w*h*rate)**.5 // 2)`; the mask zeroes shifted indices
    [n/2-line, n/2+line), i.e. signed frequencies [-line, line-1]."""
    line = int((n * n * rate) ** 0.5 // 2)
    return -line, line - 1


def hfc_fft(x: Tensor, rate: float = 0.125) -> Tensor:
    """|Re(ifft2(mask * fft2(gray)))| with the centred low-frequency square removed.

    network.py:47-55.  The reference's fftshift/ifftshift run over all four
    dims; the batch/channel rolls cancel, so only the spatial
    shift matters and is expressed here directly on signed frequencies."""
    g = grayscale(x)
    n = g.shape[-1]
    assert g.shape[-2] == n
    lo, hi = highpass_band(n, rate)
    f = torch.fft.fftfreq(n, d=1.0 / n).round().to(torch.int64)   # signed integer freqs
    inside = (f >= lo) & (f <= hi)
    keep = ~(inside[:, None] & inside[None, :])                    # (n,n) True = pass
    spec = torch.fft.fft2(g, norm="forward")
    spec = spec * keep.to(spec.dtype)
    out = torch.fft.ifft2(spec, norm="forward").real
    return out.abs()


# ----------------------------------------------------------------------------
# A4/A6  patch / HFC embed  (image_encoder.py:386-450)
# ----------------------------------------------------------------------------
def patchify(x: Tensor, patch: int) -> Tensor:
    """(B,C,H,W) -> (B, H/p, W/p, C*p*p) in Conv2d weight order (c, ky, kx)."""
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    t = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5)
    return t.reshape(B, gh, gw, C * patch * patch)


def conv_embed(x: Tensor, w: Tensor, b: Tensor, patch: int, cfg: OracleCfg) -> Tensor:
    """Conv2d(k=p, s=p) + NCHW->NHWC, as a GEMM over patches (image_encoder.py:409-417, 442-450)."""
    return linear(patchify(x, patch), w.reshape(w.shape[0], -1), b, cfg)


# ----------------------------------------------------------------------------
# A7  CrossAttentionHfcPatch  (image_encoder.py:452-516)
# ----------------------------------------------------------------------------
def mha_core(q: Tensor, k: Tensor, v: Tensor, heads: int, cfg: OracleCfg, chunk: int = 1024) -> Tensor:
    """softmax(q k^T / sqrt(hd)) v for (B,Nq,C)/(B,Nk,C) inputs, heads split on C."""
    B, Nq, C = q.shape
    hd = C // heads
    qh = q.reshape(B, Nq, heads, hd).permute(0, 2, 1, 3)
    kh = k.reshape(B, -1, heads, hd).permute(0, 2, 1, 3)
    vh = v.reshape(B, -1, heads, hd).permute(0, 2, 1, 3)
    scale = 1.0 / math.sqrt(hd)
    outs = []
    for s in range(0, Nq, chunk):
        a = (cfg.rnd(qh[:, :, s:s + chunk]) @ cfg.rnd(kh).transpose(-1, -2)) * scale
        p = a.softmax(-1)
        outs.append(cfg.rnd(p) @ cfg.rnd(vh))
    o = torch.cat(outs, dim=2)
    return o.permute(0, 2, 1, 3).reshape(B, Nq, C)


def hfc_adaptor(hfc_tok: Tensor, patch_tok: Tensor, W: Dict[str, Tensor], cfg: OracleCfg) -> Tensor:
    """hfc_tok (B,64,64,1024), patch_tok (B,64,64,D) -> (B,64,64,D)  (image_encoder.py:486-516).

    nn.MultiheadAttention in eval mode: packed in_proj (q,k,v thirds), scale
    1/sqrt(hd), out_proj; dropout off.  The sequence-first (4096,B,C) layout of
    the reference only matters for the scramble reshape at :512, restated here
    per image as "reinterpret the row-major [4096 tok, 1024 ch] buffer as
    [1024, 64, 64]"."""
    p = "image_encoder.hfc_attn."
    B, G, _, H = hfc_tok.shape
    N = G * G
    hfc = linear(hfc_tok.reshape(B, N, H), W[p + "proj_hfc.weight"].reshape(H, H), W[p + "proj_hfc.bias"], cfg)
    hfc = hfc + W[p + "pos_embed"].reshape(H, N).t()            # :494 (pos is NCHW)
    D = patch_tok.shape[-1]
    pt = linear(patch_tok.reshape(B, N, D), W[p + "proj_patch.weight"].reshape(H, D), W[p + "proj_patch.bias"], cfg)  # :495
    wi, bi = W[p + "cross_attn.in_proj_weight"], W[p + "cross_attn.in_proj_bias"]
    q = linear(pt, wi[:H], bi[:H], cfg)
    k = linear(hfc, wi[H:2 * H], bi[H:2 * H], cfg)
    v = linear(hfc, wi[2 * H:], bi[2 * H:], cfg)
    a = mha_core(q, k, v, cfg.hfc_heads, cfg)
    a = linear(a, W[p + "cross_attn.out_proj.weight"], W[p + "cross_attn.out_proj.bias"], cfg)   # :500-503
    y = layer_norm(pt + a, W[p + "norm1.weight"], W[p + "norm1.bias"], 1e-5)                      # :504-505
    z = linear(torch.relu(linear(y, W[p + "linear1.weight"], W[p + "linear1.bias"], cfg)),
               W[p + "linear2.weight"], W[p + "linear2.bias"], cfg)                               # :506
    y = layer_norm(z + y, W[p + "norm2.weight"], W[p + "norm2.bias"], 1e-5)                       # :508-509
    # scramble (:512): per image, the [N, H] token-major buffer is re-read as [H, G, G]
    scr = y.reshape(B, H, N)                       # channel' = 4*tok//... purely a reinterpretation
    back = linear(scr.transpose(1, 2), W[p + "proj_back.weight"].reshape(D, H), W[p + "proj_back.bias"], cfg)  # :513
    return back.reshape(B, G, G, D)                # :514


# ----------------------------------------------------------------------------
# A8-A13  encoder blocks  (image_encoder.py:141-383, common.py:13-26)
# ----------------------------------------------------------------------------
def rel_pos_table(size: int, table: Tensor) -> Tensor:
    """R[i, j] = table[i - j + size - 1]  (image_encoder.py:340-344, equal q/k sizes)."""
    assert table.shape[0] == 2 * size - 1, "interpolation branch (:328-335) is never taken on this path"
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return table[idx]                               # (size, size, hd)


def attention_rel(x: Tensor, W: Dict[str, Tensor], pre: str, heads: int, cfg: OracleCfg) -> Tensor:
    """x (B', S, S, D) -> same; qkv, decomposed rel-pos bias, softmax, proj (image_encoder.py:246-262)."""
    Bp, S, _, D = x.shape
    N = S * S
    hd = D // heads
    arnd = bf16_round if cfg.block_fp8 else cfg.rnd                            # fp8 mode: attention on the bf16 kernels
    qkv = block_linear(x.reshape(Bp, N, D), W[pre + "qkv.weight"], W[pre + "qkv.bias"], cfg)
    qkv = arnd(qkv).reshape(Bp, N, 3, heads, hd).permute(2, 0, 3, 1, 4)       # (3,B',h,N,hd)
    q, k, v = qkv[0], qkv[1], qkv[2]
    Rh = arnd(rel_pos_table(S, W[pre + "rel_pos_h"]))
    Rw = arnd(rel_pos_table(S, W[pre + "rel_pos_w"]))
    scale = hd ** -0.5
    out = torch.empty(Bp, heads, N, hd)
    # head-at-a-time keeps the 4096x4096 global case inside memory
    for h in range(heads):
        qh, kh, vh = q[:, h], k[:, h], v[:, h]                                 # (B',N,hd)
        a = (qh @ kh.transpose(-1, -2)) * scale                                # :253
        rq = qh.reshape(Bp, S, S, hd)                                          # unscaled q, :256
        rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh)                         # :376
        rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw)                         # :377
        a = (a.view(Bp, S, S, S, S) + rel_h[..., :, None] + rel_w[..., None, :]).view(Bp, N, N)  # :379-381
        p = a.softmax(-1)
        out[:, h] = arnd(p) @ vh
    o = out.permute(0, 2, 1, 3).reshape(Bp, N, D)
    # (fp8 mode: the attention kernels write e4m3 directly from their fp32 accumulators; block_linear rounds its input)
    o = block_linear(o, W[pre + "proj.weight"], W[pre + "proj.bias"], cfg)
    return o.reshape(Bp, S, S, D)


def to_windows(x: Tensor, ws: int) -> Tuple[Tensor, int]:
    """Zero-pad bottom/right to a multiple of ws and cut ws x ws windows (image_encoder.py:276-286)."""
    B, Hh, Ww, C = x.shape
    pad = (ws - Hh % ws) % ws
    xp = F.pad(x, (0, 0, 0, pad, 0, pad))
    n = (Hh + pad) // ws
    xw = xp.reshape(B, n, ws, n, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * n * n, ws, ws, C)
    return xw, n


def from_windows(xw: Tensor, ws: int, n: int, size: int) -> Tensor:
    """Inverse of to_windows + crop (image_encoder.py:303-311)."""
    B = xw.shape[0] // (n * n)
    x = xw.reshape(B, n, n, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, n * ws, n * ws, -1)
    return x[:, :size, :size, :].contiguous()


def encoder_block(x: Tensor, W: Dict[str, Tensor], i: int, cfg: OracleCfg) -> Tensor:
    """Block.forward (image_encoder.py:188-204)."""
    pre = f"image_encoder.blocks.{i}."
    is_global = i in cfg.global_attn_indexes
    y = layer_norm(x, W[pre + "norm1.weight"], W[pre + "norm1.bias"], 1e-6)
    if is_global:
        y = attention_rel(y, W, pre + "attn.", cfg.num_heads, cfg)
    else:
        yw, n = to_windows(y, cfg.window)          # padded tokens are zeros AFTER norm1 (:190-194)
        yw = attention_rel(yw, W, pre + "attn.", cfg.num_heads, cfg)
        y = from_windows(yw, cfg.window, n, x.shape[1])
    x = x + y
    z = layer_norm(x, W[pre + "norm2.weight"], W[pre + "norm2.bias"], 1e-6)
    z = block_linear(z, W[pre + "mlp.lin1.weight"], W[pre + "mlp.lin1.bias"], cfg)
    z = gelu_logistic(z) if cfg.block_fp8 else gelu_erf(z)
    z = block_linear(z, W[pre + "mlp.lin2.weight"], W[pre + "mlp.lin2.bias"], cfg)
    return x + z


# ----------------------------------------------------------------------------
# A14  neck  (image_encoder.py:105-121,136; common.py:31-43)
# ----------------------------------------------------------------------------
def neck(x: Tensor, W: Dict[str, Tensor], cfg: OracleCfg) -> Tensor:
    """(B,64,64,D) -> (B,256,64,64).  LayerNorm2d == LayerNorm over channels per pixel, eps 1e-6."""
    p = "image_encoder.neck."
    B, G, _, D = x.shape
    C = W[p + "0.weight"].shape[0]
    y = linear(x, W[p + "0.weight"].reshape(C, D), None, cfg)
    y = layer_norm(y, W[p + "1.weight"], W[p + "1.bias"], 1e-6)
    y = F.conv2d(cfg.rnd(y.permute(0, 3, 1, 2)), cfg.rnd(W[p + "2.weight"]), None, padding=1)
    y = layer_norm(y.permute(0, 2, 3, 1), W[p + "3.weight"], W[p + "3.bias"], 1e-6)
    return y.permute(0, 3, 1, 2).contiguous()


# ----------------------------------------------------------------------------
# A3  ImageEncoderViT.forward  (image_encoder.py:123-138)
# ----------------------------------------------------------------------------
def encoder_stem(x: Tensor, x_hfc: Tensor, W: Dict[str, Tensor], cfg: OracleCfg) -> Tensor:
    e = "image_encoder."
    t = conv_embed(x, W[e + "patch_embed.proj.weight"], W[e + "patch_embed.proj.bias"], cfg.patch, cfg)
    t = t + W[e + "pos_embed"]
    h = conv_embed(x_hfc, W[e + "hfc_embed.proj.weight"], W[e + "hfc_embed.proj.bias"], cfg.patch, cfg)
    return hfc_adaptor(h, t, W, cfg) + t            # :130-131


def encoder_forward(x: Tensor, x_hfc: Tensor, W: Dict[str, Tensor], cfg: OracleCfg,
                    taps: Optional[Dict[str, Tensor]] = None) -> Tensor:
    t = encoder_stem(x, x_hfc, W, cfg)
    if taps is not None:
        taps["stem"] = t
    for i in range(cfg.depth):
        t = encoder_block(t, W, i, cfg)
        if taps is not None:
            taps[f"block{i}"] = t
    return neck(t, W, cfg)


# ----------------------------------------------------------------------------
# A15  dense positional encoding  (pos_encoder.py:50-70)
# ----------------------------------------------------------------------------
def dense_pe(gauss: Tensor, grid: int) -> Tensor:
    """(1, 2*F, grid, grid): coords (i+0.5)/grid -> 2c-1 -> @G -> 2pi -> [sin, cos]."""
    c = (torch.arange(grid, dtype=torch.float32) + 0.5) / grid
    yy, xx = torch.meshgrid(c, c, indexing="ij")
    coords = torch.stack([xx, yy], dim=-1)          # x first, pos_encoder.py:69
    coords = 2 * coords - 1
    proj = (coords @ gauss) * (2 * np.pi)
    pe = torch.cat([proj.sin(), proj.cos()], dim=-1)
    return pe.permute(2, 0, 1).unsqueeze(0).contiguous()


# ----------------------------------------------------------------------------
# A16-A18  decoder  (box_decoder.py:71-149, transformer.py:62-240)
# ----------------------------------------------------------------------------
def dec_attention(q: Tensor, k: Tensor, v: Tensor, W: Dict[str, Tensor], pre: str, cfg: OracleCfg) -> Tensor:
    """transformer.py:217-240: q/k/v projections, heads, softmax(qk^T/sqrt(c)), out_proj."""
    qp = linear(q, W[pre + "q_proj.weight"], W[pre + "q_proj.bias"], cfg)
    kp = linear(k, W[pre + "k_proj.weight"], W[pre + "k_proj.bias"], cfg)
    vp = linear(v, W[pre + "v_proj.weight"], W[pre + "v_proj.bias"], cfg)
    o = mha_core(qp, kp, vp, cfg.dec_heads, cfg)
    return linear(o, W[pre + "out_proj.weight"], W[pre + "out_proj.bias"], cfg)


def two_way_transformer(src: Tensor, pos: Tensor, tokens: Tensor, W: Dict[str, Tensor], cfg: OracleCfg
                        ) -> Tuple[Tensor, Tensor]:
    """src (B,C,g,g), pos (1,C,g,g), tokens (B,T,C) -> (queries, keys)  (transformer.py:62-106)."""
    B, C, g, _ = src.shape
    keys = src.flatten(2).transpose(1, 2)
    kpe = pos.flatten(2).transpose(1, 2)
    queries, qpe = tokens, tokens
    t = "mask_decoder.transformer."
    ln = lambda x, n: layer_norm(x, W[n + ".weight"], W[n + ".bias"], 1e-5)
    for i in range(cfg.dec_depth):
        L = f"{t}layers.{i}."
        if i == 0:                                  # skip_first_layer_pe: no PE, no residual (:155-156)
            queries = dec_attention(queries, queries, queries, W, L + "self_attn.", cfg)
        else:
            qq = queries + qpe
            queries = queries + dec_attention(qq, qq, queries, W, L + "self_attn.", cfg)
        queries = ln(queries, L + "norm1")
        queries = queries + dec_attention(queries + qpe, keys + kpe, keys, W, L + "cross_attn_token_to_image.", cfg)
        queries = ln(queries, L + "norm2")
        m = linear(torch.relu(linear(queries, W[L + "mlp.lin1.weight"], W[L + "mlp.lin1.bias"], cfg)),
                   W[L + "mlp.lin2.weight"], W[L + "mlp.lin2.bias"], cfg)
        queries = ln(queries + m, L + "norm3")
        keys = keys + dec_attention(keys + kpe, queries + qpe, queries, W, L + "cross_attn_image_to_token.", cfg)
        keys = ln(keys, L + "norm4")
    queries = queries + dec_attention(queries + qpe, keys + kpe, keys, W, t + "final_attn_token_to_image.", cfg)
    queries = ln(queries, t + "norm_final_attn")
    return queries, keys


def mlp_head(x: Tensor, W: Dict[str, Tensor], pre: str, cfg: OracleCfg) -> Tensor:
    """3-layer MLP with ReLU between (box_decoder.py:154-176)."""
    for j in range(3):
        x = linear(x, W[f"{pre}layers.{j}.weight"], W[f"{pre}layers.{j}.bias"], cfg)
        if j < 2:
            x = torch.relu(x)
    return x


def decoder_forward(emb: Tensor, W: Dict[str, Tensor], cfg: OracleCfg) -> Dict[str, Tensor]:
    """(B,256,64,64) -> pred_logits (B,51,8), pred_boxes (B,51,4)  (box_decoder.py:96-104, 128-147)."""
    B = emb.shape[0]
    pe = dense_pe(W["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], cfg.grid)
    tokens = W["mask_decoder.mask_tokens.weight"].unsqueeze(0).expand(B, -1, -1)
    hs, _ = two_way_transformer(emb, pe, tokens, W, cfg)
    hs = hs[:, :cfg.num_queries]
    logits = mlp_head(hs, W, "mask_decoder.class_embed.", cfg)
    boxes = mlp_head(hs, W, "mask_decoder.bbox_embed.", cfg).sigmoid()
    return {"pred_logits": logits, "pred_boxes": boxes}


# ----------------------------------------------------------------------------
# A1  MedSAM.forward  (network.py:59-87)
# ----------------------------------------------------------------------------
def model_forward(x: Tensor, W: Dict[str, Tensor], cfg: OracleCfg,
                  taps: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
    with torch.no_grad():
        hfc = hfc_fft(x)
        emb = encoder_forward(x, hfc, W, cfg, taps)
        if taps is not None:
            taps["hfc"] = hfc
            taps["embedding"] = emb
        return decoder_forward(emb, W, cfg)


# ----------------------------------------------------------------------------
# A19  PostProcess  (build_sam.py:219-258, utils/box_ops.py:9-13)   PINNED: tests/golden/postprocess_ref.npz
# ----------------------------------------------------------------------------
def postprocess(logits: Tensor, boxes: Tensor, target_sizes: Tensor, thr: float = 0.05) -> List[Dict[str, Tensor]]:
    prob = logits.softmax(-1)
    scores, labels = prob[..., :-1].max(-1)         # background column (last) excluded, :233
    out = []
    for s, l, b, ts in zip(scores, labels, boxes, target_sizes):
        keep = s > thr
        if int(keep.sum()) == 0:
            out.append({"scores": torch.zeros(0), "labels": torch.zeros(0, dtype=torch.int64),
                        "boxes": torch.zeros(0, 4)})
            continue
        cx, cy, w, h = b[keep].unbind(-1)
        xyxy = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
        # build_sam.py:252-253: scale = [t[0], t[1], t[0], t[1]] (the names there are swapped)
        sc = torch.stack([ts[0], ts[1], ts[0], ts[1]]).to(xyxy.dtype)
        out.append({"scores": s[keep], "labels": l[keep], "boxes": xyxy * sc})
    return out


# ----------------------------------------------------------------------------
# A20  score cut + NMS  (visualize_prediction.py:150-157; torchvision.ops.nms, UNPINNED)
# ----------------------------------------------------------------------------
def nms(boxes: Tensor, scores: Tensor, iou_thr: float) -> Tensor:
    """Greedy class-agnostic NMS; returns kept indices by descending score.

    Published torchvision semantics: stable descending sort, IoU = inter /
    (area_a + area_b - inter) with inter extents clamped at 0, a later box is
    suppressed when IoU > thr (strict)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = torch.sort(scores, descending=True, stable=True).indices.tolist()
    b = boxes.to(torch.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = [False] * n
    kept: List[int] = []
    for ii, i in enumerate(order):
        if dead[i]:
            continue
        kept.append(i)
        for j in order[ii + 1:]:
            if dead[j]:
                continue
            w = max(0.0, float(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])))
            h = max(0.0, float(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])))
            inter = np.float32(w) * np.float32(h)
            iou = inter / (np.float32(area[i]) + np.float32(area[j]) - inter)
            if iou > iou_thr:
                dead[j] = True
    return torch.tensor(kept, dtype=torch.int64)


def detect(result: Dict[str, Tensor], score_thr: float = 0.5, iou_thr: float = 0.4) -> Dict[str, Tensor]:
    """visualize_prediction.py:150-157 on one PostProcess result."""
    keep = result["scores"] > score_thr
    s, b, l = result["scores"][keep], result["boxes"][keep], result["labels"][keep]
    idx = nms(b, s, iou_thr)
    return {"scores": s[idx], "boxes": b[idx], "labels": l[idx], "nms_index": idx}
