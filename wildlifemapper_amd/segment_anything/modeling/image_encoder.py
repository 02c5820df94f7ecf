"""Drop-in `ImageEncoderViT` (reference: segment_anything/modeling/image_encoder.py:17-138).

Same constructor arguments, same attribute / state-dict names and the same
`forward(x, x_hfc) -> (B, 256, 64, 64)` signature.  The forward pass is one call
into libwm_hip.so (wm_encoder_forward): patch / HFC embeds, the HFC cross-attention
adaptor with its scramble reshape, the window / global attention blocks with
decomposed rel-pos bias and the neck all run as HIP kernels.
"""
from __future__ import annotations

from typing import Optional, Tuple, Type

import torch
import torch.nn as nn

from ...engine import EngineHub
from .common import LayerNorm2d, MLPBlock, _ParamsOnly


class PatchEmbed(_ParamsOnly):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans: int = 3, embed_dim: int = 768):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=kernel_size, stride=stride, padding=padding)


class HfcEmbed(PatchEmbed):
    def __init__(self, kernel_size=(16, 16), stride=(16, 16), padding=(0, 0), in_chans: int = 1, embed_dim: int = 1024):
        super().__init__(kernel_size, stride, padding, in_chans, embed_dim)


class CrossAttentionHfcPatch(_ParamsOnly):
    """Parameters of image_encoder.py:452-484."""

    def __init__(self, d_model=1024, hfc_dim=1024, nhead=8, dropout=0.1, dim_feedforward=1024, activation="relu", proj_dim=1024):
        super().__init__()
        self.proj_hfc = nn.Conv2d(hfc_dim, proj_dim, (1, 1))
        self.proj_patch = nn.Conv2d(d_model, proj_dim, (1, 1))
        self.cross_attn = nn.MultiheadAttention(proj_dim, nhead, dropout=dropout)
        self.linear1 = nn.Linear(proj_dim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, dim_feedforward)
        self.norm1 = nn.LayerNorm(proj_dim)
        self.norm2 = nn.LayerNorm(dim_feedforward)
        self.embed_dim = d_model
        self.proj_back = nn.Conv2d(dim_feedforward, d_model, (1, 1))
        self.pos_embed = nn.Parameter(torch.zeros(1, proj_dim, 64, 64))


class Attention(_ParamsOnly):
    """Parameters of image_encoder.py:207-244."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = True, use_rel_pos: bool = False,
                 rel_pos_zero_init: bool = True, input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.use_rel_pos = use_rel_pos
        if use_rel_pos:
            assert input_size is not None, "Input size must be provided if using relative positional encoding."
            self.rel_pos_h = nn.Parameter(torch.zeros(2 * input_size[0] - 1, head_dim))
            self.rel_pos_w = nn.Parameter(torch.zeros(2 * input_size[1] - 1, head_dim))


class Block(_ParamsOnly):
    """Parameters of image_encoder.py:141-186."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True,
                 norm_layer: Type[nn.Module] = nn.LayerNorm, act_layer: Type[nn.Module] = nn.GELU,
                 use_rel_pos: bool = False, rel_pos_zero_init: bool = True, window_size: int = 0,
                 input_size: Optional[Tuple[int, int]] = None) -> None:
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, use_rel_pos=use_rel_pos,
                              rel_pos_zero_init=rel_pos_zero_init,
                              input_size=input_size if window_size == 0 else (window_size, window_size))
        self.norm2 = norm_layer(dim)
        self.mlp = MLPBlock(embedding_dim=dim, mlp_dim=int(dim * mlp_ratio), act=act_layer)
        self.window_size = window_size


class ImageEncoderViT(nn.Module):
    def __init__(self, img_size: int = 1024, patch_size: int = 16, in_chans: int = 3, embed_dim: int = 768,
                 depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0, out_chans: int = 256,
                 qkv_bias: bool = True, norm_layer: Type[nn.Module] = nn.LayerNorm,
                 act_layer: Type[nn.Module] = nn.GELU, use_abs_pos: bool = True, use_rel_pos: bool = False,
                 rel_pos_zero_init: bool = True, window_size: int = 0,
                 global_attn_indexes: Tuple[int, ...] = ()) -> None:
        super().__init__()
        # what the HIP path is built for is exactly what build_sam.py:274-287 constructs
        unsupported = []
        if img_size != 1024 or patch_size != 16 or in_chans != 3: unsupported.append("img_size/patch_size/in_chans != 1024/16/3")
        if out_chans != 256: unsupported.append("out_chans != 256")
        if float(mlp_ratio) != 4.0: unsupported.append("mlp_ratio != 4")
        if not (qkv_bias and use_abs_pos and use_rel_pos): unsupported.append("qkv_bias/use_abs_pos/use_rel_pos must be True")
        if window_size != 14: unsupported.append("window_size != 14")
        if act_layer is not nn.GELU: unsupported.append("act_layer != GELU")
        if unsupported:
            raise NotImplementedError("ImageEncoderViT (HIP): " + "; ".join(unsupported))
        eps = getattr(norm_layer(8), "eps", None)
        if eps is None or abs(eps - 1e-6) > 1e-12:
            raise NotImplementedError("ImageEncoderViT (HIP): block LayerNorm eps must be 1e-6 (build_sam.py:280)")
        self.img_size = img_size
        self.patch_embed = PatchEmbed((patch_size, patch_size), (patch_size, patch_size), in_chans=in_chans, embed_dim=embed_dim)
        self.hfc_embed = HfcEmbed((patch_size, patch_size), (patch_size, patch_size), in_chans=1, embed_dim=1024)
        self.pos_embed = nn.Parameter(torch.zeros(1, img_size // patch_size, img_size // patch_size, embed_dim))
        self.hfc_attn = CrossAttentionHfcPatch(d_model=embed_dim, hfc_dim=1024, nhead=8, dropout=0.1,
                                               dim_feedforward=1024, activation="relu", proj_dim=1024)
        self.blocks = nn.ModuleList()
        for i in range(depth):
            self.blocks.append(Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                                     norm_layer=norm_layer, act_layer=act_layer, use_rel_pos=use_rel_pos,
                                     rel_pos_zero_init=rel_pos_zero_init,
                                     window_size=window_size if i not in global_attn_indexes else 0,
                                     input_size=(img_size // patch_size, img_size // patch_size)))
        self.neck = nn.Sequential(
            nn.Conv2d(embed_dim, out_chans, kernel_size=1, bias=False), LayerNorm2d(out_chans),
            nn.Conv2d(out_chans, out_chans, kernel_size=3, padding=1, bias=False), LayerNorm2d(out_chans))
        self._hub = EngineHub(embed_dim, depth, num_heads, tuple(global_attn_indexes))
        self._hub.register("image_encoder.", self)

    def forward(self, x: torch.Tensor, x_hfc: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x_hfc is None:
            raise TypeError("ImageEncoderViT.forward needs x_hfc (image_encoder.py:128 embeds it unconditionally)")
        return self._hub.encoder_forward(x.contiguous().float(), x_hfc.contiguous().float())
