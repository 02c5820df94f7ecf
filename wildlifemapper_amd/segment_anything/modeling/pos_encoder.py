"""Drop-in slim `PromptEncoder` (reference: segment_anything/modeling/pos_encoder.py:9-79).

Owns the gaussian matrix buffer under the reference's name.  In the HIP path the
dense positional encoding is a constant table built once when weights are packed
(wm_finalize_weights); `get_dense_pe()` is kept for signature parity and returns a
marker the drop-in `MaskDecoder` recognises.
"""
from __future__ import annotations

from typing import Optional, Tuple, Type

import torch
from torch import nn


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))


class DensePE:
    """Handle for 'the dense PE of this prompt encoder' (computed natively from the gaussian matrix)."""

    def __init__(self, owner: "PromptEncoder") -> None:
        self.owner = owner
        self.shape = (1, owner.embed_dim, *owner.image_embedding_size)


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int, activation: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        if embed_dim != 256 or tuple(image_embedding_size) != (64, 64):
            raise NotImplementedError("PromptEncoder (HIP) is built for embed_dim 256 on a 64x64 grid (build_sam.py:288-293)")
        self.embed_dim = embed_dim
        self.image_embedding_size = tuple(image_embedding_size)
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self._hub = None

    def get_dense_pe(self) -> DensePE:
        return DensePE(self)
