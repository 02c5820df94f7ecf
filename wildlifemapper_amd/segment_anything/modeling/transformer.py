"""Parameter holders for the two-way transformer (reference: segment_anything/modeling/transformer.py).
The arithmetic runs in libwm_hip.so as part of wm_decoder_forward."""
from __future__ import annotations

from typing import Type

import torch.nn as nn

from .common import MLPBlock, _ParamsOnly


class Attention(_ParamsOnly):
    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1) -> None:
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, "num_heads must divide embedding_dim."
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)


class TwoWayAttentionBlock(_ParamsOnly):
    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, activation: Type[nn.Module] = nn.ReLU,
                 attention_downsample_rate: int = 2, skip_first_layer_pe: bool = False) -> None:
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim, activation)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe


class TwoWayTransformer(_ParamsOnly):
    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int,
                 activation: Type[nn.Module] = nn.ReLU, attention_downsample_rate: int = 2) -> None:
        super().__init__()
        if (depth, embedding_dim, num_heads, mlp_dim, attention_downsample_rate) != (2, 256, 8, 2048, 2) or activation is not nn.ReLU:
            raise NotImplementedError("TwoWayTransformer (HIP) is built for build_sam.py:297-302: depth 2, dim 256, 8 heads, mlp 2048, ReLU")
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, activation, attention_downsample_rate,
                                 skip_first_layer_pe=(i == 0)) for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)
