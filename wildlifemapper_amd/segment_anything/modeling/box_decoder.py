"""Drop-in detection decoder, class name `MaskDecoder` as in the reference
(segment_anything/modeling/box_decoder.py:16-107): 51 learned queries -> two-way
transformer -> class / box MLP heads.  forward() is one call into libwm_hip.so
(wm_decoder_forward); this module only owns the parameters.
"""
from __future__ import annotations

from typing import Dict, Optional, Type

import torch
from torch import nn

from ...engine import EngineHub
from .common import _ParamsOnly
from .pos_encoder import DensePE


class MLP(_ParamsOnly):
    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, sigmoid_output: bool = False) -> None:
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output


class MaskDecoder(nn.Module):
    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3,
                 activation: Type[nn.Module] = nn.GELU, iou_head_depth: int = 3, iou_head_hidden_dim: int = 256,
                 aux_loss=False, embed_dim=256) -> None:
        super().__init__()
        if transformer_dim != 256 or num_multimask_outputs != 50 or iou_head_depth != 3 or iou_head_hidden_dim != 256 or aux_loss:
            raise NotImplementedError("MaskDecoder (HIP) is built for build_sam.py:295-306: dim 256, 50+1 queries, 3-layer heads")
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.aux_loss = aux_loss
        self.num_classes = 6 + 1
        self.iou_token = nn.Embedding(1, transformer_dim)          # parameter kept; unused by forward (box_decoder.py:52)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.class_embed = MLP(transformer_dim, iou_head_hidden_dim, self.num_classes + 1, 3)
        self.bbox_embed = MLP(transformer_dim, iou_head_hidden_dim, 4, 3)
        self._hub: Optional[EngineHub] = None
        self._pe_owner = None

    def _ensure_hub(self, image_pe) -> EngineHub:
        owner = image_pe.owner if isinstance(image_pe, DensePE) else None
        if owner is None:
            raise TypeError("image_pe must come from PromptEncoder.get_dense_pe() of the drop-in package "
                            "(the HIP path builds the encoding from its gaussian matrix)")
        if self._hub is None:
            # standalone use: a decoder-only hub (encoder dims are irrelevant for the decoder group)
            self._hub = EngineHub(768, 12, 12, (2, 5, 8, 11))
            self._hub.register("mask_decoder.", self)
        if self._pe_owner is not owner:
            self._hub.register("prompt_encoder.", owner)
            owner._hub = self._hub
            object.__setattr__(self, "_pe_owner", owner)   # plain attribute, not a sub-module
        return self._hub

    def forward(self, image_embeddings: torch.Tensor, image_pe, sparse_prompt_embeddings=None,
                dense_prompt_embeddings=None, multimask_output: bool = False, hfc_embed=None) -> Dict[str, torch.Tensor]:
        # sparse/dense prompt embeddings, multimask_output and hfc_embed are ignored, as in box_decoder.py:119-149
        hub = self._ensure_hub(image_pe)
        return hub.decoder_forward(image_embeddings.contiguous().float())
