"""`Sam` attribute container (reference: segment_anything/modeling/sam.py:19-47).
Only the container role is kept (train.py:194-196 reads .image_encoder / .mask_decoder /
.prompt_encoder); the reference's own Sam.forward is dead code on this fork."""
from __future__ import annotations

from typing import List

import torch
from torch import nn


class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder, prompt_encoder, mask_decoder,
                 pixel_mean: List[float] = [123.675, 116.28, 103.53], pixel_std: List[float] = [58.395, 57.12, 57.375]) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.Tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.Tensor(pixel_std).view(-1, 1, 1), False)

    @property
    def device(self):
        return self.pixel_mean.device

    def forward(self, *a, **k):
        raise RuntimeError("Sam.forward is not a live path in WildlifeMapper (it omits x_hfc); use network.MedSAM")
