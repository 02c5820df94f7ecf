"""Drop-in `MedSAM` wrapper (reference: segment_anything/network.py:7-87):
FFT high-pass -> image encoder -> detection decoder, as ONE native call (wm_forward)."""
from __future__ import annotations

import torch
import torch.nn as nn


class MedSAM(nn.Module):
    def __init__(self, image_encoder, mask_decoder, prompt_encoder) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.mask_decoder = mask_decoder
        self.prompt_encoder = prompt_encoder
        # requires_grad pattern of network.py:19-34 (kept so optimizers built on it see the same set)
        for name, param in self.image_encoder.named_parameters():
            param.requires_grad = any(k in name for k in ("hfc_embed", "hfc_attn", "patch_embed"))
        for param in self.prompt_encoder.parameters():
            param.requires_grad = True
        for param in self.mask_decoder.parameters():
            param.requires_grad = True
        # one native handle for the three modules
        hub = image_encoder._hub
        hub.register("mask_decoder.", mask_decoder)
        hub.register("prompt_encoder.", prompt_encoder)
        old = getattr(mask_decoder, "_hub", None)
        if old is not None and old is not hub:
            old.close()
        mask_decoder._hub = hub
        object.__setattr__(mask_decoder, "_pe_owner", prompt_encoder)   # plain attribute, not a sub-module
        prompt_encoder._hub = hub
        self._hub = hub

    def fft(self, img, rate: float = 0.125) -> torch.Tensor:
        """High-frequency component (network.py:36-57); `rate` is fixed to 0.125 in the HIP kernel."""
        if rate != 0.125:
            raise NotImplementedError("MedSAM.fft (HIP) is built for rate=0.125 (network.py:36)")
        x = img.tensors if hasattr(img, "tensors") else img
        return self._hub.hfc_fft(x.contiguous().float())

    def forward(self, image, box=None):
        """image: NestedTensor (or anything with .tensors (B,3,1024,1024)); box is ignored as in network.py:69-78."""
        x = image.tensors if hasattr(image, "tensors") else image
        return self._hub.forward(x.contiguous().float())

    @torch.no_grad()
    def detect(self, image, target_sizes=None):
        """forward + PostProcess + score cut + NMS in the same native call; returns raw records (B,51,8)."""
        x = image.tensors if hasattr(image, "tensors") else image
        return self._hub.forward(x.contiguous().float(), target_sizes, want_records=True)
