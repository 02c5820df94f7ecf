"""Parameter holders shared by the drop-in modules.

The reference's `common.py` (MLPBlock, LayerNorm2d; segment_anything/modeling/
common.py:13-43) computes in PyTorch.  Here these classes only OWN parameters
under the reference's state-dict names; all arithmetic runs in libwm_hip.so.
Calling one of them directly raises, on purpose: there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _ParamsOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} holds parameters only; the computation runs inside the "
                           "parent module's HIP path (wildlifemapper_amd has no per-layer PyTorch fallback)")


class MLPBlock(_ParamsOnly):
    """lin1 / lin2 of common.py:13-26."""

    def __init__(self, embedding_dim: int, mlp_dim: int, act=nn.GELU) -> None:
        super().__init__()
        self.lin1 = nn.Linear(embedding_dim, mlp_dim)
        self.lin2 = nn.Linear(mlp_dim, embedding_dim)


class LayerNorm2d(_ParamsOnly):
    """weight / bias of common.py:31-43 (eps 1e-6)."""

    def __init__(self, num_channels: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps
