"""Input contract + collation helpers (reference: segment_anything/utils/misc.py).

NestedTensor / nested_tensor_from_tensor_list / custom_collate keep the reference's
behaviour (:15-84): a batch is a zero canvas (B,3,1024,1024) with each image copied
to the top-left, cropped at 1024, plus a bool mask (True = padding).

`all_gather` of the reference (:180-220) pickles arbitrary objects through a CUDA
uint8 tensor with a size pre-exchange.  On this path the only thing gathered is
detections, so the replacement is a single fixed-size all-gather of box records
(wildlifemapper_amd.dist.all_gather_records); `all_gather` here keeps the name and
list-of-per-rank-results contract for tensors.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

CANVAS = 1024


class NestedTensor(object):
    def __init__(self, tensors: Tensor, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        mask = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), mask)

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    b = len(tensor_list)
    first = tensor_list[0]
    canvas = torch.zeros((b, 3, CANVAS, CANVAS), dtype=first.dtype, device=first.device)
    mask = torch.ones((b, CANVAS, CANVAS), dtype=torch.bool, device=first.device)
    for i, img in enumerate(tensor_list):
        h, w = min(img.shape[1], CANVAS), min(img.shape[2], CANVAS)
        canvas[i, : img.shape[0], :h, :w].copy_(img[:, :h, :w])
        mask[i, :h, :w] = False
    return NestedTensor(canvas, mask)


def collate_fn(batch):
    batch = list(zip(*batch))
    batch[0] = nested_tensor_from_tensor_list(batch[0])
    return tuple(batch)


def custom_collate(batch):
    images = [d["image"] for d in batch]
    targets = [d["target"] for d in batch]
    return nested_tensor_from_tensor_list(images), targets


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def all_gather(data: Tensor) -> List[Tensor]:
    """Gather one same-shaped tensor per rank (list indexed by rank)."""
    world = get_world_size()
    if world == 1:
        return [data]
    out = [torch.empty_like(data) for _ in range(world)]
    dist.all_gather(out, data.contiguous())
    return out
