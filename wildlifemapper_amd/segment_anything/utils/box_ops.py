"""Box helpers on the inference path (reference: segment_anything/utils/box_ops.py:9-13).
Only cxcywh -> xyxy is on the path (used by PostProcess); the IoU/GIoU helpers there serve the
training loss and are out of scope."""
import torch


def box_cxcywh_to_xyxy(x: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)
