"""Large-frame front end: 6000 x 4000 aerial frames (coco_annotations/*.json image sizes) -> overlapping
1024 x 1024 tiles -> the accelerated path -> cross-tile merge of the detections.

The reference has no such step (it down-scales whole frames to 768 px, dataloader_coco.py:288), so there is no behaviour to
match: the checker is oracle/tiling_oracle.py, a numpy restatement of exactly what is done here.
  * tile_origins: the fewest tiles per axis whose neighbours overlap by at least `overlap`, evenly spread, the first and
    last flush with the frame edges (no padding unless the frame is smaller than a tile): 7 x 5 tiles for 6000 x 4000;
  * frame_to_tiles: wm_tile_frame_u8 (cut + ToTensor + Normalize on the GPU);
  * detect_frame: model.detect per batch of tiles, then wm_merge_tiles_nms -- detections that survived their own tile's
    score cut + NMS move to frame coordinates and compete in one more class-agnostic NMS (IoU 0.4).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from . import _native as N


def _axis_origins(size: int, tile: int, overlap: int) -> List[int]:
    """Fewest tiles whose neighbours overlap by at least `overlap`, spread evenly, first and last flush with the edges."""
    if size <= tile:
        return [0]
    stride = tile - overlap
    if stride <= 0:
        raise ValueError("overlap must be smaller than the tile")
    n = -(-(size - tile) // stride) + 1
    return [(i * (size - tile) + (n - 1) // 2) // (n - 1) for i in range(n)]


def tile_origins(height: int, width: int, tile: int = 1024, overlap: int = 128) -> List[Tuple[int, int]]:
    """(y0, x0) of every tile, row-major."""
    return [(y, x) for y in _axis_origins(height, tile, overlap) for x in _axis_origins(width, tile, overlap)]


def frame_to_tiles(frame: torch.Tensor, origins: torch.Tensor) -> torch.Tensor:
    """frame (H,W,3) uint8 on a ROCm device, origins (n,2) int32 (y0, x0) on the same device -> (n,3,1024,1024) fp32."""
    if not frame.is_cuda or frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[-1] != 3:
        raise RuntimeError(f"frame_to_tiles: expected an (H,W,3) uint8 ROCm tensor, got {tuple(frame.shape)} {frame.dtype} on {frame.device}")
    frame = frame.contiguous()
    origins = origins.to(device=frame.device, dtype=torch.int32).contiguous()
    n = origins.shape[0]
    out = torch.empty((n, 3, 1024, 1024), device=frame.device, dtype=torch.float32)
    with torch.cuda.device(frame.device):
        N.check(N.lib().wm_tile_frame_u8(N.ptr(frame), N.ptr(origins), N.ptr(out), n, frame.shape[0], frame.shape[1], N.stream_ptr(frame.device)))
    return out


def merge_tile_records(records: torch.Tensor, origins: torch.Tensor, iou_thr: float = 0.4) -> torch.Tensor:
    """records (n,51,8) raw per-tile records (boxes in tile pixels) -> (n,51,8) merged records in frame coordinates with
    FLAG_MERGED / nms_rank of the cross-tile NMS (wm_merge_tiles_nms)."""
    N.require_cuda(records, "records")
    origins = origins.to(device=records.device, dtype=torch.int32).contiguous()
    out = torch.empty_like(records)
    with torch.cuda.device(records.device):
        N.check(N.lib().wm_merge_tiles_nms(N.ptr(records), N.ptr(origins), records.shape[0], float(iou_thr), N.ptr(out), N.stream_ptr(records.device)))
    return out


@torch.no_grad()
def detect_frame(model, frame: torch.Tensor, overlap: int = 128, batch: int = 16, iou_thr: float = 0.4) -> Dict[str, torch.Tensor]:
    """One frame -> merged detections {'boxes' (k,4) frame xyxy, 'scores', 'labels', 'tile'} in merged-NMS order."""
    from .engine import split_records
    H, W = int(frame.shape[0]), int(frame.shape[1])
    org = torch.tensor(tile_origins(H, W, 1024, overlap), dtype=torch.int32, device=frame.device)
    recs = []
    for i in range(0, org.shape[0], batch):
        x = frame_to_tiles(frame, org[i:i + batch])
        recs.append(model.detect(x)["records"])                      # target size 1024 x 1024: boxes in tile pixels
    rec = torch.cat(recs, dim=0)
    merged = merge_tile_records(rec, org, iou_thr)
    r = split_records(merged)
    flat = {k: v.reshape(-1, *v.shape[2:]) for k, v in r.items()}
    kept = torch.nonzero((flat["flags"] & N.FLAG_MERGED) != 0).flatten()
    kept = kept[torch.argsort(flat["nms_rank"][kept])]
    return {"boxes": flat["boxes"][kept], "scores": flat["scores"][kept], "labels": flat["labels"][kept],
            "tile": kept // N.NUM_QUERIES, "origins": org, "records": merged}
