"""numpy restatement (test infrastructure only) of the large-frame front end, wildlifemapper_amd/tiling.py (SURVEY.md §8f
N3).  The reference has no such step, so this oracle restates the build's own definition: it pins the GPU kernels'
arithmetic (tile cut + normalise, cross-tile NMS), not a reference behaviour ("parity unpinned" in that sense)."""
from __future__ import annotations

import numpy as np

from oracle import wm_oracle as O
from wildlifemapper_amd import synth


def cut_tiles(frame: np.ndarray, origins) -> np.ndarray:
    """frame (H,W,3) uint8 -> (n,3,1024,1024) fp32: normalised content, zeros past the frame."""
    H, W = frame.shape[:2]
    out = np.zeros((len(origins), 3, 1024, 1024), dtype=np.float32)
    for i, (y0, x0) in enumerate(origins):
        y1, x1 = min(y0 + 1024, H), min(x0 + 1024, W)
        if y1 > y0 and x1 > x0:
            out[i, :, : y1 - y0, : x1 - x0] = synth.normalize_tile(frame[y0:y1, x0:x1])
    return out


def merge(boxes: np.ndarray, scores: np.ndarray, cand: np.ndarray, origins, iou_thr: float = 0.4):
    """boxes (n,51,4) tile pixels, scores (n,51), cand (n,51) bool (survived the tile's own NMS) -> (frame boxes (n*51,4) fp32,
    kept flat slot indices in merged-NMS order)."""
    import torch
    n = boxes.shape[0]
    shift = np.array([[x0, y0, x0, y0] for (y0, x0) in origins], dtype=np.float32)[:, None, :]
    fb = (boxes.astype(np.float32) + shift).reshape(n * 51, 4)
    idx = np.nonzero(cand.reshape(-1))[0]
    keep = O.nms(torch.from_numpy(fb[idx]), torch.from_numpy(scores.reshape(-1)[idx].astype(np.float32)), iou_thr).numpy()
    return fb, idx[keep]
