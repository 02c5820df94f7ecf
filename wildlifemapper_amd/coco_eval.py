"""Own COCO-style bbox mAP.

The reference computes mAP with pycocotools (`inference.py:92-171, 235-323`), which is not installed here, so this
evaluator is a from-scratch restatement of the published COCOeval bbox procedure (area range "all", no crowd
regions): per category and IoU threshold, detections sorted by score (stable), greedy matching of each detection to
the unmatched ground truth of highest IoU >= thr, precision made monotone from the right and sampled at 101 recall
points, AP = mean over categories that have ground truth and over IoU 0.50:0.05:0.95.  PARITY UNPINNED against
pycocotools (only hand-worked cases, tests/test_coco_eval.py).

`map_vs_reference` is the BASELINE metric's "mAP vs CPU ref": the CPU reference's detections are taken as ground
truth and the GPU path's detections as predictions; identical detections give 1.0.
"""
from __future__ import annotations

from typing import Dict, Mapping, Sequence

import numpy as np

IOU_THRS = np.linspace(0.5, 0.95, 10)
REC_THRS = np.linspace(0.0, 1.0, 101)


def _iou_matrix(d: np.ndarray, g: np.ndarray) -> np.ndarray:
    """IoU of xyxy boxes, shape (len(d), len(g))."""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    x0 = np.maximum(d[:, None, 0], g[None, :, 0]); y0 = np.maximum(d[:, None, 1], g[None, :, 1])
    x1 = np.minimum(d[:, None, 2], g[None, :, 2]); y1 = np.minimum(d[:, None, 3], g[None, :, 3])
    inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
    ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1]); ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    union = ad[:, None] + ag[None, :] - inter
    return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def bbox_map(dets: Mapping[int, Mapping[str, np.ndarray]], gts: Mapping[int, Mapping[str, np.ndarray]],
             max_dets: int = 100, iou_thrs: Sequence[float] = IOU_THRS) -> Dict[str, float]:
    """dets[image_id] = {'boxes' (n,4) xyxy, 'scores' (n,), 'labels' (n,)}; gts[image_id] = {'boxes', 'labels'}."""
    cats = sorted({int(l) for g in gts.values() for l in np.asarray(g["labels"]).reshape(-1)})
    iou_thrs = np.asarray(iou_thrs, dtype=np.float64)
    ap = np.full((len(iou_thrs), len(cats)), -1.0)
    for ci, cat in enumerate(cats):
        scores_all, match_all, n_gt = [], [], 0
        for img in sorted(set(gts) | set(dets)):
            g = gts.get(img, {"boxes": np.zeros((0, 4)), "labels": np.zeros(0)})
            d = dets.get(img, {"boxes": np.zeros((0, 4)), "scores": np.zeros(0), "labels": np.zeros(0)})
            gb = np.asarray(g["boxes"], dtype=np.float64).reshape(-1, 4)[np.asarray(g["labels"]).reshape(-1) == cat]
            sel = np.asarray(d["labels"]).reshape(-1) == cat
            db = np.asarray(d["boxes"], dtype=np.float64).reshape(-1, 4)[sel]
            ds = np.asarray(d["scores"], dtype=np.float64).reshape(-1)[sel]
            order = np.argsort(-ds, kind="mergesort")[:max_dets]
            db, ds = db[order], ds[order]
            n_gt += len(gb)
            ious = _iou_matrix(db, gb)
            matched = np.zeros((len(iou_thrs), len(db)), dtype=bool)
            for ti, thr in enumerate(iou_thrs):
                taken = np.zeros(len(gb), dtype=bool)
                for di in range(len(db)):
                    best, best_iou = -1, min(thr, 1 - 1e-10)
                    for gi in range(len(gb)):
                        if taken[gi] or ious[di, gi] < best_iou:
                            continue
                        best, best_iou = gi, ious[di, gi]
                    if best >= 0:
                        taken[best] = True
                        matched[ti, di] = True
            scores_all.append(ds)
            match_all.append(matched)
        if n_gt == 0:
            continue
        scores = np.concatenate(scores_all) if scores_all else np.zeros(0)
        match = np.concatenate(match_all, axis=1) if match_all else np.zeros((len(iou_thrs), 0), dtype=bool)
        order = np.argsort(-scores, kind="mergesort")
        match = match[:, order]
        for ti in range(len(iou_thrs)):
            tp = np.cumsum(match[ti]); fp = np.cumsum(~match[ti])
            rc = tp / n_gt
            pr = tp / np.maximum(tp + fp, np.spacing(1))
            for i in range(len(pr) - 1, 0, -1):            # monotone from the right
                if pr[i] > pr[i - 1]:
                    pr[i - 1] = pr[i]
            inds = np.searchsorted(rc, REC_THRS, side="left")
            q = np.zeros(len(REC_THRS))
            ok = inds < len(pr)
            q[ok] = pr[inds[ok]]
            ap[ti, ci] = q.mean()
    valid = ap[ap > -1]
    def at(thr):
        row = ap[np.isclose(iou_thrs, thr)]
        row = row[row > -1]
        return float(row.mean()) if row.size else float("nan")
    return {"mAP": float(valid.mean()) if valid.size else float("nan"), "mAP50": at(0.5), "mAP75": at(0.75),
            "categories": float(len(cats))}


def map_vs_reference(pred: Mapping[int, Mapping[str, np.ndarray]], ref: Mapping[int, Mapping[str, np.ndarray]]) -> Dict[str, float]:
    """mAP of `pred` detections with the reference path's detections as ground truth."""
    gts = {k: {"boxes": v["boxes"], "labels": v["labels"]} for k, v in ref.items()}
    return bbox_map(pred, gts)
