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


# ---------------------------------------------------------------------------
# Full COCO bbox protocol (12 summary numbers) behind the reference's CocoEvaluator interface
# ---------------------------------------------------------------------------
AREA_RNG = ((0.0, 1e5 ** 2), (0.0, 32.0 ** 2), (32.0 ** 2, 96.0 ** 2), (96.0 ** 2, 1e5 ** 2))     # all, small, medium, large
MAX_DETS = (1, 10, 100)


class CocoBboxEval:
    """Restatement of the published COCOeval bbox procedure (iouType 'bbox', useCats, area ranges all / small / medium /
    large, maxDets 1 / 10 / 100, crowd regions as ignore regions), what `inference.py:92-171,283-323` drives through
    pycocotools.  `stats` is the 12-number list `evaluate` returns as stats['coco_eval_bbox'] (inference.py:84-87):
    AP, AP50, AP75, APs, APm, APl, AR1, AR10, AR100, ARs, ARm, ARl; -1 where a slice has no ground truth.
    PARITY UNPINNED against pycocotools (absent from the image): hand-worked cases only (tests/test_coco_eval.py)."""

    def __init__(self, dataset: Mapping) -> None:
        self.cat_ids = sorted(int(c["id"]) for c in dataset.get("categories", []))
        self.gts: Dict[tuple, list] = {}
        for a in dataset.get("annotations", []):
            x, y, w, h = [float(v) for v in a["bbox"]]
            rec = {"box": (x, y, x + w, y + h), "area": float(a.get("area", w * h)), "crowd": int(a.get("iscrowd", 0)),
                   "ignore": int(a.get("iscrowd", 0))}       # the published COCOeval._prepare overwrites gt['ignore'] with iscrowd
            self.gts.setdefault((int(a["image_id"]), int(a["category_id"])), []).append(rec)
        if not self.cat_ids:
            self.cat_ids = sorted({k[1] for k in self.gts})
        self.dts: Dict[tuple, list] = {}
        self.img_ids: list = []
        self.eval: Dict[str, np.ndarray] = {}
        self.stats = np.full(12, -1.0)

    # detections of one image: boxes xyxy
    def add(self, image_id: int, boxes: np.ndarray, scores: np.ndarray, labels: np.ndarray) -> None:
        image_id = int(image_id)
        if image_id not in self.img_ids:
            self.img_ids.append(image_id)
        for b, s, l in zip(np.asarray(boxes, dtype=np.float64).reshape(-1, 4), np.asarray(scores, dtype=np.float64).reshape(-1),
                           np.asarray(labels).reshape(-1)):
            self.dts.setdefault((image_id, int(l)), []).append({"box": tuple(b), "score": float(s), "area": float((b[2] - b[0]) * (b[3] - b[1]))})

    @staticmethod
    def _iou(d: np.ndarray, g: np.ndarray, crowd: np.ndarray) -> np.ndarray:
        if len(d) == 0 or len(g) == 0:
            return np.zeros((len(d), len(g)))
        x0 = np.maximum(d[:, None, 0], g[None, :, 0]); y0 = np.maximum(d[:, None, 1], g[None, :, 1])
        x1 = np.minimum(d[:, None, 2], g[None, :, 2]); y1 = np.minimum(d[:, None, 3], g[None, :, 3])
        inter = np.clip(x1 - x0, 0, None) * np.clip(y1 - y0, 0, None)
        ad = (d[:, 2] - d[:, 0]) * (d[:, 3] - d[:, 1]); ag = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
        union = np.where(crowd[None, :] > 0, ad[:, None], ad[:, None] + ag[None, :] - inter)     # crowd: intersection over detection area
        return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)

    def _evaluate_img(self, img: int, cat: int, rng, max_det: int):
        gt = self.gts.get((img, cat), [])
        dt = self.dts.get((img, cat), [])
        if not gt and not dt:
            return None
        g_ig = np.array([g["ignore"] or g["area"] < rng[0] or g["area"] > rng[1] for g in gt], dtype=bool)
        gorder = np.argsort(g_ig, kind="mergesort")                     # non-ignored first
        gt = [gt[i] for i in gorder]
        g_ig = g_ig[gorder]
        dorder = np.argsort([-d["score"] for d in dt], kind="mergesort")[:max_det]
        dt = [dt[i] for i in dorder]
        crowd = np.array([g["crowd"] for g in gt], dtype=int)
        ious = self._iou(np.array([d["box"] for d in dt], dtype=np.float64).reshape(-1, 4),
                         np.array([g["box"] for g in gt], dtype=np.float64).reshape(-1, 4), crowd)
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm = -np.ones((T, G), dtype=int)
        dtm = -np.ones((T, D), dtype=int)
        dt_ig = np.zeros((T, D), dtype=bool)
        for ti, t in enumerate(IOU_THRS):
            for di in range(D):
                iou = min(t, 1 - 1e-10)
                m = -1
                for gi in range(G):
                    if gtm[ti, gi] >= 0 and not crowd[gi]:
                        continue
                    if m > -1 and not g_ig[m] and g_ig[gi]:
                        break                                           # only ignore regions left and a real match is held
                    if ious[di, gi] < iou:
                        continue
                    iou = ious[di, gi]
                    m = gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ig[m]
                dtm[ti, di] = m
                gtm[ti, m] = di
        d_out = np.array([d["area"] < rng[0] or d["area"] > rng[1] for d in dt], dtype=bool).reshape(1, D)
        dt_ig = dt_ig | ((dtm < 0) & np.repeat(d_out, T, 0))
        return {"scores": np.array([d["score"] for d in dt]), "matched": dtm >= 0, "dt_ig": dt_ig, "n_gt": int((~g_ig).sum())}

    def accumulate(self) -> None:
        T, R, K, A, M = len(IOU_THRS), len(REC_THRS), len(self.cat_ids), len(AREA_RNG), len(MAX_DETS)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        imgs = sorted(set(self.img_ids))
        for ki, cat in enumerate(self.cat_ids):
            for ai, rng in enumerate(AREA_RNG):
                per_img = [self._evaluate_img(i, cat, rng, MAX_DETS[-1]) for i in imgs]
                per_img = [e for e in per_img if e is not None]
                if not per_img:
                    continue
                for mi, md in enumerate(MAX_DETS):
                    scores = np.concatenate([e["scores"][:md] for e in per_img])
                    order = np.argsort(-scores, kind="mergesort")
                    matched = np.concatenate([e["matched"][:, :md] for e in per_img], axis=1)[:, order]
                    ig = np.concatenate([e["dt_ig"][:, :md] for e in per_img], axis=1)[:, order]
                    npig = sum(e["n_gt"] for e in per_img)
                    if npig == 0:
                        continue
                    tps = np.cumsum(matched & ~ig, axis=1).astype(np.float64)
                    fps = np.cumsum(~matched & ~ig, axis=1).astype(np.float64)
                    for ti in range(T):
                        tp, fp = tps[ti], fps[ti]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[ti, ki, ai, mi] = rc[-1] if nd else 0.0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds = np.searchsorted(rc, REC_THRS, side="left")
                        q = np.zeros(R)
                        for ri, pi in enumerate(inds):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[ti, :, ki, ai, mi] = q
        self.eval = {"precision": precision, "recall": recall}

    def summarize(self, verbose: bool = False) -> np.ndarray:
        p, r = self.eval["precision"], self.eval["recall"]

        def ap(thr=None, area=0, md=2):
            s = p[:, :, :, area, md] if thr is None else p[np.isclose(IOU_THRS, thr)][:, :, :, area, md]
            s = s[s > -1]
            return float(s.mean()) if s.size else -1.0

        def ar(area=0, md=2):
            s = r[:, :, area, md]
            s = s[s > -1]
            return float(s.mean()) if s.size else -1.0

        self.stats = np.array([ap(), ap(0.5), ap(0.75), ap(area=1), ap(area=2), ap(area=3), ar(md=0), ar(md=1), ar(md=2),
                               ar(area=1), ar(area=2), ar(area=3)])
        if verbose:
            names = ("AP", "AP50", "AP75", "APs", "APm", "APl", "AR1", "AR10", "AR100", "ARs", "ARm", "ARl")
            print("IoU metric: bbox  " + "  ".join(f"{n}={v:.3f}" for n, v in zip(names, self.stats)))
        return self.stats


def _as_coco_dataset(base_ds) -> Mapping:
    """`base_ds` as the reference hands it over (a pycocotools COCO object, inference.py:20-27: its `.dataset` dict), a
    COCO-format dict (coco_annotations/*.json layout) or a path to such a file."""
    import json
    if isinstance(base_ds, (str, bytes)):
        with open(base_ds) as f:
            return json.load(f)
    if hasattr(base_ds, "dataset") and isinstance(base_ds.dataset, Mapping):
        return base_ds.dataset
    if isinstance(base_ds, Mapping):
        return base_ds
    raise TypeError(f"base_ds: expected a COCO-format dict, a path or an object with .dataset, got {type(base_ds).__name__}")


class CocoEvaluator:
    """Same surface as the reference's CocoEvaluator (inference.py:92-171): update / synchronize_between_processes /
    accumulate / summarize, `.coco_eval['bbox'].stats`.  Rank merge: fixed-size detection records through one padded
    all-gather behind a 2-word status all-reduce (wildlifemapper_amd.dist.gather_detections), not the reference's pickle gather (utils/misc.py:180-220)."""

    def __init__(self, coco_gt, iou_types=("bbox",)) -> None:
        assert isinstance(iou_types, (list, tuple))
        if tuple(iou_types) != ("bbox",):
            raise NotImplementedError("only iou_types=('bbox',): the detection head has no masks (build_sam.py:333)")
        self.iou_types = tuple(iou_types)
        self.dataset = _as_coco_dataset(coco_gt)
        self.coco_eval = {"bbox": CocoBboxEval(self.dataset)}
        self.img_ids: list = []
        self._local: Dict[int, Dict[str, np.ndarray]] = {}

    def update(self, predictions: Mapping) -> None:
        for img_id, pred in predictions.items():
            self.img_ids.append(int(img_id))
            self._local[int(img_id)] = {k: np.asarray(v.detach().cpu() if hasattr(v, "detach") else v) for k, v in pred.items()}

    def synchronize_between_processes(self) -> None:
        from . import dist as wdist
        merged = wdist.gather_detections(self._local)
        ev = self.coco_eval["bbox"]
        for img_id in sorted(merged):                   # unique, sorted image ids (merge(), inference.py:240-259)
            d = merged[img_id]
            ev.add(img_id, d["boxes"], d["scores"], d["labels"])
        self.img_ids = sorted(merged)

    def accumulate(self) -> None:
        self.coco_eval["bbox"].accumulate()

    def summarize(self) -> None:
        self.coco_eval["bbox"].summarize(verbose=True)
