"""Drop-in `evaluate` harness (reference: wildlifemapper/inference.py:30-89).

Same signature, loop shape and return contract: for each (NestedTensor, targets) batch run the model, PostProcess the
outputs against `orig_size`, hand the per-image results to a CocoEvaluator built on `base_ds`, then
synchronize -> accumulate -> summarize and return `(stats, coco_evaluator)` with
`stats['coco_eval_bbox'] = coco_evaluator.coco_eval['bbox'].stats.tolist()` (inference.py:72-89).

Differences, all outside the accelerated path:
  * the loss (`criterion`) may be None or return an empty dict (training-loss code is out of scope); its entries, if any,
    are averaged into `stats` as the reference's MetricLogger does;
  * COCO mAP is computed by the build's own evaluator (coco_eval.py): pycocotools, which the reference imports
    (inference.py:15-18), is absent from this image, so that parity is unpinned;
  * `base_ds` may be a pycocotools-like object with `.dataset`, a COCO-format dict (coco_annotations/*.json) or a path;
    with `base_ds=None` no evaluator is built and `(stats, None)` is returned, as the reference's `if coco_evaluator is
    not None` branches allow;
  * detections are merged across ranks as fixed-size records by one padded all-gather (dist.gather_detections), not by
    pickling (utils/misc.py:180-220); `stats['images']` / `stats['detections']` count the MERGED set (each image once,
    although DistributedSampler repeats images to pad the last shard).
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict

import numpy as np
import torch

from .coco_eval import CocoEvaluator
from .segment_anything.utils import misc as utils


@torch.no_grad()
def evaluate(model, criterion, postprocessors, data_loader, base_ds, device, args):
    model.eval()
    if criterion is not None:
        criterion.eval()
    iou_types = tuple(k for k in ("bbox", "segm") if k in postprocessors.keys())
    coco_evaluator = CocoEvaluator(base_ds, iou_types) if base_ds is not None else None
    loss_sums: Dict[str, float] = defaultdict(float)
    n_batches = 0
    seen: Dict[int, int] = {}                     # image id -> detections (this rank), used when no evaluator merges
    for data in data_loader:
        image, targets = data[0], data[1]
        targets = [{k: (v.to(device) if hasattr(v, "to") else v) for k, v in t.items()} for t in targets]
        # whole image as the prompt, as inference.py:47-49 builds it (MedSAM.forward ignores it, network.py:69-78)
        b, c, h, w = image.tensors.shape
        boxes_np = np.repeat(np.array([[0, 0, h, w]]), getattr(args, "batch_size", b), axis=0)
        image = image.to(device)
        outputs = model(image, boxes_np)
        if criterion is not None:
            loss_dict = criterion(outputs, targets) or {}
            for k, v in loss_dict.items():
                loss_sums[k] += float(v)
        n_batches += 1
        orig_target_sizes = torch.stack([t["orig_size"] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, orig_target_sizes)
        res = {int(target["image_id"].item()): output for target, output in zip(targets, results)}
        for k, o in res.items():
            seen[k] = len(o["scores"])
        if coco_evaluator is not None:
            coco_evaluator.update(res)

    # gather the stats from all processes (inference.py:70-76)
    if coco_evaluator is not None:
        coco_evaluator.synchronize_between_processes()
        coco_evaluator.accumulate()
        coco_evaluator.summarize()
    stats = {k: v / max(n_batches, 1) for k, v in loss_sums.items()}
    # images / detections of the whole job, each image once: DistributedSampler pads the last shard with repeats, so a sum
    # over ranks would count those twice; the evaluator's merged set (or, without an evaluator, the same merge) does not
    if coco_evaluator is not None:
        ev = coco_evaluator.coco_eval["bbox"]
        stats["images"] = float(len(coco_evaluator.img_ids))
        stats["detections"] = float(sum(len(v) for v in ev.dts.values()))
    else:
        from . import dist as wdist
        merged = wdist.gather_detections({k: {"boxes": np.zeros((0, 4), np.float32), "scores": np.zeros((0,), np.float32),
                                              "labels": np.zeros((0,), np.int64)} for k in seen})
        stats["images"] = float(len(merged))
        if utils.get_world_size() > 1:
            t = torch.tensor([float(sum(seen.values()))], dtype=torch.float64)
            dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
            t = t.to(dev)
            torch.distributed.all_reduce(t)
            stats["detections"] = float(t.item())       # no evaluator: upper bound (sampler repeats counted per rank)
        else:
            stats["detections"] = float(sum(seen.values()))
    if coco_evaluator is not None and "bbox" in postprocessors.keys():
        stats["coco_eval_bbox"] = coco_evaluator.coco_eval["bbox"].stats.tolist()
    return stats, coco_evaluator
