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
    pickling (utils/misc.py:180-220).
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
    n_images = 0
    n_dets = 0
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
        n_images += len(res)
        n_dets += sum(len(o["scores"]) for o in res.values())
        if coco_evaluator is not None:
            coco_evaluator.update(res)

    # gather the stats from all processes (inference.py:70-76)
    if coco_evaluator is not None:
        coco_evaluator.synchronize_between_processes()
        coco_evaluator.accumulate()
        coco_evaluator.summarize()
    stats = {k: v / max(n_batches, 1) for k, v in loss_sums.items()}
    world = utils.get_world_size()
    counts = torch.tensor([n_images, n_dets], dtype=torch.float64)
    if world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        counts = counts.to(dev)
        torch.distributed.all_reduce(counts)
        counts = counts.cpu()
    stats["images"] = float(counts[0])
    stats["detections"] = float(counts[1])
    if coco_evaluator is not None and "bbox" in postprocessors.keys():
        stats["coco_eval_bbox"] = coco_evaluator.coco_eval["bbox"].stats.tolist()
    return stats, coco_evaluator
