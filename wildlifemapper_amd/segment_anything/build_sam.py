"""Drop-in model factory (reference: segment_anything/build_sam.py:19-60, 212-334).

`build_sam(checkpoint=None, args=None)` and `sam_model_registry[...]` return the same
TUPLE `(sam, criterion, postprocessors)` as the reference (:334).  `sam` holds the HIP-backed
encoder / decoder / prompt encoder; `postprocessors['bbox']` is the HIP PostProcess.
The DETR criterion + Hungarian matcher (:62-210, matcher.py) are training-loss code and out of
scope for this inference path: `criterion` is a stub whose loss dict is empty.
"""
from __future__ import annotations

from functools import partial
from typing import Dict, List, Optional

import torch
from torch import nn

from ..engine import postprocess_nms, split_records
from .. import _native as N
from .modeling import ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer


def build_sam_vit_h(checkpoint=None, args=None):
    return _build_sam(encoder_embed_dim=1280, encoder_depth=32, encoder_num_heads=16,
                      encoder_global_attn_indexes=[7, 15, 23, 31], checkpoint=checkpoint, args=args)


build_sam = build_sam_vit_h


def build_sam_vit_l(checkpoint=None, args=None):
    return _build_sam(encoder_embed_dim=1024, encoder_depth=24, encoder_num_heads=16,
                      encoder_global_attn_indexes=[5, 11, 17, 23], checkpoint=checkpoint, args=args)


def build_sam_vit_b(checkpoint=None, args=None):
    return _build_sam(encoder_embed_dim=768, encoder_depth=12, encoder_num_heads=12,
                      encoder_global_attn_indexes=[2, 5, 8, 11], checkpoint=checkpoint, args=args)


sam_model_registry = {
    "default": build_sam_vit_h,
    "vit_h": build_sam_vit_h,
    "vit_l": build_sam_vit_l,
    "vit_b": build_sam_vit_b,
}


class InferenceCriterion(nn.Module):
    """Stand-in for SetCriterion (build_sam.py:62-210): inference computes no loss."""

    def __init__(self) -> None:
        super().__init__()
        self.weight_dict: Dict[str, float] = {}

    def forward(self, outputs, targets):
        return {}


class PostProcess(nn.Module):
    """Model output -> per-image {'scores','labels','boxes'} (build_sam.py:212-258), on the GPU.

    `forward` keeps the reference contract.  `forward_with_nms` additionally applies the
    score cut + class-agnostic NMS of visualize_prediction.py:150-157 in the same kernel and
    returns, per image, the kept indices in the order torchvision.ops.nms would.
    """

    def __init__(self, confidence_threshold: float = 0.05) -> None:
        super().__init__()
        self.confidence_threshold = confidence_threshold

    def _records(self, outputs, target_sizes, score_thr, iou_thr):
        out_logits, out_bbox = outputs["pred_logits"], outputs["pred_boxes"]
        assert len(out_logits) == len(target_sizes)
        assert target_sizes.shape[1] == 2
        rec = postprocess_nms(out_logits.contiguous().float(), out_bbox.contiguous().float(), target_sizes,
                              self.confidence_threshold, score_thr, iou_thr)
        return split_records(rec)

    @torch.no_grad()
    def forward(self, outputs, target_sizes) -> List[Dict[str, torch.Tensor]]:
        r = self._records(outputs, target_sizes, 0.5, 0.4)
        results = []
        for b in range(r["scores"].shape[0]):
            keep = (r["flags"][b] & N.FLAG_CONF) != 0
            results.append({"scores": r["scores"][b][keep], "labels": r["labels"][b][keep], "boxes": r["boxes"][b][keep]})
        return results

    @torch.no_grad()
    def forward_with_nms(self, outputs, target_sizes, score_threshold: float = 0.5, iou_threshold: float = 0.4):
        r = self._records(outputs, target_sizes, score_threshold, iou_threshold)
        results = []
        for b in range(r["scores"].shape[0]):
            flags, rank = r["flags"][b], r["nms_rank"][b]
            cand = (flags & N.FLAG_SCORE) != 0                       # results[0]['scores'] > threshold
            kept = (flags & N.FLAG_NMS) != 0
            # index of each slot inside the score-filtered list, then order kept slots by NMS rank
            pos_in_cand = torch.cumsum(cand.to(torch.int64), 0) - 1
            slots = torch.nonzero(kept).flatten()
            slots = slots[torch.argsort(rank[slots])]
            results.append({"scores": r["scores"][b][slots], "labels": r["labels"][b][slots], "boxes": r["boxes"][b][slots],
                            "nms_index": pos_in_cand[slots], "slots": slots})
        return results


def _build_sam(encoder_embed_dim, encoder_depth, encoder_num_heads, encoder_global_attn_indexes, checkpoint=None, args=None):
    prompt_embed_dim = 256
    image_size = 1024
    vit_patch_size = 16
    image_embedding_size = image_size // vit_patch_size
    sam = Sam(
        image_encoder=ImageEncoderViT(
            depth=encoder_depth, embed_dim=encoder_embed_dim, img_size=image_size, mlp_ratio=4,
            norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_heads=encoder_num_heads, patch_size=vit_patch_size,
            qkv_bias=True, use_rel_pos=True, global_attn_indexes=encoder_global_attn_indexes, window_size=14,
            out_chans=prompt_embed_dim),
        prompt_encoder=PromptEncoder(embed_dim=prompt_embed_dim,
                                     image_embedding_size=(image_embedding_size, image_embedding_size),
                                     input_image_size=(image_size, image_size), mask_in_chans=16),
        mask_decoder=MaskDecoder(num_multimask_outputs=50,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=prompt_embed_dim, mlp_dim=2048, num_heads=8),
                                 transformer_dim=prompt_embed_dim, iou_head_depth=3, iou_head_hidden_dim=256),
        pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375])
    sam.eval()
    precision = getattr(args, "wm_precision", None) if args is not None else None
    if precision:
        sam.image_encoder._hub.set_precision(precision)
    if checkpoint is not None:
        # build_sam.py:311-322: SAM checkpoint, mask_decoder.* keys without 'transformer' dropped, strict=False
        state_dict = torch.load(checkpoint, map_location="cpu", weights_only=True)
        if isinstance(state_dict, dict) and "model" in state_dict and isinstance(state_dict["model"], dict):
            state_dict = state_dict["model"]
        for k in [k for k in state_dict if "mask_decoder" in k and "transformer" not in k]:
            del state_dict[k]
        sam.load_state_dict(state_dict, strict=False)
    criterion = InferenceCriterion()
    postprocessors = {"bbox": PostProcess(confidence_threshold=0.05)}
    return sam, criterion, postprocessors
