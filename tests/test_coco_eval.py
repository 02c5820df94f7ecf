"""Hand-worked cases for the own COCO-style bbox evaluator (parity with pycocotools is unpinned: not installed)."""
import numpy as np

from wildlifemapper_amd.coco_eval import bbox_map, map_vs_reference


def _d(boxes, scores, labels):
    return {"boxes": np.asarray(boxes, float).reshape(-1, 4), "scores": np.asarray(scores, float), "labels": np.asarray(labels)}


def test_identical_detections_score_one():
    ref = {1: _d([[0, 0, 10, 10], [20, 20, 40, 40]], [0.9, 0.8], [1, 2]), 2: _d([[5, 5, 9, 9]], [0.7], [1])}
    r = map_vs_reference(ref, ref)
    assert r["mAP"] == 1.0 and r["mAP50"] == 1.0 and r["categories"] == 2


def test_one_of_two_found():
    gts = {1: {"boxes": np.array([[0, 0, 10, 10], [20, 20, 30, 30]], float), "labels": np.array([1, 1])}}
    dets = {1: _d([[0, 0, 10, 10]], [0.9], [1])}
    r = bbox_map(dets, gts)
    # recall reaches 0.5 with precision 1: 51 of the 101 recall points (0.00..0.50) have precision 1
    assert abs(r["mAP"] - 51 / 101) < 1e-12


def test_false_positive_ranked_first_halves_precision():
    gts = {1: {"boxes": np.array([[0, 0, 10, 10]], float), "labels": np.array([3])}}
    dets = {1: _d([[50, 50, 60, 60], [0, 0, 10, 10]], [0.9, 0.8], [3, 3])}
    r = bbox_map(dets, gts)
    assert abs(r["mAP"] - 0.5) < 1e-12          # precision 1/2 at every recall point


def test_iou_threshold_sweep():
    gts = {1: {"boxes": np.array([[0, 0, 10, 10]], float), "labels": np.array([1])}}
    dets = {1: _d([[0, 0, 10, 8]], [0.9], [1])}     # IoU 0.8: counts at thresholds 0.50..0.80 (7 of 10)
    r = bbox_map(dets, gts)
    assert abs(r["mAP"] - 0.7) < 1e-12 and r["mAP50"] == 1.0 and r["mAP75"] == 1.0


def test_wrong_label_is_a_miss_and_empty_inputs():
    gts = {1: {"boxes": np.array([[0, 0, 10, 10]], float), "labels": np.array([1])}}
    assert bbox_map({1: _d([[0, 0, 10, 10]], [0.9], [2])}, gts)["mAP"] == 0.0
    assert bbox_map({}, gts)["mAP"] == 0.0
    assert np.isnan(bbox_map({}, {})["mAP"])
