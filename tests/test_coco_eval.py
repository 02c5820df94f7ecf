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


# ---------------------------------------------------------------------------
# CocoBboxEval / CocoEvaluator: the 12-number COCO summary behind the reference's evaluator interface
# ---------------------------------------------------------------------------
import json
import os

import pytest

from wildlifemapper_amd.coco_eval import CocoBboxEval, CocoEvaluator


def _ds(anns, cats=(1, 2, 3)):
    return {"images": [], "categories": [{"id": c} for c in cats],
            "annotations": [{"id": i + 1, "image_id": a[0], "category_id": a[1], "bbox": list(a[2]), "area": a[2][2] * a[2][3],
                             "iscrowd": a[3] if len(a) > 3 else 0} for i, a in enumerate(anns)]}


def _run(ds, dets):
    ev = CocoBboxEval(ds)
    for img, d in dets.items():
        ev.add(img, d["boxes"], d["scores"], d["labels"])
    ev.accumulate()
    return ev.summarize()


def test_coco12_hand_worked_precision_recall():
    """2 ground truths, detections TP(.9) FP(.8) TP(.7): precision 1 up to recall 0.5 (51 recall points), 2/3 beyond (50)."""
    ds = _ds([(1, 1, (0, 0, 40, 40)), (1, 1, (100, 100, 40, 40))])
    dets = {1: _d([[0, 0, 40, 40], [300, 300, 340, 340], [100, 100, 140, 140]], [0.9, 0.8, 0.7], [1, 1, 1])}
    s = _run(ds, dets)
    want = (51 * 1.0 + 50 * (2 / 3)) / 101
    assert abs(s[0] - want) < 1e-12 and abs(s[1] - want) < 1e-12 and abs(s[2] - want) < 1e-12
    # areas 1600 = medium (32^2 .. 96^2): small and large slices have no ground truth
    assert s[3] == -1 and s[5] == -1
    # the false positive (also medium-sized) stays in the medium slice
    assert abs(s[4] - want) < 1e-12
    # AR@1: the single best detection finds 1 of 2; AR@10 = AR@100 = 1
    assert abs(s[6] - 0.5) < 1e-12 and s[7] == 1.0 and s[8] == 1.0 and s[10] == 1.0


def test_coco12_area_ranges_and_ignored_detections():
    """A small (20x20) and a large (100x100) ground truth, both found: every slice that has ground truth scores 1; in the
    'small' slice the large pair is ignored (neither a miss nor a false positive)."""
    ds = _ds([(7, 2, (0, 0, 20, 20)), (7, 2, (200, 200, 100, 100))])
    dets = {7: _d([[0, 0, 20, 20], [200, 200, 300, 300]], [0.6, 0.9], [2, 2])}
    s = _run(ds, dets)
    one = pytest.approx(1.0, abs=1e-12)                      # precision = tp / (tp + fp + eps), as the published procedure
    assert s[0] == one and s[3] == one and s[5] == one and s[4] == -1
    assert s[9] == one and s[11] == one and s[10] == -1
    assert abs(s[6] - 0.5) < 1e-12                          # maxDets 1: only the 0.9 detection counts -> half the recall


def test_coco12_crowd_region_absorbs_detections():
    """Detections inside an iscrowd region are ignored (not false positives); the crowd itself is not a target."""
    ds = _ds([(1, 1, (0, 0, 50, 50)), (1, 1, (100, 100, 200, 200), 1)])
    dets = {1: _d([[0, 0, 50, 50], [150, 150, 180, 180], [160, 160, 190, 190]], [0.9, 0.8, 0.7], [1, 1, 1])}
    s = _run(ds, dets)
    assert s[0] == pytest.approx(1.0, abs=1e-12) and s[8] == 1.0


def test_coco12_agrees_with_bbox_map_on_reference_annotation_data(golden_dir):
    """On the reference's own annotation data (tests/golden/coco_val_subset.json = the first 8 images of
    coco_annotations/val.json) with jittered, partly dropped, partly spurious detections: the 12-number evaluator's AP /
    AP50 / AP75 over 'all' areas equal the independent single-range implementation (bbox_map)."""
    ds = json.load(open(os.path.join(golden_dir, "coco_val_subset.json")))
    assert len(ds["images"]) == 8 and len(ds["categories"]) == 6 and all(a["iscrowd"] == 0 for a in ds["annotations"])
    rng = np.random.default_rng(3)
    dets, gts = {}, {}
    for im in ds["images"]:
        anns = [a for a in ds["annotations"] if a["image_id"] == im["id"]]
        b = np.array([[a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2], a["bbox"][1] + a["bbox"][3]] for a in anns], float).reshape(-1, 4)
        l = np.array([a["category_id"] for a in anns], int)
        gts[im["id"]] = {"boxes": b, "labels": l}
        keep = rng.random(len(b)) < 0.8
        jb = b[keep] + rng.normal(0, 2.0, (int(keep.sum()), 4))
        fp = rng.random((3, 2)) * 3000
        fpb = np.concatenate([fp, fp + 40], axis=1)
        dets[im["id"]] = _d(np.concatenate([jb, fpb]), np.concatenate([rng.random(len(jb)) * 0.5 + 0.5, rng.random(3) * 0.6]),
                            np.concatenate([l[keep], rng.integers(1, 7, 3)]))
    s = _run(ds, dets)
    r = bbox_map(dets, gts)
    assert abs(s[0] - r["mAP"]) < 1e-12 and abs(s[1] - r["mAP50"]) < 1e-12 and abs(s[2] - r["mAP75"]) < 1e-12
    assert 0.05 < s[0] < 0.95 and s[8] >= s[7] >= s[6] > 0
    # the evaluator-class surface of the reference (update / synchronize / accumulate / summarize / coco_eval['bbox'].stats)
    ce = CocoEvaluator(ds, ("bbox",))
    ce.update(dets)
    ce.synchronize_between_processes()
    ce.accumulate()
    ce.summarize()
    assert np.allclose(ce.coco_eval["bbox"].stats, s)
    with pytest.raises(NotImplementedError):
        CocoEvaluator(ds, ("bbox", "segm"))


def test_coco12_crowd_area_split_and_ignore_field_hand_worked():
    """One image, one category, worked on paper.  Ground truth: A small 20x20; B medium 50x50; C large
    200x200 with iscrowd = 1; D medium 40x40 carrying ignore = 1 but iscrowd = 0 -- the published COCOeval overwrites
    `ignore` with `iscrowd`, so D IS a target (and is missed).  Detections: .9 = A exactly, .8 inside the crowd (absorbed),
    .7 = B exactly, .6 a small false positive.
      all:    targets A, B, D; TP, (ignored), TP, FP -> recall 1/3, 2/3 at precision 1 -> 67 of 101 recall points
      small:  target A; the B match and the crowd match are ignored, the small FP counts but ranks after the TP -> AP 1
      medium: targets B, D; the A match, the crowd match and the out-of-range FP are ignored -> recall 1/2 -> 51 / 101
      large:  only the crowd -> no target -> -1."""
    anns = [(1, (0, 0, 20, 20), 0, 0), (2, (100, 100, 50, 50), 0, 0), (3, (300, 300, 200, 200), 1, 0), (4, (600, 600, 40, 40), 0, 1)]
    ds = {"images": [], "categories": [{"id": 1}],
          "annotations": [{"id": i, "image_id": 1, "category_id": 1, "bbox": list(b), "area": b[2] * b[3], "iscrowd": c, "ignore": ig}
                          for i, b, c, ig in anns]}
    dets = {1: _d([[0, 0, 20, 20], [320, 320, 360, 360], [100, 100, 150, 150], [800, 800, 830, 830]], [0.9, 0.8, 0.7, 0.6], [1, 1, 1, 1])}
    s = _run(ds, dets)
    assert s[0] == pytest.approx(67 / 101, abs=1e-12) and s[1] == pytest.approx(67 / 101, abs=1e-12) and s[2] == pytest.approx(67 / 101, abs=1e-12)
    assert s[3] == pytest.approx(1.0, abs=1e-12)
    assert s[4] == pytest.approx(51 / 101, abs=1e-12)
    assert s[5] == -1
    assert s[6] == pytest.approx(1 / 3, abs=1e-12) and s[7] == pytest.approx(2 / 3, abs=1e-12) and s[8] == pytest.approx(2 / 3, abs=1e-12)
    assert s[9] == pytest.approx(1.0, abs=1e-12) and s[10] == pytest.approx(0.5, abs=1e-12) and s[11] == -1
