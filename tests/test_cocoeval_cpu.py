"""COCO bbox AP/AR (multibox_amd/cocoeval.py, the metric step of eval.py:212-226) on hand-computed cases.
pycocotools is not installed here ("parity unpinned"); the expected numbers follow from its published algorithm."""
import numpy as np

from multibox_amd.cocoeval import evaluate_bbox, _iou_xywh

GT = [{"image_id": 1, "bbox": [10, 10, 50, 50], "area": 2500}, {"image_id": 1, "bbox": [100, 100, 40, 40], "area": 1600},
      {"image_id": 2, "bbox": [0, 0, 100, 100], "area": 10000}]


def test_iou():
    d = np.array([[0., 0., 10., 10.], [5., 5., 10., 10.]])
    g = np.array([[0., 0., 10., 10.], [20., 20., 5., 5.]])
    iou = _iou_xywh(d, g)
    assert np.allclose(iou, [[1.0, 0.0], [25.0 / 175.0, 0.0]])


def test_perfect_detections():
    dt = [[1, 10, 10, 50, 50, 0.9, 1], [1, 100, 100, 40, 40, 0.8, 1], [2, 0, 0, 100, 100, 0.7, 1]]
    stats, lines = evaluate_bbox(GT, dt)
    assert np.allclose(stats[:3], 1.0) and stats[3] == -1.0 and np.allclose(stats[4:6], 1.0)            # no small gt
    assert np.isclose(stats[6], 2.0 / 3.0) and np.allclose(stats[7:9], 1.0)                             # AR@1: 2 of 3 gt
    assert lines[0] == " Average Precision  (AP) @[ IoU=0.50:0.95 | area=   all | maxDets=100 ] = 1.000"
    assert lines[6] == " Average Recall     (AR) @[ IoU=0.50:0.95 | area=   all | maxDets=  1 ] = 0.667"


def test_one_false_positive_between_true_positives():
    # ranked TP, FP, TP over 3 gt: precision envelope [1, 2/3, 2/3] at recalls [1/3, 1/3, 2/3]
    # -> 34 thresholds (0..0.33) at 1.0, 33 thresholds (0.34..0.66) at 2/3, the rest 0
    dt = [[1, 10, 10, 50, 50, 0.9, 1], [1, 200, 200, 40, 40, 0.8, 1], [2, 0, 0, 100, 100, 0.7, 1]]
    stats, _ = evaluate_bbox(GT, dt)
    assert np.isclose(stats[0], (34 + 33 * 2.0 / 3.0) / 101.0) and np.isclose(stats[8], 2.0 / 3.0)


def test_iou_thresholds_and_area_ranges():
    # one gt 100x100 (large); detection shifted by 10 px in x: IoU = 90*100 / (2*10000 - 9000) = 0.818 -> matches
    # at thresholds 0.50..0.80 (7 of 10)
    gt = [{"image_id": 7, "bbox": [0, 0, 100, 100], "area": 10000}]
    dt = [[7, 10, 0, 100, 100, 0.5, 1]]
    stats, _ = evaluate_bbox(gt, dt)
    assert np.isclose(stats[0], 0.7) and np.allclose(stats[1:3], 1.0) and np.isclose(stats[8], 0.7)
    assert stats[3] == -1.0 and stats[4] == -1.0 and np.isclose(stats[5], 0.7)
    # a duplicate detection of the same gt is a false positive (one match per gt), ranked below: AP unchanged
    stats2, _ = evaluate_bbox(gt, dt + [[7, 10, 0, 100, 100, 0.4, 1]])
    assert np.isclose(stats2[0], 0.7)
    # ... ranked above the good one it takes the match at the thresholds it passes
    stats3, _ = evaluate_bbox(gt, [[7, 0, 0, 100, 100, 0.4, 1], [7, 30, 0, 100, 100, 0.9, 1]])   # IoU 0.538 first
    assert np.isclose(stats3[1], 1.0) and stats3[0] < 0.999


def test_no_detections_and_no_gt():
    stats, _ = evaluate_bbox(GT, [])
    assert stats[0] == -1.0 or stats[0] == 0.0
    stats, _ = evaluate_bbox([], [[1, 0, 0, 5, 5, 0.5, 1]])
    assert stats[0] == -1.0
