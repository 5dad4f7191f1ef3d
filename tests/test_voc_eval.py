"""voc_eval: the VOC detection metric on hand-computed cases (no GPU, no oracle)."""
import numpy as np
import pytest

from k210_yolo_framework_amd import voc_eval as ve


def _row(t, l, b, r, s, c):
    return [t, l, b, r, s, c]


def test_iou_known_values():
    assert ve.box_iou([0, 0, 10, 10], [[0, 0, 10, 10]])[0] == 1.0
    assert np.isclose(ve.box_iou([0, 0, 10, 10], [[0, 5, 10, 15]])[0], 50 / 150)
    assert ve.box_iou([0, 0, 10, 10], [[20, 20, 30, 30]])[0] == 0.0
    assert np.isclose(ve.box_iou([0, 0, 9, 9], [[0, 5, 9, 14]], plus_one=True)[0], 50 / 150)     # inclusive pixel convention
    assert ve.box_iou([0, 0, 0, 0], [[0, 0, 0, 0]])[0] == 0.0                                     # empty boxes: 0, not nan


def test_average_precision_hand_computed_curve():
    """TP, FP, TP against two ground-truth boxes: precision 1, 1/2, 2/3 at recall 1/2, 1/2, 1."""
    rec, prec = np.array([.5, .5, 1.]), np.array([1., .5, 2 / 3])
    assert np.isclose(ve.average_precision(rec, prec), 0.5 * 1.0 + 0.5 * (2 / 3))
    assert np.isclose(ve.average_precision(rec, prec, use_07_metric=True), (6 * 1.0 + 5 * (2 / 3)) / 11)
    assert ve.average_precision(np.array([]), np.array([])) == 0.0


def test_evaluate_matches_the_hand_computed_case_and_takes_each_box_once():
    gt = [np.array([_row(0, 0, 10, 10, 0, 1), _row(20, 20, 40, 40, 0, 1)])]
    det = [np.array([_row(0, 0, 10, 10, .9, 1),            # TP
                     _row(0, 1, 10, 11, .8, 1),            # second hit on the same box: FP
                     _row(21, 21, 40, 40, .7, 1),          # TP
                     _row(0, 0, 10, 10, .99, 0)])]         # class 0 has no ground truth: does not enter the mean
    r = ve.evaluate(det, gt, class_num=3)
    assert r['n_gt'].tolist() == [0, 2, 0] and r['n_det'].tolist() == [1, 3, 0]
    assert r['tp'].tolist() == [0, 2, 0] and r['fp'].tolist() == [1, 1, 0]
    assert np.isnan(r['ap'][0]) and np.isnan(r['ap'][2])
    assert np.isclose(r['ap'][1], 0.5 + 0.5 * 2 / 3) and np.isclose(r['map'], r['ap'][1])
    assert np.isclose(ve.evaluate(det, gt, 3, use_07_metric=True)['map'], (6 + 5 * 2 / 3) / 11)


def test_perfect_and_empty_detections_and_score_order_across_images():
    gt = [np.array([_row(0, 0, 10, 10, 0, 2)]), np.array([_row(5, 5, 30, 30, 0, 2)]), np.zeros((0, 6))]
    same = [g.copy() for g in gt]
    for d in same:
        if len(d):
            d[:, 4] = 0.5
    assert ve.evaluate(same, gt, 4)['map'] == 1.0
    assert ve.evaluate([np.zeros((0, 6))] * 3, gt, 4)['map'] == 0.0
    # a confident false positive in an image without objects comes FIRST in the ranking: precision 0, 1/2, 2/3 -> AP = 1/2*1/2 + 1/2*2/3
    det = [np.array([_row(0, 0, 10, 10, .6, 2)]), np.array([_row(5, 5, 30, 30, .5, 2)]), np.array([_row(0, 0, 9, 9, .9, 2)])]
    assert np.isclose(ve.evaluate(det, gt, 4)['map'], 0.5 * (2 / 3) + 0.5 * (2 / 3))            # envelope: max precision to the right


def test_difficult_boxes_are_neither_positives_nor_negatives():
    gt = [np.array([_row(0, 0, 10, 10, 0, 0), _row(20, 20, 30, 30, 0, 0)])]
    det = [np.array([_row(20, 20, 30, 30, .9, 0), _row(0, 0, 10, 10, .8, 0)])]
    r = ve.evaluate(det, gt, 1, difficult=[np.array([False, True])])
    assert r['n_gt'][0] == 1 and r['tp'][0] == 1 and r['fp'][0] == 0 and r['map'] == 1.0


def test_row_helpers_and_delta():
    rows = np.arange(5 * 6, dtype=np.float32).reshape(5, 6)
    parts = ve.split_rows(rows, np.array([0, 2, 2, 5]))
    assert [len(p) for p in parts] == [2, 0, 3] and np.array_equal(parts[2], rows[2:])
    dets = np.zeros((2, 4, 6), np.float32)
    assert [len(p) for p in ve.padded_rows(dets, np.array([3, 0]))] == [3, 0]
    gt = [np.array([_row(0, 0, 10, 10, 0, 0)])]
    a, b, d = ve.map_delta([np.zeros((0, 6))], gt, gt, 1)
    assert a == 0.0 and b == 1.0 and d == -100.0
    with pytest.raises(ValueError):
        ve.evaluate([np.zeros((0, 6))], [], 1)
