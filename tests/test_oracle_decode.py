"""Known answers / properties anchoring oracle/decode_ref.py (python-mode decode + TF-1.14 NMS restatement)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import decode_ref as dr

ANCHORS = np.array([[[0.76120044, 0.57155991], [0.6923348, 0.88535553], [0.47163042, 0.34163313]],
                    [[0.33340788, 0.70065861], [0.18124964, 0.38986752], [0.08497349, 0.1527057]]])


def test_xywh_to_all_zero_logits():
    p = np.zeros((7, 10, 3, 2), np.float32)
    xy, wh = dr.xywh_to_all(p, p, (7, 10), ANCHORS[0])
    # sigmoid(0)=.5 -> cell centres; x uses the column index and /W, y the row index and /H (utils.py:250-252,545)
    assert np.allclose(xy[2, 3, 1], [(3 + .5) / 10, (2 + .5) / 7])
    assert np.allclose(wh[0, 0], ANCHORS[0].astype(np.float32))


def test_correct_box_people_jpg_known_answer():
    # people.jpg 374x499 -> new_shape (224,299), offset (0,0.0328125), scale (1,1.0702341)
    inp, img = np.array([224, 320], np.float32), np.array([374, 499], np.float32)
    new_shape = np.round(img * np.min(inp / img))
    assert new_shape.tolist() == [224.0, 299.0]
    xy = np.array([[0.5, 0.5]], np.float32)
    wh = np.array([[0.2, 0.4]], np.float32)
    b = dr.correct_box(xy, wh, [224, 320], [374, 499])[0]
    off_x, sc_x = (320 - 299) / 2 / 320, 320 / 299
    assert abs(off_x - 0.0328125) < 1e-9 and abs(sc_x - 1.0702341) < 1e-6
    cx, cy, w, h = (0.5 - off_x) * sc_x, 0.5, 0.2 * sc_x, 0.4
    want = [(cy - h / 2) * 374, (cx - w / 2) * 499, (cy + h / 2) * 374, (cx + w / 2) * 499]
    assert np.allclose(b, want, rtol=1e-5)


def test_correct_box_identity_for_dog_jpg():
    xy = np.array([[0.25, 0.75]], np.float32)
    wh = np.array([[0.5, 0.25]], np.float32)
    b = dr.correct_box(xy, wh, [224, 320], [224, 320])[0]
    assert np.allclose(b, [(0.75 - .125) * 224, 0.0, (0.75 + .125) * 224, 0.5 * 320])


def test_nms_strict_threshold_and_cap():
    # two identical boxes: IoU == 1 > thr -> second suppressed; IoU == thr exactly is NOT suppressed (strict >)
    boxes = np.array([[0, 0, 2, 2], [0, 0, 2, 2], [0, 1, 2, 3]], np.float32)   # box0 vs box2: inter 2, union 6 -> 1/3
    scores = np.array([.9, .8, .7], np.float32)
    assert dr.non_max_suppression(boxes, scores, 30, 0.5) == [0, 2]
    assert dr.non_max_suppression(boxes, scores, 30, float(np.float32(2) / np.float32(6))) == [0, 2]
    assert dr.non_max_suppression(boxes, scores, 30, 0.33) == [0]
    far = np.array([[i * 10, 0, i * 10 + 1, 1] for i in range(40)], np.float32)
    assert len(dr.non_max_suppression(far, np.linspace(1, .5, 40).astype(np.float32), 30, .5)) == 30
    # flipped corners are normalised, zero area -> IoU 0 (never suppresses / suppressed)
    z = np.array([[0, 0, 2, 2], [2, 2, 0, 0], [1, 1, 1, 3]], np.float32)
    assert dr.non_max_suppression(z, np.array([.9, .8, .7], np.float32), 30, .5) == [0, 2]


def test_tie_break_is_ascending_index():
    boxes = np.array([[0, 0, 1, 1], [5, 5, 6, 6], [0, 0, 1, 1]], np.float32)
    assert dr.non_max_suppression(boxes, np.array([.5, .5, .5], np.float32), 30, .5) == [0, 1]


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 40), st.integers(0, 2 ** 31 - 1), st.floats(0.05, 0.9))
def test_nms_properties(n, seed, thr):
    rng = np.random.default_rng(seed)
    yx = rng.uniform(0, 10, (n, 2))
    hw = rng.uniform(0.1, 6, (n, 2))
    boxes = np.concatenate([yx, yx + hw], 1).astype(np.float32)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    sel = dr.non_max_suppression(boxes, scores, 30, thr)
    assert len(sel) <= 30 and len(set(sel)) == len(sel)
    assert all(scores[a] >= scores[b] for a, b in zip(sel, sel[1:]))                  # descending
    for i, a in enumerate(sel):                                                       # mutually non-overlapping
        for b in sel[:i]:
            assert not dr.tf_iou(boxes, a, b) > np.float32(thr)
    if len(sel) < 30:                                                                 # maximality
        for c in set(range(n)) - set(sel):
            assert any(dr.tf_iou(boxes, c, s) > np.float32(thr) and
                       (scores[s], -s) >= (scores[c], -c) for s in sel)
    again = dr.non_max_suppression(boxes[sel], scores[sel], 30, thr)                  # idempotent
    assert again == list(range(len(sel)))


def test_decode_image_order_and_multiclass_emit():
    rng = np.random.default_rng(7)
    preds = [rng.normal(0, 1, (7, 10, 3, 25)).astype(np.float32), rng.normal(0, 1, (14, 20, 3, 25)).astype(np.float32)]
    for p in preds:
        p[..., 4] -= 4
    # one confident box that passes for two classes -> emitted twice (keras_inference.py:122-131)
    preds[1][3, 4, 1, 4] = 9
    preds[1][3, 4, 1, 5 + 2] = 9
    preds[1][3, 4, 1, 5 + 11] = 9
    dets, idx = dr.decode_image(preds, ANCHORS, [224, 320], [224, 320], 0.7, 0.5)
    assert dets.shape[1] == 6 and len(dets) >= 2
    assert list(dets[:, 5]) == sorted(dets[:, 5])                                     # class-major ascending
    g = 210 + (3 * 20 + 4) * 3 + 1                                                    # (layer, h, w, anchor) row order
    assert set(dets[idx == g][:, 5].astype(int)) >= {2, 11}
    for c in np.unique(dets[:, 5]):
        s = dets[dets[:, 5] == c][:, 4]
        assert all(a >= b for a, b in zip(s, s[1:]))


@pytest.mark.parametrize('seed,obj,iou', [(0, 0.7, 0.5), (1, 0.3, 0.3), (2, 0.05, 0.6)])
def test_c_nms_helper_is_bit_equal_to_the_python_restatement(seed, obj, iou):
    """oracle/decode_nms_ref.c (what bench.py's cpu_baseline times) against decode_ref.decode_batch on seeded logits: the same rows, in
    the same order, bit for bit, the same box indices - also with ties (quantised logits), cap 30 reached and empty classes."""
    rng = np.random.default_rng(seed)
    B, A, Cn = 3, 3, 20
    preds = []
    for (h, w) in ((7, 10), (14, 20)):
        p = rng.normal(0, 2.0, (B, h, w, A, 5 + Cn)).astype(np.float32)
        p[..., 4] += 1.0
        if seed == 1:
            p = np.round(p * 2) / 2                                  # many exactly equal scores: the tie order matters
        p[..., 5 + 7] = -30.0                                        # a class nobody passes
        preds.append(p.astype(np.float32))
    want = dr.decode_batch(preds, ANCHORS, (224, 320), (240, 320), obj, iou)
    got = dr.decode_batch_fast(preds, ANCHORS, (224, 320), (240, 320), obj, iou, threads=2)
    assert sum(len(w_[0]) for w_ in want) > 100
    for (wr, wi), (gr, gi) in zip(want, got):
        assert wr.shape == gr.shape and np.array_equal(wr.view(np.uint32), gr.view(np.uint32))
        assert np.array_equal(wi, gi)


# ---- the third-party kernel behind keras_inference.py:125 (tf.image.non_max_suppression, TensorFlow 1.14, un-vendored): its OWN published
# known-answer tests, restated from tensorflow/core/kernels/non_max_suppression_op_test.cc (NonMaxSuppressionOpTest; the same boxes and
# scores are tensorflow/python/ops/image_ops_test.py::NonMaxSuppressionTest.testSelectFromThreeClusters).  Boxes are (y1, x1, y2, x2).
_TF_BOXES = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], np.float32)
_TF_SCORES = np.array([.9, .75, .6, .95, .5, .3], np.float32)


@pytest.mark.parametrize('name,boxes,scores,max_out,iou,want', [
    ('TestSelectFromThreeClusters', _TF_BOXES, _TF_SCORES, 3, 0.5, [3, 0, 5]),
    ('TestSelectFromThreeClustersFlippedCoordinates',
     np.array([[1, 1, 0, 0], [0, 0.1, 1, 1.1], [0, .9, 1, -0.1], [0, 10, 1, 11], [1, 10.1, 0, 11.1], [1, 101, 0, 100]], np.float32), _TF_SCORES, 3, 0.5,
     [3, 0, 5]),
    ('TestSelectAtMostTwoBoxesFromThreeClusters', _TF_BOXES, _TF_SCORES, 2, 0.5, [3, 0]),
    ('TestSelectWithNegativeScores', _TF_BOXES, _TF_SCORES - np.float32(10.0), 6, 0.5, [3, 0, 5]),
    ('TestSelectAtMostThirtyBoxesFromThreeClusters', _TF_BOXES, _TF_SCORES, 30, 0.5, [3, 0, 5]),
    ('TestSelectSingleBox', np.array([[0, 0, 1, 1]], np.float32), np.array([.9], np.float32), 3, 0.5, [0]),
    ('TestSelectFromTenIdenticalBoxes', np.tile(np.array([[0, 0, 1, 1]], np.float32), (10, 1)), np.full(10, .9, np.float32), 3, 0.5, [0]),
    ('TestEmptyInput', np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 30, 0.5, []),
])
def test_nms_restatement_against_tensorflows_own_known_answer_tests(name, boxes, scores, max_out, iou, want):
    """Pins oracle/decode_ref.non_max_suppression (and, through tests/test_gpu_decode.py's exact row-for-row comparisons, the HIP kernel) to
    the vectors TensorFlow itself tests this kernel with: greedy in descending score order, IoU on min/max-normalised corners (flipped
    coordinates), strict `>`, the cap, negative scores, identical boxes collapsing to the first."""
    assert dr.non_max_suppression(boxes, scores, max_out, iou) == want, name
    # the answer follows the boxes, not their positions: the reversed input gives the mirrored indices (identical boxes: the first again)
    assert dr.non_max_suppression(boxes[::-1].copy(), scores[::-1].copy(), max_out, iou) == [len(scores) - 1 - k for k in want] or name == 'TestSelectFromTenIdenticalBoxes'
