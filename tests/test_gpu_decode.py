"""GPU parity: batched Python-mode decode + per-class NMS (yk_decode_py) vs oracle/decode_ref.py."""
import numpy as np
import pytest

from oracle import decode_ref as dr

pytestmark = pytest.mark.gpu

ANCHORS = np.array([[[0.76120044, 0.57155991], [0.6923348, 0.88535553], [0.47163042, 0.34163313]],
                    [[0.33340788, 0.70065861], [0.18124964, 0.38986752], [0.08497349, 0.1527057]]])


def _preds(rng, B, hw, C, kind):
    out = []
    for (h, w) in hw:
        if kind == 'uniform':
            p = rng.uniform(-6, 6, (B, h, w, 3, 5 + C)).astype(np.float32)
        else:
            p = rng.normal(0, 1, (B, h, w, 3, 5 + C)).astype(np.float32)
            p[..., 4] = rng.normal(-4, 2, (B, h, w, 3))
            for b in range(B):
                for _ in range(5):
                    y, x, a, c = rng.integers(h), rng.integers(w), rng.integers(3), rng.integers(C)
                    p[b, y, x, :, 4] = rng.uniform(3, 7, 3)
                    p[b, y, x, :, 5 + c] = rng.uniform(3, 7, 3)
                    p[b, y, min(x + 1, w - 1), a, 4] = 6
                    p[b, y, min(x + 1, w - 1), a, 5 + c] = 6
        out.append(p)
    return out


def _run(preds, anchors, in_hw, image_hw, obj, iou, max_out=30):
    import torch
    from k210_yolo_framework_amd import engine
    engine.require_gpu()
    B = preds[0].shape[0]
    C = preds[0].shape[-1] - 5
    cfg = engine.make_decode_cfg(anchors, C, in_hw, [p.shape[1:3] for p in preds])
    dev = [torch.from_numpy(p.reshape(B, p.shape[1], p.shape[2], -1)).cuda() for p in preds]
    dets, counts = engine.decode_py(cfg, dev, B, image_hw, obj, iou, max_out)
    torch.cuda.synchronize()
    return dets.cpu().numpy(), counts.cpu().numpy()


@pytest.mark.parametrize('kind,obj,iou,image_hw', [
    ('typical', 0.7, 0.5, None), ('uniform', 0.7, 0.5, None), ('typical', 0.7, 0.3, (374, 499)),
    ('uniform', 0.3, 0.45, (480, 640)), ('uniform', 0.05, 0.5, None)])
def test_decode_batch_vs_oracle(kind, obj, iou, image_hw):
    rng = np.random.default_rng(abs(hash((kind, obj))) % 2 ** 31)
    B = 6
    preds = _preds(rng, B, [(7, 10), (14, 20)], 20, kind)
    dets, counts = _run(preds, ANCHORS, (224, 320), image_hw, obj, iou)
    ref = dr.decode_batch(preds, ANCHORS, (224, 320), image_hw if image_hw else (224, 320), obj, iou)
    total = 0
    for b in range(B):
        rd, _ = ref[b]
        assert counts[b] == len(rd), (b, counts[b], len(rd))
        d = dets[b, :counts[b]]
        assert np.array_equal(d[:, 5], rd[:, 5])                                  # classes, class-major order
        np.testing.assert_allclose(d[:, 4], rd[:, 4], rtol=1e-5, atol=1e-6)       # scores
        np.testing.assert_allclose(d[:, :4], rd[:, :4], rtol=1e-5, atol=1e-3)     # pixels of the original image
        total += len(rd)
    assert total > 0


def test_decode_per_image_shapes_and_three_scales():
    rng = np.random.default_rng(5)
    B = 3
    anchors = rng.uniform(0.05, 0.8, (3, 3, 2))
    preds = _preds(rng, B, [(13, 13), (26, 26), (52, 52)], 4, 'typical')
    ihw = np.array([[416, 416], [300, 500], [720, 405]], np.float32)
    dets, counts = _run(preds, anchors, (416, 416), ihw, 0.6, 0.4)
    ref = dr.decode_batch(preds, anchors, (416, 416), ihw, 0.6, 0.4)
    for b in range(B):
        rd, _ = ref[b]
        assert counts[b] == len(rd)
        d = dets[b, :counts[b]]
        assert np.array_equal(d[:, 5], rd[:, 5])
        np.testing.assert_allclose(d[:, :5], rd[:, :5], rtol=1e-5, atol=2e-3)


def test_cap_30_per_class_and_overflow_path():
    """every box passes for class 0 (1050 candidates ... and > 2048 with 3 scales): max_output_size honoured."""
    rng = np.random.default_rng(9)
    preds = [rng.uniform(-1, 1, (1, h, w, 3, 6)).astype(np.float32) for (h, w) in [(13, 13), (26, 26), (52, 52)]]
    for p in preds:
        p[..., 4] = 8
        p[..., 5] = rng.uniform(3, 9, p.shape[:-1])
        p[..., 2:4] = -3
    anchors = np.full((3, 3, 2), 0.05)
    dets, counts = _run(preds, anchors, (416, 416), None, 0.5, 0.5)
    rd, _ = dr.decode_batch(preds, anchors, (416, 416), (416, 416), 0.5, 0.5)[0]
    assert counts[0] == len(rd) == 30
    np.testing.assert_allclose(dets[0, :30, :5], rd[:, :5], rtol=1e-5, atol=2e-3)


def test_no_detections():
    preds = [np.full((2, 7, 10, 3, 25), -9, np.float32), np.full((2, 14, 20, 3, 25), -9, np.float32)]
    dets, counts = _run(preds, ANCHORS, (224, 320), None, 0.7, 0.5)
    assert counts.tolist() == [0, 0]


def test_overflow_chunks_ties_and_suppression_carried_across_chunks():
    """10 647 candidates with IDENTICAL (saturated) scores: order falls to the box index, the first 2 535 boxes are giants that the
    very first one suppresses, the survivors come from the third scale - so the kernel must walk several LDS chunks in key order and
    apply the boxes selected in earlier chunks to later ones."""
    rng = np.random.default_rng(11)
    preds = [rng.uniform(-1, 1, (2, h, w, 3, 6)).astype(np.float32) for (h, w) in [(13, 13), (26, 26), (52, 52)]]
    for li, p in enumerate(preds):
        p[..., 4] = 30.0                      # sigmoid == 1.0 exactly in fp32
        p[..., 5] = 30.0
        p[..., 2:4] = 5.0 if li < 2 else -3.0
    preds[2][1, ..., 5] = rng.uniform(2, 30, preds[2][1, ..., 5].shape)     # image 1: third scale with distinct scores
    anchors = np.full((3, 3, 2), 0.05)
    dets, counts = _run(preds, anchors, (416, 416), None, 0.5, 0.5)
    ref = dr.decode_batch(preds, anchors, (416, 416), (416, 416), 0.5, 0.5)
    for b in range(2):
        rd, _ = ref[b]
        assert counts[b] == len(rd) == 30
        np.testing.assert_allclose(dets[b, :30, :5], rd[:, :5], rtol=1e-5, atol=2e-3)
    # image 0: the first box, then third-scale boxes in index order (all scores tie)
    assert dets[0, 0, 2] - dets[0, 0, 0] > 1000 and (dets[0, 1:30, 2] - dets[0, 1:30, 0] < 5).all()


@pytest.mark.parametrize('seed,levels,conf_shift', [(0, 4, 0.0), (1, 7, 1.0), (2, 64, -1.0), (3, 2, 2.5), (4, 100000, 0.5)])
def test_many_candidates_with_tied_scores_every_nms_path_vs_oracle(seed, levels, conf_shift):
    """Round 6 rewrote the per-class NMS for more than 512 candidates (a bitonic sort + TF's own sweep, chunk bounds by radix select): random
    heads of tiny_yolo's size (2 535 boxes) and of Darknet-53's (10 647) whose logits are QUANTISED to a few levels - hundreds to thousands
    of exactly tied scores per class, so the order falls to the box index - with overlapping boxes, for candidate counts on both sides of
    512, of the LDS capacity (2 048) and across several chunks.  Detections must equal the oracle's row for row."""
    rng = np.random.default_rng(100 + seed)
    for hw, in_hw in ([(13, 13), (26, 26)], (416, 416)), ([(13, 13), (26, 26), (52, 52)], (416, 416)):
        C = 6
        preds = []
        for (h, w) in hw:
            p = rng.normal(0, 1.5, (2, h, w, 3, 5 + C)).astype(np.float32)
            q = np.round((p + conf_shift) * levels / 6.0) * 6.0 / levels                 # few distinct logits -> few distinct scores
            p[..., 4:] = q[..., 4:]
            p[..., 2:4] = rng.uniform(-1.0, 1.5, p[..., 2:4].shape)                      # boxes of many sizes: plenty of suppression
            preds.append(p.astype(np.float32))
        anchors = np.tile(ANCHORS[:1], (len(hw), 1, 1)) * np.linspace(1.0, 0.3, len(hw))[:, None, None]
        for obj, iou in ((0.3, 0.5), (0.05, 0.3), (0.6, 0.7)):
            dets, counts = _run(preds, anchors, in_hw, None, obj, iou)
            ref = dr.decode_batch(preds, anchors, in_hw, in_hw, obj, iou)
            for b in range(2):
                rd, _ = ref[b]
                assert counts[b] == len(rd), (hw, obj, iou, b, counts[b], len(rd))
                d = dets[b, :counts[b]]
                assert np.array_equal(d[:, 5], rd[:, 5])
                np.testing.assert_allclose(d[:, :5], rd[:, :5], rtol=1e-5, atol=2e-3)


def _logit(p):
    p = np.asarray(p, np.float64)
    return np.log(p / (1.0 - p))


def _heads_for_boxes(boxes_yxyx, scores, W=128.0):
    """Logits of a two-scale head ([1,1,1,3,6] + [1,1,1,3,6], anchors 1.0, one class, 128x128 network and image) whose python-mode decode
    gives `boxes_yxyx` (pixels, (y1, x1, y2, x2)) and `scores`, in this order: (layer, h, w, anchor) - keras_inference.py:107-108."""
    assert len(boxes_yxyx) == 6
    preds = []
    for l in range(2):
        p = np.zeros((1, 1, 1, 3, 6), np.float32)
        for a in range(3):
            y1, x1, y2, x2 = boxes_yxyx[l * 3 + a]
            p[0, 0, 0, a, 0] = _logit(((x1 + x2) / 2) / W)               # x = (sigmoid(tx) + 0) / 1
            p[0, 0, 0, a, 1] = _logit(((y1 + y2) / 2) / W)
            p[0, 0, 0, a, 2] = np.log((x2 - x1) / W)                     # w = exp(tw) * anchor (1.0), relative to the image
            p[0, 0, 0, a, 3] = np.log((y2 - y1) / W)
            p[0, 0, 0, a, 4] = 30.0                                      # sigmoid == 1.0 in fp32
            p[0, 0, 0, a, 5] = _logit(scores[l * 3 + a])
        preds.append(p)
    return preds


@pytest.mark.parametrize('max_out,want', [(3, [3, 0, 5]), (2, [3, 0]), (30, [3, 0, 5])])
def test_tensorflows_own_nms_known_answers_through_the_hip_kernel(max_out, want):
    """tf.image.non_max_suppression's published test vectors (tensorflow/core/kernels/non_max_suppression_op_test.cc:
    TestSelectFromThreeClusters / AtMostTwoBoxes / AtMostThirtyBoxes; the same six boxes in image_ops_test.py) pushed through the whole
    python-mode path on the GPU: logits -> decode -> `>=` mask -> per-class NMS -> compaction, with the box INDEX of every kept row.
    (The x axis is shifted by +1 pixel: a box centre must lie inside the image for the sigmoid; IoU does not care.)"""
    import torch
    from k210_yolo_framework_amd import engine
    boxes = np.array([[0, 0, 1, 1], [0, 0.1, 1, 1.1], [0, -0.1, 1, 0.9], [0, 10, 1, 11], [0, 10.1, 1, 11.1], [0, 100, 1, 101]], np.float64)
    boxes[:, [1, 3]] += 1.0
    boxes[:, [0, 2]] += 1.0
    scores = np.array([.9, .75, .6, .95, .5, .3])
    preds = _heads_for_boxes(boxes, scores)
    anchors = np.ones((2, 3, 2))
    cfg = engine.make_decode_cfg(anchors, 1, (128, 128), [(1, 1), (1, 1)])
    dev = [torch.from_numpy(p.reshape(1, 1, 1, -1)).cuda() for p in preds]
    dets, counts, index = engine.decode_py(cfg, dev, 1, None, 0.2, 0.5, max_out, return_index=True)
    torch.cuda.synchronize()
    n = int(counts[0])
    assert index[0, :n].cpu().tolist() == want
    d = dets[0, :n].cpu().numpy()
    np.testing.assert_allclose(d[:, :4], boxes[want], atol=2e-3)
    np.testing.assert_allclose(d[:, 4], scores[want], atol=1e-6)
    ref, ridx = dr.decode_batch(preds, anchors, (128, 128), (128, 128), 0.2, 0.5, max_out)[0]
    assert ridx.tolist() == want
