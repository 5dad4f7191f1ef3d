"""N3 host logic (CPU): per-rank sharding of the epoch order and the box half of the letterbox, against Helper._process_img / batches()."""
import numpy as np

from k210_yolo_framework_amd import pipeline, training
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS


def _helper():
    return Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])


def test_ranks_partition_every_global_batch_and_drop_the_remainder():
    order = pipeline.epoch_order(103, seed=7, epoch=2, shuffle=True)
    assert sorted(order) == list(range(103))
    assert np.array_equal(order, pipeline.epoch_order(103, 7, 2, True))          # same on every rank
    assert not np.array_equal(order, pipeline.epoch_order(103, 7, 3, True))      # reshuffled every epoch
    rows = [pipeline.rank_rows(order, 16, r, 4) for r in range(4)]
    assert all(len(x) == 103 // 16 for x in rows)
    for step in range(103 // 16):
        got = np.concatenate([rows[r][step] for r in range(4)])
        assert np.array_equal(got, order[step * 16:(step + 1) * 16])             # rank r holds rows [r*4, r*4+4) of the global batch
    assert np.array_equal(pipeline.epoch_order(5, 0, 0, False), np.arange(5))


def test_box_letterbox_equals_process_img():
    h = _helper()
    rng = np.random.default_rng(0)
    for hw in ((240, 320), (375, 500), (500, 333), (224, 320)):
        boxes = np.concatenate([rng.integers(0, 20, (4, 1)).astype(float), rng.uniform(0.1, 0.9, (4, 4))], 1)
        img = rng.integers(1, 255, (*hw, 3), dtype=np.uint8)
        _, want = h._process_img(img, boxes.copy(), is_training=False, is_resize=True)
        assert np.array_equal(pipeline.letterbox_boxes(h, hw, boxes), want)
    assert pipeline.letterbox_boxes(h, (240, 320), np.zeros((0, 5))).shape == (0, 5)


def test_batch_label_encoder_and_batch_letterbox_equal_the_per_sample_functions():
    """The input pipeline encodes a whole batch with a handful of array operations (Helper.batch_box_to_label, letterbox_boxes_batch):
    bit-identical to the per-sample functions of tools/utils.py:207-230 / :386-389, including two boxes in one (cell, anchor) slot (the later
    one replaces xywh and adds its class bit), empty samples and mixed image sizes."""
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
    from k210_yolo_framework_amd.pipeline import letterbox_boxes, letterbox_boxes_batch
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    rng = np.random.default_rng(0)
    for trial in range(60):
        n = int(rng.integers(1, 9))
        hws = [(int(rng.integers(100, 500)), int(rng.integers(100, 600))) if rng.random() < 0.4 else (240, 320) for _ in range(n)]
        boxes = []
        for _ in range(n):
            k = int(rng.integers(0, 6))
            b = np.column_stack([rng.integers(0, 20, k), rng.uniform(0.05, 0.95, (k, 2)), rng.uniform(0.01, 0.5, (k, 2))]) if k else np.zeros((0, 5))
            if k >= 2 and rng.random() < 0.5:
                b[1, 1:5] = b[0, 1:5] * np.array([1, 1, 1.01, 0.99])
            boxes.append(b)
        moved_ref = [letterbox_boxes(h, hw, b) for hw, b in zip(hws, boxes)]
        moved = letterbox_boxes_batch(h, hws, boxes)
        for r, g in zip(moved_ref, moved):
            assert np.array_equal(np.asarray(r).reshape(-1, 5), g)
        ref = [np.stack(x) for x in zip(*[h.box_to_label(b) for b in moved_ref])]
        got = h.batch_box_to_label(moved)
        for r, g in zip(ref, got):
            assert r.dtype == g.dtype == np.float32 and np.array_equal(r, g)
