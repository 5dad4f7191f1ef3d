"""Host side of `make train` (row N3): the batch generator that replaces tools/utils.py:417-450 and the CLI contract."""
import numpy as np
import pytest

from k210_yolo_framework_amd import engine, training
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS


def _helper():
    return Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])


def test_batches_shapes_drop_remainder_and_label_content():
    h = _helper()
    items = training.synthetic_list(11, (224, 320), 20, seed=3)
    got = list(training.batches(h, items, 4, np.random.default_rng(0), shuffle=True))
    assert len(got) == 2                                                  # 11 // 4, remainder dropped (utils.py:447)
    x, ys = got[0]
    assert x.shape == (4, 224, 320, 3) and x.dtype == np.float32 and x.max() == 1.0 and x.min() >= 0.0   # img / np.max(img)
    assert [y.shape for y in ys] == [(4, 7, 10, 3, 25), (4, 14, 20, 3, 25)]
    n_obj = sum(int(y[..., 4].sum()) for y in ys)
    assert 4 <= n_obj <= 12                                               # 1..3 boxes per image, each lands in exactly one cell
    # unshuffled order is the list order and the labels are what box_to_label gives for the (already letterboxed) boxes
    x0, y0 = next(training.batches(h, items, 2, np.random.default_rng(0), shuffle=False))
    ref = h.box_to_label(np.array(items[1][1], np.float64))
    assert np.array_equal(y0[1][1], ref[1].astype(np.float32))


def test_batches_letterboxes_other_aspect_ratios_and_moves_the_boxes():
    h = _helper()
    img = np.full((100, 100, 3), 200, np.uint8)
    boxes = np.array([[3, 0.5, 0.5, 0.4, 0.4]])
    x, ys = next(training.batches(h, [(img, boxes)], 1, np.random.default_rng(0), shuffle=False))
    assert x[0, :, :40].max() == 0 and x[0, :, 290:].max() == 0 and x[0, 112, 160, 0] == 1.0     # 224x224 image centred in 224x320
    lab = h.label_to_box(ys, 0.7)
    # translation = int((320 - 100*2.24)/2) = int(47.99999..) = 47: the reference's truncation (utils.py:385), reproduced
    np.testing.assert_allclose(lab[0], [3, (112 + 47) / 320, 0.5, 0.4 * 224 / 320, 0.4], atol=1e-6)


def test_cli_refuses_to_run_without_gpu_and_rejects_unsupported_modes():
    import torch
    if torch.cuda.is_available():
        pytest.skip('CPU-only contract')
    with pytest.raises(engine.YkError):
        training.cli(['--synthetic', '8', '--max_nrof_epochs', '1'])
    with pytest.raises(engine.YkError, match='pruning'):
        training.cli(['--synthetic', '8', '--is_prune', 'True'])
    with pytest.raises(engine.YkError, match='imgaug'):
        training.cli(['--synthetic', '8', '--augmenter', 'True'])
