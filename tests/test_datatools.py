"""Offline dataset tools (row N3): VOC list builder and anchor k-means, against hand-derived answers."""
import numpy as np
from PIL import Image

from k210_yolo_framework_amd import datatools, training
from k210_yolo_framework_amd.helper import Helper


def _dataset(tmp_path, n=48, seed=0):
    rng = np.random.default_rng(seed)
    (tmp_path / 'JPEGImages').mkdir()
    (tmp_path / 'labels').mkdir()
    paths = []
    for i in range(n):
        h, w = (240, 320) if i % 2 else (300, 200)
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / 'JPEGImages' / f'{i:03d}.jpg')
        k = 1 + i % 3
        boxes = np.stack([rng.integers(0, 20, k), rng.uniform(.3, .7, k), rng.uniform(.3, .7, k), rng.uniform(.05, .5, k), rng.uniform(.05, .5, k)], 1)
        np.savetxt(tmp_path / 'labels' / f'{i:03d}.txt', boxes.reshape(k, 5))
        paths.append(str(tmp_path / 'JPEGImages' / f'{i:03d}.jpg'))
    (tmp_path / 'train.txt').write_text('\n'.join(paths) + '\n')
    return paths


def test_make_voc_list_rows_and_training_pipeline_reads_them(tmp_path):
    paths = _dataset(tmp_path)
    (tmp_path / 'data').mkdir()
    rows = datatools.make_voc_list(str(tmp_path / 'train.txt'), str(tmp_path / 'data' / 'voc_img_ann.npy'))
    back = np.load(tmp_path / 'data' / 'voc_img_ann.npy', allow_pickle=True)
    assert back.shape == (48, 3) and back[0, 0] == paths[0]
    assert back[0, 1].shape == (1, 5) and back[2, 1].shape == (3, 5)             # ndmin=2 even for one box
    assert back[0, 2].tolist() == [300, 200] and back[1, 2].tolist() == [240, 320]   # (h, w)
    assert np.allclose(rows[5, 1], back[5, 1])
    # the list feeds Helper / training.batches unchanged
    anchors = datatools.make_anchor_list('voc', is_random=False, data_dir=str(tmp_path / 'data'))
    h = Helper(str(tmp_path / 'data' / 'voc_img_ann.npy'), 20, str(tmp_path / 'data' / 'voc_anchor.npy'), [[224, 320]], [[7, 10], [14, 20]], 0.25)
    assert h.train_total_data == 36 and h.test_total_data == 12 and np.array_equal(h.anchors, anchors)
    x, ys = next(training.batches(h, [(r[0], r[1]) for r in h.train_list], 4, np.random.default_rng(0), shuffle=False))
    assert x.shape == (4, 224, 320, 3) and [y.shape for y in ys] == [(4, 7, 10, 3, 25), (4, 14, 20, 3, 25)]


def test_fake_iou_distance_known_values():
    d = datatools.fake_iou_distance(np.array([[0.2, 0.4], [0.5, 0.5]]), np.array([[0.2, 0.4], [0.4, 0.2], [1.0, 1.0]]))
    assert np.allclose(d[0], [0.0, 1 - 0.04 / (0.08 + 0.08 - 0.04), 1 - 0.08 / 1.0])
    assert np.allclose(d[1], [1 - 0.08 / 0.25, 1 - 0.08 / 0.25, 1 - 0.25 / 1.0])


def test_kmeans_recovers_separated_clusters_sorted_by_width():
    rng = np.random.default_rng(1)
    centres = np.array([[0.08, 0.1], [0.2, 0.35], [0.4, 0.3], [0.6, 0.7], [0.8, 0.5], [0.95, 0.9]])
    x = np.vstack([c * rng.uniform(0.97, 1.03, (40, 2)) for c in centres])
    c, idx = datatools.run_kmeans(x, centres * 1.1, 10)
    assert np.allclose(c, centres, rtol=0.02)
    assert all(len(set(idx[i * 40:(i + 1) * 40])) == 1 for i in range(6))
    # one assignment + mean step by hand
    c1, idx1 = datatools.run_kmeans(np.array([[0.1, 0.1], [0.12, 0.1], [0.5, 0.5]]), np.array([[0.1, 0.1], [0.6, 0.6]]), 1)
    assert idx1.tolist() == [0, 0, 1] and np.allclose(c1, [[0.11, 0.1], [0.5, 0.5]])


def test_anchor_list_letterboxes_boxes_like_helper_and_orders_descending(tmp_path):
    _dataset(tmp_path)
    (tmp_path / 'data').mkdir()
    rows = datatools.make_voc_list(str(tmp_path / 'train.txt'), str(tmp_path / 'data' / 'voc_img_ann.npy'))
    wh = datatools.letterbox_boxes(rows, (224, 320))
    # image 0 is 300x200 (h,w): scale = min(320/200, 224/300) = 0.7467 -> w' = w*200*s/320, h' = h*300*s/224 = h
    s = 224 / 300
    assert np.allclose(wh[0], [rows[0, 1][0, 3] * 200 * s / 320, rows[0, 1][0, 4]])
    a = datatools.make_anchor_list('voc', is_random=False, data_dir=str(tmp_path / 'data'), save=False)
    if not np.isnan(a).any():
        flat = a.reshape(-1, 2)
        assert a.shape == (2, 3, 2) and (np.diff(flat[:, 0]) <= 1e-12).all()
