"""Offline dataset tools of the reference (row N3): `make_voc_list.py` and `make_anchor_list.py`, host-side numpy.

  make_voc_list.py:9-26     image list -> data/<set>_img_ann.npy rows [path, boxes[n,5] (cls,x,y,w,h), (h,w)]
  make_anchor_list.py:176-218  letterbox the boxes to the network frame, k-means on (w,h) under the centred-IoU
                               distance 1 - IoU (:10-39), 10 iterations (the reference passes a literal 10 at :207, not
                               --max_iters), centroids sorted by descending w, reshaped to [layers, anchors, 2]

The reference evaluates the distance in a TF-1 session; it is plain float64 arithmetic, restated here with numpy."""
from __future__ import annotations

import os
import re
from typing import Sequence

import numpy as np


def make_voc_list(train_file: str, output_file: str) -> np.ndarray:
    """train_file: one image path per line; labels sit beside them with JPEGImages->labels, .jpg->.txt."""
    from PIL import Image
    paths = [str(p) for p in np.atleast_1d(np.loadtxt(train_file, dtype=str))]
    rows = np.empty((len(paths), 3), dtype=object)
    for i, p in enumerate(paths):
        ann = re.sub(r'.jpg', '.txt', re.sub(r'JPEGImages', 'labels', p))
        with Image.open(p) as im:
            hw = np.array([im.height, im.width])
        rows[i, 0], rows[i, 1], rows[i, 2] = p, np.loadtxt(ann, dtype=float, ndmin=2), hw
    os.makedirs(os.path.dirname(os.path.abspath(output_file)), exist_ok=True)
    np.save(output_file, rows, allow_pickle=True)
    return rows


def fake_iou_distance(x: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """make_anchor_list.py:10-39: boxes centred at the origin, [m,2] vs [k,2] -> 1 - IoU, [m,k]."""
    x, c = x[:, None, :], centroids[None, :, :]
    wh = np.maximum(np.minimum(x / 2., c / 2.) - np.maximum(-x / 2., -c / 2.), 0.)
    inter = wh[..., 0] * wh[..., 1]
    return 1 - inter / (x[..., 0] * x[..., 1] + c[..., 0] * c[..., 1] - inter)


def run_kmeans(x: np.ndarray, initial_centroids: np.ndarray, iters: int = 10):
    """make_anchor_list.py:143-173 (empty clusters give NaN centroids, which the caller reports like the reference)."""
    c = np.array(initial_centroids, np.float64)
    idx = np.zeros(len(x), np.int64)
    for _ in range(iters):
        idx = np.argmin(fake_iou_distance(x, c), axis=1)
        with np.errstate(invalid='ignore'), np.testing.suppress_warnings() as sup:
            sup.filter(RuntimeWarning)
            c = np.stack([x[idx == i].mean(axis=0) if (idx == i).any() else np.full(x.shape[1], np.nan) for i in range(len(c))])
    return c, idx


def letterbox_boxes(rows: np.ndarray, in_hw: Sequence[int]) -> np.ndarray:
    """make_anchor_list.py:181-193: every annotation mapped into the network frame; returns all (w,h) stacked."""
    in_wh = np.array(in_hw[::-1], np.float64)
    out = []
    for r in rows:
        box = np.array(r[1], np.float64, copy=True)
        img_wh = np.array(r[2][::-1], np.float64)
        scale = in_wh / img_wh
        scale[:] = np.min(scale)
        translation = ((in_wh - img_wh * scale) / 2).astype(int)
        box[:, 1:3] = (box[:, 1:3] * img_wh * scale + translation) / in_wh
        box[:, 3:5] = (box[:, 3:5] * img_wh * scale) / in_wh
        out.append(box)
    return np.vstack(out)[:, 3:]


def make_anchor_list(train_set: str, in_hw=(224, 320), out_hw=(7, 10, 14, 20), anchor_num: int = 3, is_random: bool = False,
                     low=(0., 0.), high=(1., 1.), seed=None, data_dir: str = 'data', save: bool = True) -> np.ndarray:
    rows = np.load(os.path.join(data_dir, f'{train_set}_img_ann.npy'), allow_pickle=True)
    x = letterbox_boxes(rows, in_hw)
    layers = len(out_hw) // 2
    k = layers * anchor_num
    if is_random:
        rng = np.random.default_rng(seed)
        init = np.hstack((rng.uniform(low[0], high[0], (k, 1)), rng.uniform(low[1], high[1], (k, 1))))
    else:
        init = np.vstack((np.linspace(0.05, 0.3, num=k), np.linspace(0.05, 0.5, num=k))).T
    centroids, _ = run_kmeans(x, init, 10)
    centroids = np.array(sorted(centroids, key=lambda v: -v[0])).reshape(layers, anchor_num, 2)
    if np.any(np.isnan(centroids)):
        print('[ERROR] Result have NaN value please Rerun!')
    elif save:
        np.save(os.path.join(data_dir, f'{train_set}_anchor.npy'), centroids)
    return centroids
