"""PASCAL-VOC detection metric (per-class average precision, mAP) for the rows this package's decode produces.

The reference publishes no mAP and ships no evaluator (README.md shows demo pictures only); BASELINE.json's north star asks for
"VOC mAP within 0.1 pt of the Keras reference", so the metric is stated here once, the way the VOC devkit (VOCevaldet.m / the
widely copied `voc_eval.py` of py-faster-rcnn) defines it:

  * per class, detections of ALL images are visited in descending score order;
  * a detection is compared with ALL ground-truth boxes of ITS image and class (taken or not): the one with the highest IoU decides.
    IoU >= `iou_thresh` (0.5) on a box nobody has taken yet: true positive, the box is then taken; on an already taken box: FALSE
    positive (the devkit does not fall back to the second-best box); below the threshold: false positive;
  * boxes flagged `difficult` are neither positives nor negatives: they do not count in the recall denominator and a detection
    matched to one is ignored;
  * AP = area under the monotone precision envelope (VOC2010+, `use_07_metric=False`) or the 11-point mean of it (VOC2007);
  * mAP = mean over the classes that have at least one (non-difficult) ground-truth box.

Row format on both sides: (top, left, bottom, right, score, class) — what `yk_decode_py` / keras_inference.py:133-135 emit; the
ground truth uses the same six columns (score ignored).  Pure numpy; used by tools/map_eval.py and the -m gpu accuracy test.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def box_iou(box: np.ndarray, boxes: np.ndarray, plus_one: bool = False) -> np.ndarray:
    """IoU of one (top,left,bottom,right) box with [n,4] boxes.  plus_one: the devkit's inclusive integer-pixel convention."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
    box = np.asarray(box, np.float64)
    o = 1.0 if plus_one else 0.0
    ih = np.minimum(box[2], boxes[:, 2]) - np.maximum(box[0], boxes[:, 0]) + o
    iw = np.minimum(box[3], boxes[:, 3]) - np.maximum(box[1], boxes[:, 1]) + o
    inter = np.clip(ih, 0.0, None) * np.clip(iw, 0.0, None)
    area = (box[2] - box[0] + o) * (box[3] - box[1] + o)
    areas = (boxes[:, 2] - boxes[:, 0] + o) * (boxes[:, 3] - boxes[:, 1] + o)
    union = area + areas - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(union > 0, inter / union, 0.0)


def average_precision(recall: np.ndarray, precision: np.ndarray, use_07_metric: bool = False) -> float:
    """AP of one precision / recall curve (points in detection order)."""
    recall, precision = np.asarray(recall, np.float64), np.asarray(precision, np.float64)
    if use_07_metric:
        ap = 0.0
        for t in np.arange(0.0, 1.1, 0.1):
            above = precision[recall >= t - 1e-12]
            ap += (above.max() if above.size else 0.0) / 11.0
        return float(ap)
    r = np.concatenate(([0.0], recall, [1.0]))
    p = np.concatenate(([0.0], precision, [0.0]))
    p = np.maximum.accumulate(p[::-1])[::-1]            # monotone envelope
    step = np.nonzero(r[1:] != r[:-1])[0]
    return float(np.sum((r[step + 1] - r[step]) * p[step + 1]))


def evaluate(detections: Sequence[np.ndarray], ground_truth: Sequence[np.ndarray], class_num: int, iou_thresh: float = 0.5,
             use_07_metric: bool = False, difficult: Optional[Sequence[np.ndarray]] = None,
             plus_one: bool = False) -> Dict[str, object]:
    """detections[i]: [k,6] rows of image i; ground_truth[i]: [g,6] (or [g,5+]: columns 0-3 box, LAST column class) of image i;
    difficult[i]: [g] bool.  -> {'map', 'ap' [class_num] (nan where a class has no ground truth), 'n_gt' [class_num],
    'n_det' [class_num], 'tp' [class_num], 'fp' [class_num]}."""
    if len(detections) != len(ground_truth):
        raise ValueError(f'{len(detections)} images of detections, {len(ground_truth)} of ground truth')
    n_img = len(detections)
    def rows2d(a, width):
        a = np.asarray(a, np.float64)
        if a.size == 0:
            return np.zeros((0, width))
        return a[None] if a.ndim == 1 else a
    gts = [rows2d(g, 6) for g in ground_truth]
    dets = [rows2d(d, 6) for d in detections]
    diff = [np.zeros(len(g), bool) if difficult is None else np.asarray(difficult[i], bool) for i, g in enumerate(gts)]
    ap = np.full(class_num, np.nan)
    n_gt = np.zeros(class_num, int)
    n_det = np.zeros(class_num, int)
    tps = np.zeros(class_num, int)
    fps = np.zeros(class_num, int)
    for c in range(class_num):
        gt_c, taken, hard = [], [], []
        for i in range(n_img):
            sel = gts[i][:, -1].astype(int) == c if len(gts[i]) else np.zeros(0, bool)
            gt_c.append(gts[i][sel, :4])
            hard.append(diff[i][sel])
            taken.append(np.zeros(int(sel.sum()), bool))
        n_gt[c] = sum(int((~h).sum()) for h in hard)
        rows = [(d[k, 4], i, d[k, :4]) for i, d in enumerate(dets) for k in np.nonzero(d[:, 5].astype(int) == c)[0]]
        n_det[c] = len(rows)
        # descending score; ties keep image / row order (a stable sort, as the devkit's)
        order = sorted(range(len(rows)), key=lambda k: -rows[k][0])
        tp, fp = np.zeros(len(rows)), np.zeros(len(rows))
        for rank, k in enumerate(order):
            _, i, box = rows[k]
            best, j = -1.0, -1
            if len(gt_c[i]):
                iou = box_iou(box, gt_c[i], plus_one)
                j = int(np.argmax(iou))
                best = float(iou[j])
            if best >= iou_thresh:
                if hard[i][j]:
                    continue                             # ignored
                if not taken[i][j]:
                    tp[rank] = 1.0
                    taken[i][j] = True
                else:
                    fp[rank] = 1.0
            else:
                fp[rank] = 1.0
        tps[c], fps[c] = int(tp.sum()), int(fp.sum())
        if n_gt[c] == 0:
            continue
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        rec = ctp / n_gt[c]
        prec = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        ap[c] = average_precision(rec, prec, use_07_metric)
    have = ~np.isnan(ap)
    return {'map': float(ap[have].mean()) if have.any() else float('nan'), 'ap': ap, 'n_gt': n_gt, 'n_det': n_det, 'tp': tps, 'fp': fps}


def split_rows(rows: np.ndarray, offsets: np.ndarray) -> List[np.ndarray]:
    """The packed form of engine.Ticket.result() / yk_decode_py_packed -> one [k,6] array per image."""
    return [np.asarray(rows[offsets[b]:offsets[b + 1]]) for b in range(len(offsets) - 1)]


def padded_rows(dets: np.ndarray, counts: np.ndarray) -> List[np.ndarray]:
    """The padded form of yk_decode_py ([B, C*max_out, 6] + counts [B]) -> one [k,6] array per image."""
    return [np.asarray(dets[b, :int(counts[b])]) for b in range(len(counts))]


def map_delta(candidate: Sequence[np.ndarray], reference: Sequence[np.ndarray], ground_truth: Sequence[np.ndarray], class_num: int,
              **kw) -> Tuple[float, float, float]:
    """(mAP of `candidate`, mAP of `reference`, their difference in POINTS = 100 x) on the same ground truth: the quantity the
    north star bounds by 0.1."""
    a = evaluate(candidate, ground_truth, class_num, **kw)['map']
    b = evaluate(reference, ground_truth, class_num, **kw)['map']
    return a, b, 100.0 * (a - b)
