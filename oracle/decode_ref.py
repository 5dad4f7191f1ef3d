"""decode_ref.py — CPU ORACLE (test infrastructure, NOT the product path).

numpy fp32 restatement of the reference's Python decode + per-class NMS:

  keras_inference.py:94-111   split / sigmoid*sigmoid scores / reshape(-1,..)
  tools/utils.py:524-547      tf_xywh_to_all
  keras_inference.py:32-72    correct_box
  keras_inference.py:113-135  `>= obj_thresh` mask + per-class tf.image.non_max_suppression

PARITY UNPINNED for the third-party part: `tf.sigmoid/exp/round` and
`tf.image.non_max_suppression(max_output_size=30)` live in tensorflow_gpu==1.14.0
(requirements.txt:3, not vendored, not installable here); the reference has no test
vectors at this call site.  NMS is restated from TF 1.14's published algorithm
(core/kernels/non_max_suppression_op.cc): greedy in descending score order, a candidate
is dropped iff IoU(candidate, any selected) > iou_threshold (strict), boxes are
normalised with min/max of their corners, zero/negative area => IoU 0, stop at
max_output_size.  TF 1.14 leaves the order of equal scores unspecified (heap order);
this restatement — and the HIP kernel — use ascending box index (what later TF
versions define).  The pure-numpy members are anchored by hand-derived known answers in
tests/test_oracle_decode.py (e.g. SURVEY.md 8(b): people.jpg => new_shape (224,299)).

Every arithmetic step is done in np.float32, one rounding per TF op.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

F = np.float32


def sigmoid(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, F)
    return (F(1) / (F(1) + np.exp(-x, dtype=F))).astype(F)


def xywh_to_all(pred_xy: np.ndarray, pred_wh: np.ndarray, out_hw: Tuple[int, int], anchors: np.ndarray):
    """tools/utils.py:545-546.  pred_* [..., h, w, A, 2]."""
    h, w = out_hw
    gy, gx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    offset = np.stack([gx, gy], -1)[:, :, None, :].astype(F)          # [h,w,1,2] = (col,row), utils.py:250-252
    wh = np.array([w, h], F)
    all_xy = ((sigmoid(pred_xy) + offset).astype(F) / wh).astype(F)
    all_wh = (np.exp(np.asarray(pred_wh, F), dtype=F) * np.asarray(anchors, F)).astype(F)
    return all_xy, all_wh


def correct_box(box_xy: np.ndarray, box_wh: np.ndarray, input_shape, image_shape) -> np.ndarray:
    """keras_inference.py:51-72 -> [..., 4] = ymin,xmin,ymax,xmax in original-image pixels."""
    box_yx = box_xy[..., ::-1].astype(F)
    box_hw = box_wh[..., ::-1].astype(F)
    inp = np.asarray(input_shape, F)
    img = np.asarray(image_shape, F)
    new_shape = np.round((img * np.min((inp / img).astype(F))).astype(F)).astype(F)   # tf.round: half-to-even
    offset = (((inp - new_shape).astype(F) / F(2.)).astype(F) / inp).astype(F)
    scale = (inp / new_shape).astype(F)
    box_yx = ((box_yx - offset).astype(F) * scale).astype(F)
    box_hw = (box_hw * scale).astype(F)
    half = (box_hw / F(2.)).astype(F)
    mins = (box_yx - half).astype(F)
    maxes = (box_yx + half).astype(F)
    boxes = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], -1)
    return (boxes * np.concatenate([img, img])).astype(F)


def tf_iou(boxes: np.ndarray, i: int, j: int) -> np.float32:
    """TF 1.14 non_max_suppression_op.cc IOU on (y1,x1,y2,x2) rows, fp32."""
    bi, bj = boxes[i], boxes[j]
    ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
    ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
    ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
    area_i = F(F(ymax_i - ymin_i) * F(xmax_i - xmin_i))
    area_j = F(F(ymax_j - ymin_j) * F(xmax_j - xmin_j))
    if area_i <= 0 or area_j <= 0:
        return F(0)
    iy0, ix0 = max(ymin_i, ymin_j), max(xmin_i, xmin_j)
    iy1, ix1 = min(ymax_i, ymax_j), min(xmax_i, xmax_j)
    inter = F(max(F(iy1 - iy0), F(0)) * max(F(ix1 - ix0), F(0)))
    return F(inter / F(F(area_i + area_j) - inter))


def non_max_suppression(boxes: np.ndarray, scores: np.ndarray, max_output_size: int, iou_threshold: float):
    """-> indices into boxes, selection order (score descending, ties by ascending index)."""
    order = sorted(range(len(scores)), key=lambda k: (-float(scores[k]), k))
    thr = F(iou_threshold)
    sel: List[int] = []
    for c in order:
        if len(sel) >= max_output_size:
            break
        keep = True
        for s in reversed(sel):
            if tf_iou(boxes, c, s) > thr:
                keep = False
                break
        if keep:
            sel.append(c)
    return sel


def decode_boxes_scores(preds: Sequence[np.ndarray], anchors: np.ndarray, in_hw, image_hw):
    """One image.  preds[l]: [h,w,A,5+C] fp32.  -> boxes [N,4], scores [N,C] in the reference's
    (layer, h, w, anchor) row order (keras_inference.py:107-114)."""
    bl, sl = [], []
    for l, p in enumerate(preds):
        p = np.asarray(p, F)
        h, w = p.shape[0], p.shape[1]
        scores = (sigmoid(p[..., 5:]) * sigmoid(p[..., 4:5])).astype(F)
        xy, wh = xywh_to_all(p[..., 0:2], p[..., 2:4], (h, w), anchors[l])
        boxes = correct_box(xy, wh, in_hw, image_hw)
        bl.append(boxes.reshape(-1, 4))
        sl.append(scores.reshape(-1, scores.shape[-1]))
    return np.concatenate(bl, 0), np.concatenate(sl, 0)


def decode_image(preds: Sequence[np.ndarray], anchors: np.ndarray, in_hw, image_hw, obj_thresh: float,
                 iou_thresh: float, max_out: int = 30):
    """-> (dets [K,6] = top,left,bottom,right,score,class ; box_index [K]) for one image."""
    boxes, scores = decode_boxes_scores(preds, anchors, in_hw, image_hw)
    mask = scores >= F(obj_thresh)
    rows, idxs = [], []
    for c in range(scores.shape[1]):
        cand = np.nonzero(mask[:, c])[0]
        if cand.size == 0:
            continue
        sel = non_max_suppression(boxes[cand], scores[cand, c], max_out, iou_thresh)
        for s in sel:
            g = cand[s]
            rows.append([*boxes[g], scores[g, c], F(c)])
            idxs.append(g)
    if not rows:
        return np.zeros((0, 6), F), np.zeros((0,), np.int64)
    return np.asarray(rows, F), np.asarray(idxs, np.int64)


def decode_batch(preds: Sequence[np.ndarray], anchors, in_hw, image_hw, obj_thresh, iou_thresh, max_out=30):
    """preds[l]: [B,h,w,A,5+C]; image_hw: (h,w) or [B,2]."""
    B = preds[0].shape[0]
    ihw = np.broadcast_to(np.asarray(image_hw, F), (B, 2))
    return [decode_image([p[b] for p in preds], anchors, in_hw, ihw[b], obj_thresh, iou_thresh, max_out)
            for b in range(B)]


def decode_batch_fast(preds: Sequence[np.ndarray], anchors, in_hw, image_hw, obj_thresh, iou_thresh, max_out=30, threads: int = 1):
    """decode_batch with the per-class mask + greedy NMS done by oracle/decode_nms_ref.c (the same operations in the same order, held
    bit-equal to decode_image by tests/test_oracle_decode.py; images in parallel on `threads` OpenMP threads).  The box / score
    arithmetic stays the numpy code above.  This is what bench.py's cpu_baseline times."""
    import ctypes as C
    import oracle
    L = oracle.lib()
    B = preds[0].shape[0]
    ihw = np.broadcast_to(np.asarray(image_hw, F), (B, 2))
    bs = [decode_boxes_scores([p[b] for p in preds], anchors, in_hw, ihw[b]) for b in range(B)]
    boxes = np.ascontiguousarray(np.stack([x[0] for x in bs]), F)
    scores = np.ascontiguousarray(np.stack([x[1] for x in bs]), F)
    n, nc = scores.shape[1], scores.shape[2]
    cap = nc * max_out
    rows = np.zeros((B, cap, 6), F)
    idx = np.zeros((B, cap), np.int32)
    counts = np.zeros((B,), np.int32)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    L.yk_ref_nms_batch(boxes.ctypes.data_as(fp), scores.ctypes.data_as(fp), C.c_int(B), C.c_int(n), C.c_int(nc), C.c_float(obj_thresh),
                       C.c_float(iou_thresh), C.c_int(max_out), rows.ctypes.data_as(fp), idx.ctypes.data_as(ip), counts.ctypes.data_as(ip),
                       C.c_int(int(threads)))
    return [(rows[b, :counts[b]].copy(), idx[b, :counts[b]].astype(np.int64)) for b in range(B)]
