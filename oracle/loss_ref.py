"""loss_ref.py — CPU ORACLE (test infrastructure, NOT the product path).

numpy fp32 restatement of the reference's training-step arithmetic at the loss level:

  tools/utils.py:708-793   create_loss_fn / loss_fn           (rows T2)
  tools/utils.py:662-705   calc_ignore_mask, :617-659 tf_iou  (row T3)
  tools/utils.py:524-572   tf_xywh_to_all / tf_xywh_to_grid
  tools/custom.py:13-75    Yolo_Precision / Yolo_Recall       (row T4; they threshold the RAW logit, custom.py:33)

PARITY UNPINNED against the reference itself (TensorFlow 1.14 ops, not installable; the reference has no
tests).  Substitute arbiter: tests/test_oracle_loss.py rebuilds the same loss with torch-CPU ops and checks the
value AND the analytic gradient dL/dy_pred given here against torch.autograd.

The gradient is what TF's autodiff produces for this graph: `ignore_mask` comes from a comparison
(`tf.cast(best_iou < iou_thresh, tf.float32)`, utils.py:704) and therefore carries no gradient.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

F = np.float32


def _sigmoid(x):
    return (F(1) / (F(1) + np.exp(-x, dtype=F))).astype(F)


def bce_logits(labels, logits):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1 + exp(-|x|))."""
    x = np.asarray(logits, F)
    z = np.asarray(labels, F)
    return (np.maximum(x, F(0)) - x * z + np.log1p(np.exp(-np.abs(x), dtype=F), dtype=F)).astype(F)


def iou_center(p_xy, p_wh, g_xy, g_wh):
    """tools/utils.py:617-659: [...,2] vs [n,2] -> [..., n]."""
    b1_xy, b1_wh = p_xy[..., None, :], p_wh[..., None, :]
    b1_min, b1_max = b1_xy - b1_wh / F(2), b1_xy + b1_wh / F(2)
    b2_min, b2_max = g_xy - g_wh / F(2), g_xy + g_wh / F(2)
    iw = np.maximum(np.minimum(b1_max, b2_max) - np.maximum(b1_min, b2_min), F(0))
    inter = iw[..., 0] * iw[..., 1]
    return (inter / (b1_wh[..., 0] * b1_wh[..., 1] + g_wh[..., 0] * g_wh[..., 1] - inter)).astype(F)


def yolo_loss(y_true: np.ndarray, y_pred: np.ndarray, anchors_l: np.ndarray, obj_thresh: float, iou_thresh: float,
              obj_weight: float, noobj_weight: float, wh_weight: float, batch_size: int = None):
    """One output layer.  y_true / y_pred: [B,h,w,A,5+C] fp32.
    -> dict(total, xy, wh, obj, noobj, cls), grad dL/dy_pred [B,h,w,A,5+C], ignore_mask [B,h,w,A], (tp, fp, fn)."""
    yt, yp = np.asarray(y_true, F), np.asarray(y_pred, F)
    B, h, w, A, E = yp.shape
    bs = F(batch_size if batch_size else B)
    anc = np.asarray(anchors_l, F)
    gy, gx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    offset = np.stack([gx, gy], -1)[:, :, None, :].astype(F)
    wh_hw = np.array([w, h], F)

    pxy, pwh, pconf, pcls = yp[..., 0:2], yp[..., 2:4], yp[..., 4:5], yp[..., 5:]
    txy, twh, tconf, tcls = yt[..., 0:2], yt[..., 2:4], yt[..., 4:5], yt[..., 5:]
    obj = tconf
    obj_bool = yt[..., 4] > F(obj_thresh)

    # ---- ignore mask (utils.py:693-705): per image, best IoU of every prediction against that image's GT boxes
    all_xy = ((_sigmoid(pxy) + offset) / wh_hw).astype(F)
    all_wh = (np.exp(pwh, dtype=F) * anc).astype(F)
    ignore = np.ones((B, h, w, A), F)
    for b in range(B):
        g_xy, g_wh = txy[b][obj_bool[b]], twh[b][obj_bool[b]]
        if len(g_xy):
            best = iou_center(all_xy[b], all_wh[b], g_xy, g_wh).max(-1)
            ignore[b] = (best < F(iou_thresh)).astype(F)

    # ---- targets in grid scale (utils.py:550-572,762-764)
    g_txy = (txy * wh_hw - offset).astype(F)
    with np.errstate(divide='ignore', invalid='ignore'):
        g_twh = np.log(twh / anc, dtype=F)
    g_twh = np.where(obj_bool[..., None], g_twh, F(0)).astype(F)
    cw = (F(2) - twh[..., 0:1] * twh[..., 1:2]).astype(F)

    xy_l = (obj * cw * bce_logits(g_txy, pxy)).sum(dtype=np.float64) / bs
    wh_l = (obj * cw * F(wh_weight) * np.square(g_twh - pwh)).sum(dtype=np.float64) / bs
    bce_c = bce_logits(tconf, pconf)
    obj_l = F(obj_weight) * (obj * bce_c).sum(dtype=np.float64) / bs
    noobj_l = F(noobj_weight) * ((F(1) - obj) * ignore[..., None] * bce_c).sum(dtype=np.float64) / bs
    cls_l = (obj * bce_logits(tcls, pcls)).sum(dtype=np.float64) / bs
    losses = dict(xy=float(xy_l), wh=float(wh_l), obj=float(obj_l), noobj=float(noobj_l), cls=float(cls_l))
    losses['total'] = float(obj_l + noobj_l + cls_l + xy_l + wh_l)      # utils.py:789

    # ---- analytic gradient (d BCE / d logit = sigmoid(x) - z)
    grad = np.zeros_like(yp)
    grad[..., 0:2] = obj * cw * (_sigmoid(pxy) - g_txy) / bs
    grad[..., 2:4] = obj * cw * F(wh_weight) * F(2) * (pwh - g_twh) / bs
    dconf = _sigmoid(pconf) - tconf
    grad[..., 4:5] = (F(obj_weight) * obj + F(noobj_weight) * (F(1) - obj) * ignore[..., None]) * dconf / bs
    grad[..., 5:] = obj * (_sigmoid(pcls) - tcls) / bs

    # ---- Yolo_Precision / Yolo_Recall counters (custom.py:29-40,61-72): raw logit vs threshold
    t_pos = yt[..., 4] > F(obj_thresh)
    p_pos = yp[..., 4] > F(obj_thresh)
    tp = int((t_pos & p_pos).sum())
    fp = int((~t_pos & p_pos).sum())
    fn = int((t_pos & ~p_pos).sum())
    return losses, grad.astype(F), ignore, (tp, fp, fn)
