"""Host-side mirror of the reference's `tools/utils.py` API surface for the hot path.

Same names, argument meaning and array shapes as the reference (`Helper`, `tf_xywh_to_all`,
`tf_xywh_to_grid`, `tf_iou`), with numpy / torch tensors instead of `tf.Tensor`.  The pure-numpy
members are restated from tools/utils.py (file:line in each docstring); the heavy arithmetic
(model forward, decode, NMS, loss) is NOT here — it runs in libyolo_hip.so.

Out of scope: imgaug augmentation, the tf.data pipeline, matplotlib drawing.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

INFO, ERROR, NOTE = '[ INFO  ]', '[ ERROR ]', '[ NOTE  ]'     # tools/utils.py:15-17 (uncoloured)

# data/voc_anchor.npy of the reference == main.c:46-52 (w,h) relative to the whole image
VOC_ANCHORS = np.array([[[0.76120044, 0.57155991], [0.6923348, 0.88535553], [0.47163042, 0.34163313]],
                        [[0.33340788, 0.70065861], [0.18124964, 0.38986752], [0.08497349, 0.1527057]]], np.float64)


class Helper(object):
    """tools/utils.py:53-521."""

    def __init__(self, image_ann: Optional[str], class_num: int, anchors: Union[str, np.ndarray, None],
                 in_hw: Sequence, out_hw: Sequence, validation_split: float = 0.1):
        self.in_hw = np.array(in_hw)
        assert self.in_hw.ndim == 2
        self.out_hw = np.array(out_hw)
        assert self.out_hw.ndim == 2
        self.validation_split = validation_split
        if image_ann is None:
            self.train_list = None
            self.test_list = None
        else:
            img_ann_list = np.load(image_ann, allow_pickle=True)
            num = int(len(img_ann_list) * self.validation_split)
            self.train_list = img_ann_list[num:]
            self.test_list = img_ann_list[:num]
            self.train_total_data = len(self.train_list)
            self.test_total_data = len(self.test_list)
        self.grid_wh = (1 / self.out_hw)[:, [1, 0]]                         # utils.py:70
        if class_num:
            self.class_num = class_num
        if anchors is not None:
            self.anchors = np.load(anchors) if isinstance(anchors, str) else np.asarray(anchors)
            self.anchor_number = len(self.anchors[0])
            self.output_number = len(self.anchors)
            self.xy_offset = Helper._coordinate_offset(self.anchors, self.out_hw)
            self.wh_scale = Helper._anchor_scale(self.anchors, self.grid_wh)
        self.output_shapes = [[None] + list(self.out_hw[i]) + [len(self.anchors[i]), self.class_num + 5]
                              for i in range(len(self.anchors))]
        self.batch_size = None
        self.colormap = [                                                    # utils.py:89-105
            (255, 82, 0), (0, 255, 245), (0, 61, 255), (0, 255, 112), (0, 255, 133),
            (255, 0, 0), (255, 163, 0), (255, 102, 0), (194, 255, 0), (0, 143, 255),
            (51, 255, 0), (0, 82, 255), (0, 255, 41), (0, 255, 173), (10, 0, 255),
            (173, 255, 0), (0, 255, 153), (255, 92, 0), (255, 0, 255), (255, 0, 245),
            (128, 0, 0), (0, 128, 0), (128, 128, 0), (0, 0, 128), (128, 0, 128),
            (0, 128, 128), (128, 128, 128), (64, 0, 0), (192, 0, 0), (64, 128, 0),
            (192, 128, 0), (64, 0, 128), (192, 0, 128), (64, 128, 128), (192, 128, 128),
            (0, 64, 0), (128, 64, 0), (0, 192, 0), (128, 192, 0), (0, 64, 128)]

    # ---- label side (tools/utils.py:140-230) --------------------------------------------------
    def _xy_grid_index(self, box_xy: np.ndarray, layer: int):
        """utils.py:156: floor(xy * (w,h)) -> [idx, idy]."""
        return np.floor(box_xy * self.out_hw[layer][::-1]).astype('int')

    @staticmethod
    def _fake_iou(a: np.ndarray, b: np.ndarray):
        """utils.py:159-188: IoU of two boxes sharing a centre."""
        a_maxes = a / 2.
        a_mins = -a_maxes
        b_maxes = b / 2.
        b_mins = -b_maxes
        iner_wh = np.maximum(np.minimum(a_maxes, b_maxes) - np.maximum(a_mins, b_mins), 0.)
        iner_area = iner_wh[..., 0] * iner_wh[..., 1]
        s1 = a[..., 0] * a[..., 1]
        s2 = b[..., 0] * b[..., 1]
        return iner_area / (s1 + s2 - iner_area)

    def _get_anchor_index(self, wh: np.ndarray):
        """utils.py:190-205 -> (layer, anchor) of the best centred-IoU anchor over ALL layers."""
        iou = Helper._fake_iou(wh, self.anchors)
        return np.unravel_index(np.argmax(iou), iou.shape)

    def box_to_label(self, true_box: np.ndarray) -> List[np.ndarray]:
        """utils.py:207-230. true_box [n,5] = [cls,x,y,w,h] image-relative -> L x [h,w,A,5+C] fp32."""
        labels = [np.zeros((self.out_hw[i][0], self.out_hw[i][1], len(self.anchors[i]), 5 + self.class_num),
                           dtype='float32') for i in range(self.output_number)]
        for box in true_box:
            l, n = self._get_anchor_index(box[3:5])
            idx, idy = self._xy_grid_index(box[1:3], l)
            labels[l][idy, idx, n, 0:4] = np.clip(box[1:5], 1e-8, 1.)
            labels[l][idy, idx, n, 4] = 1.
            labels[l][idy, idx, n, 5 + int(box[0])] = 1.
        return labels

    @staticmethod
    def _coordinate_offset(anchors: np.ndarray, out_hw: np.ndarray) -> np.ndarray:
        """utils.py:233-253: per layer [h,w,1,2] with [...,0]=col (x), [...,1]=row (y)."""
        grid = []
        for l in range(len(anchors)):
            gy = np.tile(np.reshape(np.arange(0, stop=out_hw[l][0]), [-1, 1, 1, 1]), [1, out_hw[l][1], 1, 1])
            gx = np.tile(np.reshape(np.arange(0, stop=out_hw[l][1]), [1, -1, 1, 1]), [out_hw[l][0], 1, 1, 1])
            grid.append(np.concatenate([gx, gy], axis=-1))
        return np.array(grid, dtype=object) if len({g.shape for g in grid}) > 1 else np.array(grid)

    @staticmethod
    def _anchor_scale(anchors: np.ndarray, grid_wh: np.ndarray) -> np.ndarray:
        """utils.py:256-271."""
        return np.array([anchors[i] * grid_wh[i] for i in range(len(anchors))])

    def _xy_to_all(self, labels):
        """utils.py:273-281."""
        for i in range(len(labels)):
            labels[i][..., 0:2] = labels[i][..., 0:2] * self.grid_wh[i] + self.xy_offset[i]

    def _wh_to_all(self, labels):
        """utils.py:283-291."""
        for i in range(len(labels)):
            labels[i][..., 2:4] = np.exp(labels[i][..., 2:4]) * self.anchors[i]

    def label_to_box(self, labels, thersh=.7) -> np.ndarray:
        """utils.py:293-307."""
        new_boxs = np.vstack([label[np.where(label[..., 4] > thersh)] for label in labels])
        return np.c_[np.argmax(new_boxs[:, 5:], axis=-1), new_boxs[:, :4]]

    # ---- image side (tools/utils.py:339-406) --------------------------------------------------
    def _read_img(self, img_path: str) -> np.ndarray:
        """utils.py:339-355 (skimage.io.imread -> PIL): RGB uint8, gray->rgb, alpha dropped."""
        from PIL import Image
        img = np.asarray(Image.open(img_path))
        if img.ndim != 3:
            img = np.stack([img] * 3, -1)
        return img[..., :3]

    def letterbox_params(self, img_hw) -> Tuple[float, np.ndarray]:
        """utils.py:378-385: scale = min(in_wh/img_wh); translation = ((in_wh - img_wh*scale)/2).astype(int)."""
        img_wh = np.array([img_hw[1], img_hw[0]])
        in_wh = self.in_hw[0][::-1]
        scale = in_wh / img_wh
        scale[:] = np.min(scale)
        translation = ((in_wh - img_wh * scale) / 2).astype(int)
        return scale, translation

    def _process_img(self, img: np.ndarray, true_box, is_training: bool, is_resize: bool):
        """utils.py:357-406 without augmentation: letterbox (bilinear, zero fill, truncating uint8
        cast) then `img / np.max(img)`.  The warp restates skimage.transform.warp(order=1,
        mode='constant', cval=0) — third-party, parity unpinned except for the identity case."""
        if is_resize:
            scale, translation = self.letterbox_params(img.shape[:2])
            if isinstance(true_box, np.ndarray):
                img_wh = np.array([img.shape[1], img.shape[0]])
                in_wh = self.in_hw[0][::-1]
                true_box[:, 1:3] = (true_box[:, 1:3] * img_wh * scale + translation) / in_wh
                true_box[:, 3:5] = (true_box[:, 3:5] * img_wh * scale) / in_wh
            img = letterbox_bilinear(img, tuple(self.in_hw[0]), float(scale[0]), translation)
        if is_training:
            raise NotImplementedError('imgaug augmentation is out of scope (SURVEY.md §2 #6)')
        img = img / np.max(img)
        return img, true_box

    # ---- box format helpers (tools/utils.py:492-521) ------------------------------------------
    def center_to_corner(self, true_box, to_all_scale=True):
        sx, sy = (self.in_hw[0, 1], self.in_hw[0, 0]) if to_all_scale else (1, 1)
        x1 = (true_box[:, 0:1] - true_box[:, 2:3] / 2) * sx
        y1 = (true_box[:, 1:2] - true_box[:, 3:4] / 2) * sy
        x2 = (true_box[:, 0:1] + true_box[:, 2:3] / 2) * sx
        y2 = (true_box[:, 1:2] + true_box[:, 3:4] / 2) * sy
        return np.hstack([x1, y1, x2, y2])

    def corner_to_center(self, xyxy_box, from_all_scale=True):
        sx, sy = (self.in_hw[0, 1], self.in_hw[0, 0]) if from_all_scale else (1, 1)
        x = ((xyxy_box[:, 2:3] + xyxy_box[:, 0:1]) / 2) / sx
        y = ((xyxy_box[:, 3:4] + xyxy_box[:, 1:2]) / 2) / sy
        w = (xyxy_box[:, 2:3] - xyxy_box[:, 0:1]) / sx
        h = (xyxy_box[:, 3:4] - xyxy_box[:, 1:2]) / sy
        return np.hstack([x, y, w, h])


def letterbox_bilinear(img: np.ndarray, out_hw: Tuple[int, int], scale: float, translation) -> np.ndarray:
    """Inverse-mapped bilinear warp: out(x,y) = in((x-tx)/s, (y-ty)/s), zero outside, uint8 truncation."""
    H, W = out_hw
    ih, iw = img.shape[:2]
    if scale == 1.0 and tuple(translation) == (0, 0) and (ih, iw) == (H, W):
        return img.astype('uint8')
    xs = (np.arange(W) - translation[0]) / scale
    ys = (np.arange(H) - translation[1]) / scale
    x0 = np.floor(xs).astype(int)
    y0 = np.floor(ys).astype(int)
    fx = (xs - x0)[None, :, None]
    fy = (ys - y0)[:, None, None]
    pad = np.zeros((ih + 2, iw + 2, img.shape[2]), np.float64)
    pad[1:-1, 1:-1] = img

    def at(yy, xx):
        yy = np.clip(yy + 1, 0, ih + 1)
        xx = np.clip(xx + 1, 0, iw + 1)
        return pad[yy[:, None], xx[None, :]]
    out = (at(y0, x0) * (1 - fy) * (1 - fx) + at(y0, x0 + 1) * (1 - fy) * fx +
           at(y0 + 1, x0) * fy * (1 - fx) + at(y0 + 1, x0 + 1) * fy * fx)
    inside = ((xs > -1) & (xs < iw))[None, :, None] & ((ys > -1) & (ys < ih))[:, None, None]
    return np.where(inside, out, 0.0).astype('uint8')


# ---- free functions (tools/utils.py:524-572, 617-659) on numpy arrays ---------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def tf_xywh_to_all(grid_pred_xy, grid_pred_wh, layer: int, h: Helper):
    """utils.py:524-547 (host numpy version for small inputs; the batched path is yk_decode_py)."""
    all_pred_xy = (_sigmoid(grid_pred_xy) + h.xy_offset[layer]) / h.out_hw[layer][::-1]
    all_pred_wh = np.exp(grid_pred_wh) * h.anchors[layer]
    return all_pred_xy, all_pred_wh


def tf_xywh_to_grid(all_true_xy, all_true_wh, layer: int, h: Helper):
    """utils.py:550-572."""
    grid_true_xy = (all_true_xy * h.out_hw[layer][::-1]) - h.xy_offset[layer]
    with np.errstate(divide='ignore'):
        grid_true_wh = np.log(all_true_wh / h.anchors[layer])
    return grid_true_xy, grid_true_wh


def tf_iou(pred_xy, pred_wh, vaild_xy, vaild_wh):
    """utils.py:617-659: [h,w,A,2] vs [n,2] -> [h,w,A,n]."""
    b1_xy = np.expand_dims(pred_xy, -2)
    b1_wh = np.expand_dims(pred_wh, -2)
    b1_mins, b1_maxes = b1_xy - b1_wh / 2., b1_xy + b1_wh / 2.
    b2_xy = np.expand_dims(vaild_xy, 0)
    b2_wh = np.expand_dims(vaild_wh, 0)
    b2_mins, b2_maxes = b2_xy - b2_wh / 2., b2_xy + b2_wh / 2.
    iw = np.maximum(np.minimum(b1_maxes, b2_maxes) - np.maximum(b1_mins, b2_mins), 0.)
    inter = iw[..., 0] * iw[..., 1]
    return inter / (b1_wh[..., 0] * b1_wh[..., 1] + b2_wh[..., 0] * b2_wh[..., 1] - inter)


# ---- loss / metrics (tools/utils.py:708-793, tools/custom.py:13-75): same names, GPU arithmetic ---------------
def create_loss_fn(h: Helper, obj_thresh: float, iou_thresh: float, obj_weight: float, noobj_weight: float,
                   wh_weight: float, layer: int):
    """tools/utils.py:708-793.  Returns loss_fn(y_true, y_pred) -> scalar loss (cuda tensor); the five terms, the
    gradient dL/dy_pred and the ignore mask of the last call are kept on the function object
    (`loss_fn.terms`, `loss_fn.grad`).  y_* are cuda fp32 tensors [B,h,w,A,5+C]; all arithmetic is in
    libyolo_hip.so (yk_yolo_loss)."""
    from . import engine

    def loss_fn(y_true, y_pred):
        loss, grad, _ = engine.yolo_loss(y_true, y_pred, h.anchors[layer], obj_thresh, iou_thresh, obj_weight,
                                         noobj_weight, wh_weight, batch_size=h.batch_size)
        loss_fn.terms, loss_fn.grad = loss, grad
        return loss[0]
    loss_fn.terms = loss_fn.grad = None
    return loss_fn


class _YoloMetric:
    """Running tp/fp/fn over batches, thresholding the RAW confidence logit exactly like custom.py:33."""

    def __init__(self, thresholds=None, name=None):
        self.thresholds = 0.5 if thresholds is None else thresholds
        self.name = name
        self._counts = None

    def reset_states(self):
        self._counts = None

    def update_state(self, y_true, y_pred, sample_weight=None, *, anchors_l=None):
        import torch
        from . import engine
        if self._counts is None:
            self._counts = torch.zeros(3, dtype=torch.float32, device=y_pred.device)
        A = y_pred.shape[3]
        anc = anchors_l if anchors_l is not None else np.ones((A, 2), np.float32)
        engine.yolo_loss(y_true, y_pred, anc, self.thresholds, 0.5, 1.0, 1.0, 1.0, counts=self._counts, want_grad=False)

    def _c(self):
        return [0.0, 0.0, 0.0] if self._counts is None else [float(v) for v in self._counts.cpu()]


class Yolo_Precision(_YoloMetric):
    """tools/custom.py:13-43: tp / (tp + fp), div_no_nan."""

    def result(self):
        tp, fp, _ = self._c()
        return tp / (tp + fp) if tp + fp > 0 else 0.0


class Yolo_Recall(_YoloMetric):
    """tools/custom.py:46-75: tp / (tp + fn), div_no_nan."""

    def result(self):
        tp, _, fn = self._c()
        return tp / (tp + fn) if tp + fn > 0 else 0.0
