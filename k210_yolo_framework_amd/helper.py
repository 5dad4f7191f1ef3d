"""Host-side mirror of the reference's `tools/utils.py` API surface for the hot path.

Same names, argument meaning and array shapes as the reference (`Helper`, `tf_xywh_to_all`, `tf_xywh_to_grid`, `tf_iou`,
`calc_ignore_mask`, `create_loss_fn`), with numpy / torch tensors instead of `tf.Tensor`.  The members are written for this
code base (vectorised numpy, tables built with broadcasting); each docstring names the reference lines whose BEHAVIOUR it
reproduces, and tests/test_helper.py pins every one of them with hand-derived known answers.  The heavy arithmetic (model
forward, decode, NMS, loss) is NOT here — it runs in libyolo_hip.so.

Out of scope: imgaug augmentation, matplotlib drawing.
"""
from __future__ import annotations

import os
from typing import Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

INFO, ERROR, NOTE = '[ INFO  ]', '[ ERROR ]', '[ NOTE  ]'     # tools/utils.py:15-17 (uncoloured)

# data/voc_anchor.npy of the reference == main.c:46-52 (w,h) relative to the whole image
VOC_ANCHORS = np.array([[[0.76120044, 0.57155991], [0.6923348, 0.88535553], [0.47163042, 0.34163313]],
                        [[0.33340788, 0.70065861], [0.18124964, 0.38986752], [0.08497349, 0.1527057]]], np.float64)

# tools/utils.py:89-105 (data table): 20 VOC colours + 20 more
_COLORMAP = [(255, 82, 0), (0, 255, 245), (0, 61, 255), (0, 255, 112), (0, 255, 133), (255, 0, 0), (255, 163, 0), (255, 102, 0),
             (194, 255, 0), (0, 143, 255), (51, 255, 0), (0, 82, 255), (0, 255, 41), (0, 255, 173), (10, 0, 255), (173, 255, 0),
             (0, 255, 153), (255, 92, 0), (255, 0, 255), (255, 0, 245), (128, 0, 0), (0, 128, 0), (128, 128, 0), (0, 0, 128),
             (128, 0, 128), (0, 128, 128), (128, 128, 128), (64, 0, 0), (192, 0, 0), (64, 128, 0), (192, 128, 0), (64, 0, 128),
             (192, 0, 128), (64, 128, 128), (192, 128, 128), (0, 64, 0), (128, 64, 0), (0, 192, 0), (128, 192, 0), (0, 64, 128),
             # entries 40-79 of the reference's constant table (tools/utils.py:97-104): the 80-class (COCO) case of keras_inference.py:157
             (61, 230, 250), (255, 6, 51), (11, 102, 255), (255, 7, 71), (255, 9, 224), (9, 7, 230), (220, 220, 220), (255, 9, 92),
             (112, 9, 255), (8, 255, 214), (7, 255, 224), (255, 184, 6), (10, 255, 71), (255, 41, 10), (7, 255, 255), (224, 255, 8),
             (102, 8, 255), (255, 61, 6), (255, 194, 7), (255, 122, 8), (0, 255, 20), (255, 8, 41), (255, 5, 153), (6, 51, 255),
             (235, 12, 255), (160, 150, 20), (0, 163, 255), (140, 140, 140), (250, 10, 15), (20, 255, 0), (31, 255, 0), (255, 31, 0),
             (255, 224, 0), (153, 255, 0), (0, 0, 255), (255, 71, 0), (0, 235, 255), (0, 173, 255), (31, 0, 255), (11, 200, 200)]


def write_arguments_to_file(args, filename) -> None:
    """The `args.txt` dump next to every checkpoint (tools/utils.py:47-50, keras_train.py:23-26,41): one `key: value` line per
    parsed argument, in declaration order."""
    lines = [f'{key}: {value}\n' for key, value in vars(args).items()]
    with open(filename, 'w') as f:
        f.writelines(lines)


class Helper(object):
    """tools/utils.py:53-521: anchors, grid tables, label encode/decode, image reading and letterboxing, dataset iteration."""

    def __init__(self, image_ann: Optional[str], class_num: int, anchors: Union[str, np.ndarray, None],
                 in_hw: Sequence, out_hw: Sequence, validation_split: float = 0.1):
        self.in_hw, self.out_hw = np.asarray(in_hw), np.asarray(out_hw)
        if self.in_hw.ndim != 2 or self.out_hw.ndim != 2:
            raise AssertionError('in_hw / out_hw must be [[h, w], ...]')
        self.validation_split = validation_split
        self.train_list = self.test_list = None
        if image_ann is not None:                        # [path, boxes, ...] rows; the head of the list is the validation split
            rows = np.load(image_ann, allow_pickle=True)
            n_val = int(len(rows) * validation_split)
            self.test_list, self.train_list = rows[:n_val], rows[n_val:]
            self.train_total_data, self.test_total_data = len(self.train_list), len(self.test_list)
        self.grid_wh = 1.0 / self.out_hw[:, ::-1]        # size of one cell, (w, h) per layer           utils.py:70
        if class_num:
            self.class_num = class_num
        if anchors is not None:
            self.anchors = np.load(anchors) if isinstance(anchors, str) else np.asarray(anchors)
            self.output_number, self.anchor_number = len(self.anchors), len(self.anchors[0])
            self.xy_offset = Helper._coordinate_offset(self.anchors, self.out_hw)
            self.wh_scale = Helper._anchor_scale(self.anchors, self.grid_wh)
        self.output_shapes = [[None, int(hw[0]), int(hw[1]), len(a), self.class_num + 5] for hw, a in zip(self.out_hw, self.anchors)]
        self.batch_size = None
        self.colormap = list(_COLORMAP)

    # ---- label side (behaviour of tools/utils.py:140-307) -----------------------------------------
    def _xy_grid_index(self, box_xy: np.ndarray, layer: int):
        """Cell (column, row) that holds a centre given relative to the image (utils.py:156)."""
        h, w = self.out_hw[layer]
        return np.floor(np.asarray(box_xy) * (w, h)).astype(int)

    def _xy_to_grid(self, xy: np.ndarray, layer: int) -> np.ndarray:
        """Image-relative label centres [out_h, out_w, A, 2] -> position inside their own cell, in cell units (utils.py:107-122): the
        inverse of `_xy_to_all`."""
        cells_wh = self.out_hw[layer][::-1]
        return np.asarray(xy) * cells_wh - self.xy_offset[layer]

    @staticmethod
    def _fake_iou(a: np.ndarray, b: np.ndarray):
        """IoU of boxes (w,h) that share their centre (utils.py:159-188): the overlap is min(w)*min(h)."""
        a, b = np.asarray(a, float), np.asarray(b, float)
        overlap = np.clip(np.minimum(a, b), 0.0, None).prod(axis=-1)
        return overlap / (a.prod(axis=-1) + b.prod(axis=-1) - overlap)

    def _get_anchor_index(self, wh: np.ndarray):
        """(layer, anchor) whose shape fits `wh` best over ALL layers (utils.py:190-205)."""
        table = Helper._fake_iou(wh, self.anchors)
        return np.unravel_index(int(table.argmax()), table.shape)

    def box_to_label(self, true_box: np.ndarray, out: Optional[List[np.ndarray]] = None) -> List[np.ndarray]:
        """[n,5] = [cls,x,y,w,h] (image-relative) -> one [h,w,A,5+C] float32 grid per layer (utils.py:207-230).
        A box lands in the cell of its centre, at the best-fitting anchor: xywh clipped to [1e-8, 1], conf 1, one-hot class.
        Boxes are written in order, so a later box in the same slot replaces xywh and ADDS its class bit, like the reference."""
        if out is None:
            grids = [np.zeros((*map(int, self.out_hw[l]), len(self.anchors[l]), 5 + self.class_num), np.float32)
                     for l in range(self.output_number)]
        else:                                              # caller's buffers (the input pipeline's pinned staging rows): no copy later
            grids = out
            for g in grids:
                g.fill(0.0)
        boxes = np.asarray(true_box, float).reshape(-1, 5)
        if len(boxes):
            fit = Helper._fake_iou(boxes[:, None, None, 3:5], self.anchors[None])            # [n, L, A]
            layer, anchor = np.unravel_index(fit.reshape(len(boxes), -1).argmax(1), fit.shape[1:])
            for k, (l, a) in enumerate(zip(layer, anchor)):
                cx, cy = self._xy_grid_index(boxes[k, 1:3], l)
                slot = grids[l][cy, cx, a]
                slot[:4] = np.clip(boxes[k, 1:5], 1e-8, 1.0)
                slot[4] = 1.0
                slot[5 + int(boxes[k, 0])] = 1.0
        return grids

    def batch_box_to_label(self, boxes_per_sample: Sequence[np.ndarray]) -> List[np.ndarray]:
        """`box_to_label` of every sample of a batch in a handful of array operations -> one [n,h,w,A,5+C] float32 array per layer,
        bit-identical to stacking the per-sample results (boxes are scattered in order: a later box in the same slot replaces xywh
        and adds its class bit).  The input pipeline's host side is bound by the NUMBER of small numpy calls on a busy host, not by
        their work: ~10 calls per batch here instead of ~12 per box."""
        n = len(boxes_per_sample)
        grids = [np.zeros((n, *map(int, self.out_hw[l]), len(self.anchors[l]), 5 + self.class_num), np.float32)
                 for l in range(self.output_number)]
        per = [np.asarray(b, float).reshape(-1, 5) for b in boxes_per_sample]
        counts = [len(b) for b in per]
        if sum(counts) == 0:
            return grids
        allb = np.concatenate(per)
        sid = np.repeat(np.arange(n), counts)
        fit = Helper._fake_iou(allb[:, None, None, 3:5], self.anchors[None])                 # [N, L, A]
        layer, anchor = np.unravel_index(fit.reshape(len(allb), -1).argmax(1), fit.shape[1:])
        xywh = np.clip(allb[:, 1:5], 1e-8, 1.0)
        cls = allb[:, 0].astype(int)
        for l in range(self.output_number):
            m = np.nonzero(layer == l)[0]
            if not len(m):
                continue
            h, w = self.out_hw[l]
            cell = np.floor(allb[m, 1:3] * (w, h)).astype(int)
            s_, cy, cx, a = sid[m], cell[:, 1], cell[:, 0], anchor[m]
            grids[l][s_, cy, cx, a, :4] = xywh[m]          # repeated slots: numpy assigns in order, the last box wins like the loop
            grids[l][s_, cy, cx, a, 4] = 1.0
            grids[l][s_, cy, cx, a, 5 + cls[m]] = 1.0
        return grids

    @staticmethod
    def _coordinate_offset(anchors: np.ndarray, out_hw: np.ndarray) -> np.ndarray:
        """Per layer [h,w,1,2]: (column, row) of every cell (utils.py:233-253)."""
        tables = []
        for h, w in np.asarray(out_hw)[:len(anchors)]:
            rows, cols = np.indices((int(h), int(w)))
            tables.append(np.stack([cols, rows], -1)[:, :, None, :])
        if len({t.shape for t in tables}) > 1:
            out = np.empty(len(tables), dtype=object)
            out[:] = tables
            return out
        return np.stack(tables)

    @staticmethod
    def _anchor_scale(anchors: np.ndarray, grid_wh: np.ndarray) -> np.ndarray:
        """Anchors measured in cells^-1: anchor * cell size, per layer (utils.py:256-271)."""
        return np.stack([np.asarray(a) * g for a, g in zip(anchors, grid_wh)])

    def _xy_to_all(self, labels):
        """In place: cell-relative xy -> image-relative (utils.py:273-281)."""
        for lab, cell, off in zip(labels, self.grid_wh, self.xy_offset):
            lab[..., :2] *= cell
            lab[..., :2] += off

    def _wh_to_all(self, labels):
        """In place: log-space wh -> image-relative (utils.py:283-291)."""
        for lab, anc in zip(labels, self.anchors):
            lab[..., 2:4] = anc * np.exp(lab[..., 2:4])

    def label_to_box(self, labels, thersh=.7) -> np.ndarray:
        """Grids -> [k,5] = [cls,x,y,w,h] of the slots whose confidence exceeds `thersh` (utils.py:293-307)."""
        hot = np.concatenate([lab[lab[..., 4] > thersh] for lab in labels], axis=0)
        return np.column_stack([hot[:, 5:].argmax(axis=1), hot[:, :4]])

    # ---- image side (behaviour of tools/utils.py:339-406) -----------------------------------------
    def _read_img(self, img_path: str) -> np.ndarray:
        """RGB [H,W,3] like skimage.io.imread + gray2rgb / alpha drop (utils.py:339-355); pinned against the real skimage for every PIL
        mode by tests/golden/imread_golden.npz."""
        from PIL import Image
        im = Image.open(img_path)
        if im.mode.startswith('I;16'):                   # 16-bit gray stays uint16 (imread's PIL plugin keeps the sample depth)
            img = np.asarray(im).astype(np.uint16)
        else:
            if im.mode not in ('RGB', 'RGBA', 'L'):      # palette, LA, CMYK, bilevel ...: what imread's PIL plugin converts
                im = im.convert('RGBA' if 'A' in im.mode or 'transparency' in im.info else 'RGB')
            img = np.asarray(im)
        if img.ndim == 2:
            img = np.repeat(img[..., None], 3, axis=-1)
        return img[..., :3]

    def letterbox_params(self, img_hw) -> Tuple[np.ndarray, np.ndarray]:
        """utils.py:378-385: one scale = min(in_wh / img_wh) for both axes; translation = ((in_wh - img_wh*scale)/2) truncated."""
        img_wh = np.array([img_hw[1], img_hw[0]], float)
        in_wh = self.in_hw[0][::-1].astype(float)
        scale = np.full(2, (in_wh / img_wh).min())
        return scale, ((in_wh - img_wh * scale) / 2).astype(int)

    def _process_img(self, img: np.ndarray, true_box, is_training: bool, is_resize: bool):
        """utils.py:357-406 without augmentation: letterbox (bilinear, zero fill, truncating uint8 cast) then `img / np.max(img)`.
        The warp is skimage.transform.warp(order=1, mode='constant', cval=0, preserve_range=True) restated; pinned against the
        real skimage by tests/golden/letterbox_golden.npz."""
        if is_resize:
            scale, translation = self.letterbox_params(img.shape[:2])
            if isinstance(true_box, np.ndarray):
                # centre and size (fractions of the source image) -> pixels of the scaled image -> fractions of the network tensor;
                # only the centre is shifted by the letterbox margin (utils.py:386-389)
                src_wh, net_wh = np.tile(img.shape[1::-1], 2), np.tile(self.in_hw[0][::-1], 2)
                moved = true_box[:, 1:5] * src_wh * np.tile(scale, 2)
                moved[:, :2] += translation
                true_box[:, 1:5] = moved / net_wh
            img = letterbox_bilinear(img, tuple(self.in_hw[0]), float(scale[0]), translation)
        if is_training:
            raise NotImplementedError('imgaug augmentation is out of scope (SURVEY.md §2 #6)')
        img = img / np.max(img)
        return img, true_box

    # ---- dataset iteration (behaviour of tools/utils.py:408-450) ----------------------------------
    def generator(self, is_training=True, is_resize=True, is_make_lable=True, train_list=None):
        """utils.py:408-415: (image, labels | boxes) one sample at a time."""
        rows = self.train_list if train_list is None or train_list is True else train_list
        for row in rows:
            src, boxes = row[0], np.array(row[1], float, copy=True)
            img = self._read_img(str(src)) if isinstance(src, (str, os.PathLike)) else src
            img, boxes = self._process_img(img, boxes, is_training, is_resize)
            yield img, (self.box_to_label(boxes) if is_make_lable else boxes)

    def _create_dataset(self, image_ann_list, batch_size: int, rand_seed: int, is_training: bool, is_resize: bool,
                        repeat: bool = True) -> Iterator[Tuple[np.ndarray, List[np.ndarray]]]:
        """What the tf.data pipeline of utils.py:417-441 yields: (images [B,H,W,3] float32, one label tensor [B,h,w,A,5+C]
        per layer), reshuffled every pass, incomplete batches dropped (`batch(batch_size, True)`), repeating for ever."""
        print(INFO, 'data augment is ', str(is_training))
        rng = np.random.default_rng(rand_seed)
        rows = list(image_ann_list)
        if not rows:
            if repeat:                                     # the reference's infinite generator over an empty list never yields either;
                raise ValueError('empty image list: nothing to batch')   # say so instead of spinning
            return

        def batch_of(picked):
            samples = list(self.generator(is_training, is_resize, True, picked))
            imgs = np.stack([im.astype(np.float32) for im, _ in samples])
            labs = [np.stack([lab[l] for _, lab in samples]).astype(np.float32) for l in range(self.output_number)]
            return imgs, labs
        if repeat and len(rows) < batch_size:
            # fewer rows than one batch (a small validation split): the reference repeats BEFORE it batches (utils.py:438-441), so its
            # batches run across passes; a per-pass loop would never complete one
            pending: List = []
            while True:
                pending += [rows[i] for i in rng.permutation(len(rows))]
                while len(pending) >= batch_size:
                    yield batch_of(pending[:batch_size])
                    pending = pending[batch_size:]
        while True:
            order = rng.permutation(len(rows)) if is_training or repeat else np.arange(len(rows))
            for s in range(0, len(order) - batch_size + 1, batch_size):
                yield batch_of([rows[i] for i in order[s:s + batch_size]])
            if not repeat:
                return

    def set_dataset(self, batch_size, rand_seed, is_training=True, is_resize=True):
        """utils.py:443-450."""
        self.batch_size = batch_size
        for split, lst, train in (('train', self.train_list, is_training), ('test', self.test_list, False)):
            setattr(self, split + '_dataset', self._create_dataset(lst, batch_size, rand_seed, train, is_resize))
            setattr(self, split + '_epoch_step', getattr(self, split + '_total_data') // batch_size)

    def get_iter(self, is_training=True):
        """utils.py:452-456: the next batch of the chosen dataset."""
        return next(self.train_dataset if is_training else self.test_dataset)

    # ---- box format helpers (behaviour of tools/utils.py:492-521) ---------------------------------
    def _pixel_scale(self, on: bool) -> np.ndarray:
        return np.array([self.in_hw[0, 1], self.in_hw[0, 0]], float) if on else np.ones(2)

    def center_to_corner(self, true_box, to_all_scale=True):
        """[n, 4] x,y,w,h -> x1,y1,x2,y2 (in network-input pixels when to_all_scale)."""
        box = np.asarray(true_box, float)
        half = box[:, 2:4] / 2
        return np.hstack([box[:, 0:2] - half, box[:, 0:2] + half]) * np.tile(self._pixel_scale(to_all_scale), 2)

    def corner_to_center(self, xyxy_box, from_all_scale=True):
        """[n, 4] x1,y1,x2,y2 -> x,y,w,h (from network-input pixels when from_all_scale)."""
        box = np.asarray(xyxy_box, float)
        lo, hi = box[:, 0:2], box[:, 2:4]
        return np.hstack([(lo + hi) / 2, hi - lo]) / np.tile(self._pixel_scale(from_all_scale), 2)


def letterbox_bilinear(img: np.ndarray, out_hw: Tuple[int, int], scale: float, translation) -> np.ndarray:
    """skimage.transform.warp(img, AffineTransform(scale, translation).inverse, output_shape=out_hw, order=1, mode='constant',
    cval=0, preserve_range=True).astype('uint8') with skimage's own float64 arithmetic: source coordinate c = (1/s)*x + (-(tx*(1/s)))
    (the inverse matrix numpy.linalg.inv returns), corner pixels floor/ceil, rows blended after columns, zero outside, truncating
    cast.  Bit-identical to scikit-image on tests/golden/letterbox_golden.npz."""
    H, W = int(out_hw[0]), int(out_hw[1])
    ih, iw = img.shape[:2]
    inv = 1.0 / float(scale)
    c = inv * np.arange(W, dtype=np.float64) + (-(float(translation[0]) * inv))
    r = inv * np.arange(H, dtype=np.float64) + (-(float(translation[1]) * inv))
    c_lo, c_hi, r_lo, r_hi = np.floor(c), np.ceil(c), np.floor(r), np.ceil(r)
    dc, dr = (c - c_lo)[None, :, None], (r - r_lo)[:, None, None]
    src = img.astype(np.float64)

    def corner(rows, cols):
        rows, cols = rows.astype(int), cols.astype(int)
        live = ((rows >= 0) & (rows < ih))[:, None] & ((cols >= 0) & (cols < iw))[None, :]
        return np.where(live[..., None], src[rows.clip(0, ih - 1)[:, None], cols.clip(0, iw - 1)[None, :]], 0.0)
    top = (1 - dc) * corner(r_lo, c_lo) + dc * corner(r_lo, c_hi)
    bottom = (1 - dc) * corner(r_hi, c_lo) + dc * corner(r_hi, c_hi)
    return ((1 - dr) * top + dr * bottom).astype('uint8')


# ---- free functions (behaviour of tools/utils.py:524-572, 617-705) on numpy arrays ---------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def tf_xywh_to_all(grid_pred_xy, grid_pred_wh, layer: int, h: Helper):
    """utils.py:524-547 (host numpy version for small inputs; the batched path is yk_decode_py)."""
    cells_wh = h.out_hw[layer][::-1]
    return (_sigmoid(grid_pred_xy) + h.xy_offset[layer]) / cells_wh, np.exp(grid_pred_wh) * h.anchors[layer]


def tf_xywh_to_grid(all_true_xy, all_true_wh, layer: int, h: Helper):
    """utils.py:550-572."""
    cells_wh = h.out_hw[layer][::-1]
    with np.errstate(divide='ignore'):
        return all_true_xy * cells_wh - h.xy_offset[layer], np.log(all_true_wh / h.anchors[layer])


def tf_reshape_box(true_xy_A, true_wh_A, p_xy_A, p_wh_A, layer: int, helper: Helper):
    """utils.py:575-614: the broadcast shapes the reference's (older) ignore-mask code pairs every prediction with every true box in:
    true [n,2] -> [B,h,w,A,n,2] and predictions [B,h,w,A,2] -> [B,h,w,A,n,2].  Returned as numpy broadcast VIEWS (the reference
    materialises them with tf.tile); `helper.batch_size` must be set, as there."""
    t_xy, t_wh = np.asarray(true_xy_A), np.asarray(true_wh_A)
    q_xy, q_wh = np.asarray(p_xy_A), np.asarray(p_wh_A)
    n = t_xy.shape[0]
    gh, gw = (int(v) for v in helper.out_hw[layer])
    lead = (int(helper.batch_size), gh, gw, int(helper.anchor_number))
    if q_xy.shape[:4] != lead:
        raise ValueError(f'tf_reshape_box: predictions {q_xy.shape} do not match (batch, h, w, anchors) = {lead}')
    return (np.broadcast_to(t_xy, lead + (n, 2)), np.broadcast_to(t_wh, lead + (n, 2)),
            np.broadcast_to(q_xy[..., None, :], lead + (n, 2)), np.broadcast_to(q_wh[..., None, :], lead + (n, 2)))


def tf_iou(pred_xy, pred_wh, vaild_xy, vaild_wh):
    """utils.py:617-659: every predicted box [h,w,A,2] against every valid box [n,2] -> IoU [h,w,A,n]."""
    p_xy, p_wh = np.asarray(pred_xy)[..., None, :], np.asarray(pred_wh)[..., None, :]
    v_xy, v_wh = np.asarray(vaild_xy)[None], np.asarray(vaild_wh)[None]
    lo = np.maximum(p_xy - p_wh / 2., v_xy - v_wh / 2.)
    hi = np.minimum(p_xy + p_wh / 2., v_xy + v_wh / 2.)
    inter = np.clip(hi - lo, 0., None).prod(-1)
    return inter / (p_wh.prod(-1) + v_wh.prod(-1) - inter)


def calc_ignore_mask(t_xy_A, t_wh_A, p_xy, p_wh, obj_mask, iou_thresh: float, layer: int, helper: Helper):
    """utils.py:662-705: 1 where a prediction's best IoU with the image's own ground-truth boxes is below `iou_thresh`.
    t_xy_A / t_wh_A: image-scale truth [B,h,w,A,2]; p_xy / p_wh: raw predictions [B,h,w,A,2]; obj_mask [B,h,w,A] bool.
    -> cuda float32 [B,h,w,A,1].  The arithmetic is the mask output of yk_yolo_loss (libyolo_hip.so), the same code the
    training step uses; inputs may be numpy arrays or cuda tensors."""
    import torch
    from . import engine
    engine.require_gpu()

    def dev(x):
        return x.float().cuda() if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda()
    t_xy, t_wh, q_xy, q_wh = dev(t_xy_A), dev(t_wh_A), dev(p_xy), dev(p_wh)
    mask = obj_mask.cuda() if isinstance(obj_mask, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(obj_mask)).cuda()
    B, gh, gw, A, _ = q_xy.shape
    y_true = torch.zeros((B, gh, gw, A, 6), dtype=torch.float32, device=q_xy.device)
    y_pred = torch.zeros_like(y_true)
    y_true[..., 0:2], y_true[..., 2:4], y_true[..., 4] = t_xy, t_wh, mask.float()
    y_pred[..., 0:2], y_pred[..., 2:4] = q_xy, q_wh
    _, _, ign = engine.yolo_loss(y_true, y_pred, helper.anchors[layer], 0.5, iou_thresh, 1.0, 1.0, 1.0, batch_size=B,
                                 want_grad=False, want_ignore=True)
    return ign[..., None]


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
