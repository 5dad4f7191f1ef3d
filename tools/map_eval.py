#!/usr/bin/env python
"""VOC mAP of a checkpoint on an annotation list, through the HIP engine (BASELINE.json: "VOC mAP within 0.1 pt of the Keras reference").

    python tools/map_eval.py CKPT [--ann data/voc_img_ann.npy] [--model_def yolo_mobilev1 --depth_multiplier 0.75]
                             [--precision f16x2] [--compare f16] [--obj_thresh 0.05] [--iou_thresh 0.5] [--limit N] [--voc07]

`--ann`: the list `make_voc_list.py` writes (rows [image path, boxes [n,5] = (class, cx, cy, w, h) relative to the image, ...]); the
validation head of it (Helper's validation_split) is evaluated unless --all.  The reference has no evaluator: the metric is the VOC
devkit's (k210_yolo_framework_amd/voc_eval.py).  With --compare the second precision mode is evaluated on the same images and the mAP
difference is printed in points.
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from k210_yolo_framework_amd import voc_eval                         # noqa: E402
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS      # noqa: E402
from k210_yolo_framework_amd.inference import detect                 # noqa: E402
from k210_yolo_framework_amd.yolonet import MODEL_DEFS               # noqa: E402


def ground_truth_rows(boxes: np.ndarray, img_hw) -> np.ndarray:
    """[n,5] (class, cx, cy, w, h) relative -> [n,6] (top, left, bottom, right, 1, class) in pixels of the original image."""
    boxes = np.asarray(boxes, np.float64).reshape(-1, 5)
    ih, iw = float(img_hw[0]), float(img_hw[1])
    cx, cy, w, h = boxes[:, 1] * iw, boxes[:, 2] * ih, boxes[:, 3] * iw, boxes[:, 4] * ih
    return np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2, np.ones(len(boxes)), boxes[:, 0]], 1)


def run(model, h: Helper, rows, obj_thresh, iou_nms, batch=32):
    # NOTE: detect() keeps at most 30 detections per class and image (keras_inference.py:125 max_output_size=30): at a low obj_thresh the
    # low-score tail of a crowded image is cut, which lowers recall - and mAP - slightly against an uncapped evaluator.  Printed with the result.
    dets, gts = [], []
    for k in range(0, len(rows), batch):
        imgs = [h._read_img(str(r[0])) for r in rows[k:k + batch]]
        dets += detect(h, model, imgs, obj_thresh, iou_nms)
        gts += [ground_truth_rows(r[1], im.shape[:2]) for r, im in zip(rows[k:k + batch], imgs)]
    return dets, gts


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('ckpt')
    p.add_argument('--ann', default='data/voc_img_ann.npy')
    p.add_argument('--train_set', default='voc')
    p.add_argument('--class_num', type=int, default=20)
    p.add_argument('--model_def', default='yolo_mobilev1')
    p.add_argument('--depth_multiplier', type=float, default=0.75)
    p.add_argument('--image_size', type=int, default=(224, 320), nargs='+')
    p.add_argument('--output_size', type=int, default=(7, 10, 14, 20), nargs='+')
    p.add_argument('--precision', choices=['f16', 'f16x2'], default='f16x2')
    p.add_argument('--compare', choices=['f16', 'f16x2'], default=None)
    p.add_argument('--obj_thresh', type=float, default=0.05)
    p.add_argument('--nms_iou', type=float, default=0.5, help='IoU of the per-class NMS (keras_inference.py --iou_thresh)')
    p.add_argument('--iou_thresh', type=float, default=0.5, help='IoU a detection needs with a ground-truth box')
    p.add_argument('--voc07', action='store_true', help='11-point AP')
    p.add_argument('--all', action='store_true', help='evaluate the whole list, not its validation head')
    p.add_argument('--limit', type=int, default=0)
    a = p.parse_args(sys.argv[1:] if argv is None else argv)
    anchor_file = Path(f'data/{a.train_set}_anchor.npy')
    h = Helper(a.ann, a.class_num, str(anchor_file) if anchor_file.exists() else VOC_ANCHORS, np.reshape(np.array(a.image_size), (-1, 2)),
               np.reshape(np.array(a.output_size), (-1, 2)))
    rows = list(h.train_list) + list(h.test_list) if a.all else list(h.test_list)
    if a.limit:
        rows = rows[:a.limit]
    res = {}
    for prec in [a.precision] + ([a.compare] if a.compare and a.compare != a.precision else []):
        model, _ = MODEL_DEFS[a.model_def]([a.image_size[0], a.image_size[1], 3], len(h.anchors[0]), a.class_num, alpha=a.depth_multiplier,
                                           precision=prec)
        model.load_weights(a.ckpt)
        dets, gts = run(model, h, rows, a.obj_thresh, a.nms_iou)
        r = voc_eval.evaluate(dets, gts, a.class_num, a.iou_thresh, a.voc07)
        res[prec] = r
        print(f'{prec}: mAP {100 * r["map"]:.2f} over {len(rows)} images ({"VOC07 11-point" if a.voc07 else "area"} AP, IoU {a.iou_thresh}; '
              f'obj_thresh {a.obj_thresh}, at most 30 detections per class and image as keras_inference.py:125)')
        for c in range(a.class_num):
            if r['n_gt'][c]:
                print(f'   class {c:2d}: AP {100 * r["ap"][c]:6.2f}   gt {r["n_gt"][c]:5d}  det {r["n_det"][c]:6d}  tp {r["tp"][c]:5d}')
    if len(res) == 2:
        k = list(res)
        print(f'mAP({k[1]}) - mAP({k[0]}) = {100 * (res[k[1]]["map"] - res[k[0]]["map"]):+.3f} points')
    return res


if __name__ == '__main__':
    main()
