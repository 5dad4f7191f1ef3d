#!/usr/bin/env python
"""A mAP check that can tell the precision modes apart (BASELINE north_star "VOC mAP within 0.1 pt of the Keras reference").

No VOC images and no trained checkpoint ship with the reference tree (`.MISSING_LARGE_BLOBS`), so: train yolo_mobilev1-0.75 on generated
images whose boxes are KNOWN (training.synthetic_list: coloured rectangles on noise, class = colour), then score the detections of
  (a) the fp32 oracle (oracle/yolo_net_ref.c + decode_ref),  (b) the engine in f16x2,  (c) the engine in f16
against the GENERATED ground truth with voc_eval.py (devkit definition), on images the training never saw.

    python tools/r05_map_synth.py [--steps 2500] [--train 2048] [--eval 1024] [--out profiles/r05_map_eval.json]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
import oracle  # noqa: E402
from oracle import decode_ref  # noqa: E402
from k210_yolo_framework_amd import netspec, training, voc_eval  # noqa: E402
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS  # noqa: E402
from k210_yolo_framework_amd.inference import detect  # noqa: E402
from k210_yolo_framework_amd.pipeline import InputPipeline  # noqa: E402
from k210_yolo_framework_amd.train import Trainer  # noqa: E402
from k210_yolo_framework_amd.yolonet import MODEL_DEFS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=2500)
ap.add_argument('--train', type=int, default=2048)
ap.add_argument('--eval', type=int, default=1024)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--lr', type=float, default=1e-3)
ap.add_argument('--out', default='profiles/r05_map_eval.json')
a = ap.parse_args()

IN_HW, CAM_HW, C = (224, 320), (240, 320), 20
h = Helper(None, C, VOC_ANCHORS, np.array([IN_HW]), np.array([[7, 10], [14, 20]]))
h.batch_size = a.batch
train_items = training.synthetic_list(a.train, CAM_HW, C, seed=1)
eval_items = training.synthetic_list(a.eval, CAM_HW, C, seed=99)
spec = netspec.yolo_mobilev1((*IN_HW, 3), 3, C, alpha=0.75)
tr = Trainer(spec, spec.init_keras_default(6), h.anchors, a.batch, obj_thresh=0.7, iou_thresh=0.3, obj_weight=5.0, noobj_weight=0.5, wh_weight=0.5,
             lr=a.lr, decay=0.0)                                    # the Makefile's training defaults
t0, steps, losses, epoch = time.time(), 0, [], 0
while steps < a.steps:
    pipe = InputPipeline(h, train_items, a.batch, 0, 1, seed=6, epoch=epoch, shuffle=True, device=0)
    try:
        for x, ys in pipe:
            out = tr.step(x, ys)
            steps += 1
            if steps % 100 == 0 or steps == 1:
                losses.append((steps, round(out['loss'], 3)))
                print(f'step {steps}: loss {out["loss"]:.3f}', flush=True)
            if steps >= a.steps:
                break
    finally:
        pipe.close()
    epoch += 1
train_s = time.time() - t0
weights = tr.export_weights()
del tr
torch.cuda.empty_cache()


def gt_rows(boxes, hw):
    b = np.asarray(boxes, np.float64).reshape(-1, 5)
    ih, iw = hw
    cx, cy, w, hh = b[:, 1] * iw, b[:, 2] * ih, b[:, 3] * iw, b[:, 4] * ih
    return np.stack([cy - hh / 2, cx - w / 2, cy + hh / 2, cx + w / 2, np.ones(len(b)), b[:, 0]], 1)


imgs = [it[0] for it in eval_items]
gts = [gt_rows(it[1], CAM_HW) for it in eval_items]
res = {}
for obj in (0.05, 0.7):
    dets = {}
    # (a) fp32 oracle: the host letterbox of the product's Helper (== tools/utils.py:378-399), the C conv stack, decode_ref
    x = np.stack([h._process_img(im, None, is_training=False, is_resize=True)[0] for im in imgs]).astype(np.float32)
    plan = spec.compile_plan(weights)
    outs = [oracle.net_forward(plan, x[k:k + 64], emulate_f16=False, out_ids=spec.outputs) for k in range(0, len(x), 64)]
    preds = [np.concatenate([o[l] for o in outs]) for l in range(len(spec.outputs))]
    rd = decode_ref.decode_batch_fast([p.reshape(len(x), p.shape[1], p.shape[2], 3, 5 + C) for p in preds], VOC_ANCHORS, IN_HW, CAM_HW, obj, 0.5, threads=16)
    dets['fp32_oracle'] = [r[0] for r in rd]
    # (b), (c) the product path
    for prec in ('f16x2', 'f16'):
        model, _ = MODEL_DEFS['yolo_mobilev1']([*IN_HW, 3], 3, C, alpha=0.75, precision=prec)
        model.set_weights(weights)
        d = []
        for k in range(0, len(imgs), 32):
            d += detect(h, model, imgs[k:k + 32], obj, 0.5)
        dets[prec] = d
    row = {}
    for k, d in dets.items():
        r = voc_eval.evaluate(d, gts, C, 0.5)
        row[k] = {'mAP_percent': round(100 * float(r['map']), 4), 'detections': int(sum(len(x_) for x_ in d)), 'tp': int(np.sum(r['tp'])), 'fp': int(np.sum(r['fp']))}
    # how the engines' detection LISTS differ from the oracle's (class + nearest box pairing is what voc_eval does; here: counts per image)
    for prec in ('f16x2', 'f16'):
        row[prec]['images_with_a_different_detection_count'] = int(sum(len(x_) != len(y_) for x_, y_ in zip(dets[prec], dets['fp32_oracle'])))
        row[prec]['delta_points_vs_fp32_oracle'] = round(row[prec]['mAP_percent'] - row['fp32_oracle']['mAP_percent'], 4)
    res[f'obj_thresh_{obj}'] = row
    print(obj, json.dumps(row), flush=True)
out = {'what': 'yolo_mobilev1-0.75 trained here on generated images with known boxes (training.synthetic_list), evaluated on unseen generated images with '
               'voc_eval.py (devkit AP, IoU 0.5, NMS IoU 0.5) against the GENERATED ground truth',
       'train': {'images': a.train, 'steps': steps, 'batch': a.batch, 'lr': a.lr, 'seconds': round(train_s, 1), 'loss_curve': losses},
       'eval_images': a.eval, 'results': res}
Path(ROOT / a.out).write_text(json.dumps(out, indent=1))
print('wrote', a.out)
