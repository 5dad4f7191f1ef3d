"""`make inference` — the reference's keras_inference.py:75-176 on the HIP engine.

Same positional arguments and flags (keras_inference.py:180-190), same stdout table
(`[top left bottom right score class]`, :146,154).  Pipeline:
  Helper._read_img -> Helper._process_img (letterbox on the host; u8 frame)      tools/utils.py:339-406
  -> yk_run_u8 (normalise + conv stack, GPU) -> yk_decode_py (decode + per-class NMS, GPU)
  -> print table, draw boxes with PIL and save next to the input (the reference calls pil_img.show()).
"""
from __future__ import annotations

import argparse
import sys
from pathlib import Path

import numpy as np

from .helper import INFO, NOTE, Helper, VOC_ANCHORS
from .yolonet import MODEL_DEFS


def detect(h: Helper, model, orig_imgs, obj_thresh: float, iou_thresh: float):
    """orig_imgs: list of HxWx3 uint8 arrays -> list of [K,6] (top,left,bottom,right,score,class) numpy."""
    import torch
    from . import engine
    shapes = [img.shape[:2] for img in orig_imgs]
    n = len(orig_imgs)
    in_hw = tuple(int(v) for v in h.in_hw[0])
    if len(set(shapes)) == 1:
        src = torch.from_numpy(np.ascontiguousarray(np.stack(orig_imgs), np.uint8)).cuda()
        frames = engine.letterbox_u8(src, in_hw)                       # yk_letterbox_u8 (tools/utils.py:378-399)
    else:
        frames = torch.cat([engine.letterbox_u8(torch.from_numpy(np.ascontiguousarray(im[None], np.uint8)).cuda(), in_hw)
                            for im in orig_imgs])
    plan = model._plan(n)
    plan.run_u8(frames.contiguous())
    cfg = engine.make_decode_cfg(h.anchors, h.class_num, h.in_hw[0], h.out_hw)
    dets, counts = engine.decode_py(cfg, plan.outputs(), n, np.asarray(shapes, np.float32), obj_thresh, iou_thresh)
    torch.cuda.synchronize()
    dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
    return [dets[i, :counts[i]] for i in range(n)]


def main(ckpt_weights, image_size, output_size, model_def, class_num, depth_multiplier, obj_thresh, iou_thresh,
         train_set, test_image, anchors=None, precision='f16x2'):
    anchor_file = Path(f'data/{train_set}_anchor.npy')
    anc = str(anchor_file) if anchor_file.exists() else (anchors if anchors is not None else VOC_ANCHORS)
    h = Helper(None, class_num, anc, np.reshape(np.array(image_size), (-1, 2)), np.reshape(np.array(output_size), (-1, 2)))
    network = MODEL_DEFS[model_def]
    yolo_model, yolo_model_warpper = network([image_size[0], image_size[1], 3], len(h.anchors[0]), class_num,
                                             alpha=depth_multiplier, precision=precision)
    if ckpt_weights and str(ckpt_weights) not in ('None', '""', ''):
        yolo_model_warpper.load_weights(str(ckpt_weights))
        print(INFO, f' Load CKPT {str(ckpt_weights)}')
    else:
        print(NOTE, ' no checkpoint given: seeded random weights')
    orig_img = h._read_img(str(test_image))
    dets = detect(h, yolo_model, [orig_img], obj_thresh, iou_thresh)[0]
    if len(dets) > 0:
        print('[top\tleft\tbottom\tright\tscore\tclass]')
        for top, left, bottom, right, score, c in dets:
            print(f'[{top:.1f}\t{left:.1f}\t{bottom:.1f}\t{right:.1f}\t{score:.2f}\t{int(c):2d}]')
        try:
            from PIL import Image, ImageDraw
            pil_img = Image.fromarray(orig_img)
            draw = ImageDraw.Draw(pil_img)
            ih, iw = orig_img.shape[:2]
            thickness = max(1, (ih + iw) // 300)
            for top, left, bottom, right, score, c in dets:
                t, l = max(0, int(np.floor(top + 0.5))), max(0, int(np.floor(left + 0.5)))
                b, r = min(ih, int(np.floor(bottom + 0.5))), min(iw, int(np.floor(right + 0.5)))
                if b <= t or r <= l:
                    continue
                for j in range(thickness):
                    if r - j > l + j and b - j > t + j:
                        draw.rectangle([l + j, t + j, r - j, b - j], outline=h.colormap[int(c) % len(h.colormap)])
                draw.text((l, t + 1), '{:2d} {:.2f}'.format(int(c), score), fill=(0, 0, 0))
            out = Path(str(test_image)).with_suffix('').as_posix() + '_res.jpg'
            pil_img.save(out)
            print(INFO, f' saved {out}')
        except Exception as e:  # drawing is boundary glue, never fatal
            print(NOTE, f' drawing skipped: {e}')
    else:
        print(NOTE, ' no boxes detected')
    return dets


def cli(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--train_set', type=str, help='trian file lists', default='voc')
    parser.add_argument('--class_num', type=int, help='trian class num', default=20)
    parser.add_argument('--model_def', type=str, help='Model definition.', default='yolo_mobilev2')
    parser.add_argument('--depth_multiplier', type=float, help='mobilenet depth_multiplier', choices=[0.5, 0.75, 1.0], default=1.0)
    parser.add_argument('--image_size', type=int, help='net work input image size', default=(224, 320), nargs='+')
    parser.add_argument('--output_size', type=int, help='net work output image size', default=(7, 10, 14, 20), nargs='+')
    parser.add_argument('--obj_thresh', type=float, help='obj mask thresh', default=0.7)
    parser.add_argument('--iou_thresh', type=float, help='iou mask thresh', default=0.3)
    parser.add_argument('--precision', type=str, choices=['f16', 'f16x2'], default='f16x2',
                        help="arithmetic of the conv stack (not in the reference): 'f16x2' = fp32-class results (default), 'f16' = fastest")
    parser.add_argument('pre_ckpt', type=str, help='pre-train weights path')
    parser.add_argument('test_image', type=str, help='test image path')
    args = parser.parse_args(sys.argv[1:] if argv is None else argv)
    return main(args.pre_ckpt, args.image_size, args.output_size, args.model_def, args.class_num, args.depth_multiplier,
                args.obj_thresh, args.iou_thresh, args.train_set, args.test_image, precision=args.precision)


if __name__ == '__main__':
    cli()
