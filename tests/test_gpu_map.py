"""BASELINE.json north star: "VOC mAP within 0.1 pt of the Keras reference".  No VOC data and no trained Keras checkpoint travel with
this repo, so the statement is tested where it can be: the reference's ONLY trained weights (yolo.kmodel, dequantised), 256 seeded
views of the K210 demo picture (the dog / bicycle / car scene: affine warps, flips, brightness and contrast changes), the fp32 oracle's
detections on every view as the ground truth.  mAP of the oracle against itself is 100 by construction; what each precision mode of
the HIP engine loses against it, in points, is the quantity the north star bounds.  The measured figures go to gpurun_out/map_eval.json
(copied to profiles/ by hand)."""
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
from oracle import decode_ref
from k210_yolo_framework_amd import kmodel, netspec as ns, voc_eval

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / 'golden'
ROOT = Path(__file__).resolve().parents[1]
N_VIEWS, OBJ, IOU_NMS = 256, 0.25, 0.5


def views(img: np.ndarray, n: int, seed: int = 7) -> np.ndarray:
    """n seeded views [n,224,320,3] u8 of one picture: scale 0.7-1.5, rotation +-12 degrees, shift, mirror, gain / offset (bilinear)."""
    rng = np.random.default_rng(seed)
    H, W = img.shape[:2]
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((n, H, W, 3), np.uint8)
    src = img.astype(np.float64)
    for k in range(n):
        s, th = rng.uniform(0.7, 1.5), np.deg2rad(rng.uniform(-12, 12))
        tx, ty = rng.uniform(-0.15, 0.15) * W, rng.uniform(-0.15, 0.15) * H
        flip = rng.random() < 0.5
        c, sn = np.cos(th) / s, np.sin(th) / s
        u = c * (xx - W / 2) - sn * (yy - H / 2) + W / 2 - tx
        v = sn * (xx - W / 2) + c * (yy - H / 2) + H / 2 - ty
        if flip:
            u = W - 1 - u
        u0, v0 = np.floor(u).astype(int), np.floor(v).astype(int)
        fu, fv = (u - u0)[..., None], (v - v0)[..., None]
        ok = ((u0 >= 0) & (u0 < W - 1) & (v0 >= 0) & (v0 < H - 1))[..., None]
        u0, v0 = np.clip(u0, 0, W - 2), np.clip(v0, 0, H - 2)
        val = (src[v0, u0] * (1 - fu) * (1 - fv) + src[v0, u0 + 1] * fu * (1 - fv) + src[v0 + 1, u0] * (1 - fu) * fv + src[v0 + 1, u0 + 1] * fu * fv)
        val = np.where(ok, val, 127.0) * rng.uniform(0.7, 1.25) + rng.uniform(-25, 25)
        out[k] = np.clip(np.rint(val), 0, 255).astype(np.uint8)
    out[0] = img                                                                    # the picture itself
    return out


def test_map_of_both_precision_modes_against_the_fp32_oracle():
    import torch
    from k210_yolo_framework_amd import engine
    gold = np.load(GOLD / 'kmodel_dog_golden.npz')
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w, _ = kmodel.to_float_weights(kmodel.parse((GOLD / 'yolo.kmodel').read_bytes()))
    anchors = gold['anchors'].reshape(2, 3, 2).astype(np.float32)
    frames = views(gold['image'].transpose(1, 2, 0).copy(), N_VIEWS)
    B = 32
    plan = spec.compile_plan(w)
    oracle.set_threads(32)
    truth = []
    for k in range(0, N_VIEWS, B):
        ref = oracle.net_forward(plan, oracle.normalise_u8(frames[k:k + B]), False, spec.outputs)
        rd = decode_ref.decode_batch([r.reshape(B, r.shape[1], r.shape[2], 3, 25) for r in ref], anchors, (224, 320), (224, 320), OBJ, IOU_NMS)
        truth += [x[0] for x in rd]
    n_truth = sum(len(t) for t in truth)
    assert n_truth >= N_VIEWS, n_truth                                              # the scene is detected in (nearly) every view
    assert voc_eval.evaluate(truth, truth, 20)['map'] == 1.0
    report = {'views': N_VIEWS, 'obj_thresh': OBJ, 'nms_iou': IOU_NMS, 'truth_detections': n_truth,
              'weights': 'yolo.kmodel (kfpkg/kpu_yolov3.kfpkg), dequantised', 'ground_truth': 'fp32 oracle detections on the same frames'}
    for prec in ('f16x2', 'f16'):
        pipe = engine.Pipeline(spec, w, anchors, max_batch=B, depth=2, precision=prec)
        got = []
        for k in range(0, N_VIEWS, B):
            rows, off = pipe.submit_host(frames[k:k + B], obj_thresh=OBJ, iou_thresh=IOU_NMS).result()
            got += voc_eval.split_rows(rows, off)
        pipe.close()
        r = voc_eval.evaluate(got, truth, 20)
        r07 = voc_eval.evaluate(got, truth, 20, use_07_metric=True)
        report[prec] = {'map': r['map'], 'map_voc07': r07['map'], 'delta_points': 100 * (r['map'] - 1.0), 'delta_points_voc07': 100 * (r07['map'] - 1.0),
                        'detections': int(sum(len(g) for g in got)), 'tp': int(r['tp'].sum()), 'fp': int(r['fp'].sum())}
    out = ROOT / 'gpurun_out'
    out.mkdir(exist_ok=True)
    (out / 'map_eval.json').write_text(json.dumps(report, indent=1))
    assert abs(report['f16x2']['delta_points']) <= 0.1, report['f16x2']            # the north-star bound, in the conforming mode
    assert abs(report['f16x2']['delta_points_voc07']) <= 0.1, report['f16x2']
    assert report['f16']['delta_points'] >= -3.0, report['f16']                     # the fp16-storage mode: recorded, loosely bounded
