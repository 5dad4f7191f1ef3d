"""GPU half of SURVEY.md 8(f) N4: the reference's only trained weights (yolo.kmodel, dequantised by kmodel.py) through the HIP engine on
the K210 demo picture.  The conv stack is compared with the fp32 CPU oracle on the same float weights; the detections come from the
drop-in C region layer (main.c's thresholds) and must be the dog, the bicycle and the car of asset/dog_res.jpg / README.md:121-128."""
from pathlib import Path

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import kmodel, netspec as ns

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / 'golden'
VOC = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse', 'motorbike',
       'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


@pytest.mark.parametrize('precision,tol', [('f16x2', 1e-4), ('f16', 3e-2)])
def test_k210_demo_weights_on_the_demo_picture(precision, tol):
    from k210_yolo_framework_amd import engine, yolonet
    gold = np.load(GOLD / 'kmodel_dog_golden.npz')
    infer, _ = yolonet.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75, precision=precision)
    infer.load_weights(str(GOLD / 'yolo.kmodel'))
    rep = infer.last_load_report
    assert abs(rep['layers']['conv_pw_1']['alpha'] - 0.3) < 2e-3
    frame = gold['image'].transpose(1, 2, 0)[None].copy()                     # [1,224,320,3] u8, max 255
    outs = infer.predict(frame)
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w, _ = kmodel.to_float_weights(kmodel.parse((GOLD / 'yolo.kmodel').read_bytes()))
    ref = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frame), False, spec.outputs)
    for g, r in zip(outs, ref):
        assert np.isfinite(g).all()
        assert np.abs(g - r).max() <= tol * np.abs(r).max(), (precision, np.abs(g - r).max(), np.abs(r).max())
    hip = engine.lib()
    found = {}
    for li, (W, H) in enumerate([(10, 7), (20, 14)]):
        x = outs[li][0].transpose(2, 0, 1).reshape(3, 25, H, W).astype(np.float32).copy()
        res = oracle.drive_region_abi(hip, x, gold['anchors'][li], W, H, 3, 20, 0.5, 0.3)      # region_layer_init / run / draw_boxes
        for r in np.asarray(res[3]).reshape(-1, 6):
            found[VOC[int(r[4])]] = [int(v) for v in r[:4]]
    assert set(found) == {'dog', 'bicycle', 'car'}, found
    # same boxes as the CPU oracle's region layer on the fp32 logits (within a pixel for the fp16-storage mode)
    want = {}
    for li, (W, H) in enumerate([(10, 7), (20, 14)]):
        xx = ref[li][0].transpose(2, 0, 1).reshape(3, 25, H, W).astype(np.float32).copy()
        _, bx, pr = oracle.region_run(xx, gold['anchors'][li], W, H, 3, 20, 0.5, 0.3)
        for r in oracle.region_draw(bx, pr, 0.5).reshape(-1, 6):
            want[VOC[int(r[4])]] = [int(v) for v in r[:4]]
    for k in want:
        assert np.abs(np.array(found[k]) - np.array(want[k])).max() <= (0 if precision == 'f16x2' else 2), (k, found[k], want[k])
