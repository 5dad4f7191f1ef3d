"""Generator of the N4 fixtures (run in the build container, where /root/reference exists):

    python tests/golden/make_kmodel_golden.py

  tests/golden/yolo.kmodel            the `yolo.kmodel` member of /root/reference/yolo3_frame_test_public/kfpkg/kpu_yolov3.kfpkg, byte for byte
                                      (a DATA file of the reference: the trained 8-bit yolo_mobilev1-0.75 its K210 demo flashes, main.c:57,213,274)
  tests/golden/kmodel_dog_golden.npz  image   uint8 [3][224][320]: the numbers of yolo3_frame_test_public/aiimg.h (kfpkg/dog.jpg resized by
                                              kfpkg/mkaiimg.py - the exact bytes main.c:303 feeds the KPU)
                                      y1_q, y2_q   the two float outputs of oracle/kpu_ref.py (the KPU integer pipeline) on that image
                                      dets    [n][6] x1 y1 x2 y2 class prob-bits of the reference's C region layer at main.c's thresholds
                                              (0.6 / 0.3) on those outputs
"""
import re
import sys
import zipfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
REF = Path('/root/reference/yolo3_frame_test_public')
ANCHORS = [[0.76120044, 0.57155991, 0.6923348, 0.88535553, 0.47163042, 0.34163313],
           [0.33340788, 0.70065861, 0.18124964, 0.38986752, 0.08497349, 0.1527057]]          # main.c:46-52


def main():
    import oracle
    from k210_yolo_framework_amd import kmodel
    from oracle import kpu_ref
    with zipfile.ZipFile(REF / 'kfpkg' / 'kpu_yolov3.kfpkg') as z:
        data = z.read('yolo.kmodel')
    (ROOT / 'tests' / 'golden' / 'yolo.kmodel').write_bytes(data)
    txt = (REF / 'aiimg.h').read_text()
    img = np.array([int(v) for v in txt[txt.index('{') + 1: txt.rindex('}')].split(',')], np.uint8).reshape(3, 224, 320)
    outs = kpu_ref.run(kmodel.parse(data), img)
    dets = []
    for li, (W, H) in enumerate([(10, 7), (20, 14)]):
        x = outs[li].reshape(3, 25, H, W).astype(np.float32)
        run = oracle.ref_region_run if oracle.have_ref() else oracle.region_run
        res = run(x, ANCHORS[li], W, H, 3, 20, 0.6, 0.3)
        d = res[3] if len(res) == 4 else oracle.region_draw(res[1], res[2], 0.6)
        dets.append(np.asarray(d).reshape(-1, 6))
    np.savez_compressed(ROOT / 'tests' / 'golden' / 'kmodel_dog_golden.npz', image=img, y1_q=outs[0], y2_q=outs[1],
                        dets=np.concatenate(dets, 0), anchors=np.array(ANCHORS, np.float32))
    print('dets', np.concatenate(dets, 0))


if __name__ == '__main__':
    main()
