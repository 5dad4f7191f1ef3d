"""preprocess_ref.py — CPU ORACLE (test infrastructure, NOT the product path).

Pure-Python/numpy restatement of Helper._process_img's letterbox (tools/utils.py:378-399), written independently
of k210_yolo_framework_amd/helper.py (explicit per-pixel loops, float64): scale = min(in_wh/img_wh), translation =
trunc((in_wh - img_wh*scale)/2), output(x,y) = bilinear(input, ((x-tx)/s, (y-ty)/s)), zero outside, truncating cast.
PARITY UNPINNED for the warp itself (scikit-image 0.15 is third-party and absent); the identity case (dog.jpg is
already 320x224) and the published constants for people.jpg (scale 0.598930, translation (10,0)) are known answers."""
import math

import numpy as np


def letterbox_params(src_hw, dst_hw):
    sh, sw = src_hw
    dh, dw = dst_hw
    scale = min(dw / sw, dh / sh)
    return scale, int((dw - sw * scale) / 2), int((dh - sh * scale) / 2)


def letterbox(img: np.ndarray, dst_hw) -> np.ndarray:
    sh, sw = img.shape[:2]
    dh, dw = dst_hw
    scale, tx, ty = letterbox_params((sh, sw), dst_hw)
    out = np.zeros((dh, dw, 3), np.uint8)
    f = img.astype(np.float64)
    for y in range(dh):
        fy = (y - ty) / scale
        if not (-1.0 < fy < sh):
            continue
        y0 = math.floor(fy)
        ay = fy - y0
        for x in range(dw):
            fx = (x - tx) / scale
            if not (-1.0 < fx < sw):
                continue
            x0 = math.floor(fx)
            ax = fx - x0

            def px(yy, xx):
                return f[yy, xx] if (0 <= yy < sh and 0 <= xx < sw) else np.zeros(3)
            v = px(y0, x0) * (1.0 - ay) * (1.0 - ax) + px(y0, x0 + 1) * (1.0 - ay) * ax + \
                px(y0 + 1, x0) * ay * (1.0 - ax) + px(y0 + 1, x0 + 1) * ay * ax
            out[y, x] = v.astype(np.uint8)
    return out
