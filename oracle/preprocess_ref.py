"""preprocess_ref.py — CPU ORACLE (test infrastructure, NOT the product path).

Pure-Python restatement of Helper._process_img's letterbox (tools/utils.py:378-399), written independently of
k210_yolo_framework_amd/helper.py (explicit per-pixel loops, Python floats = IEEE double): scale = min(in_wh/img_wh),
translation = trunc((in_wh - img_wh*scale)/2), then scikit-image's warp arithmetic for order=1 / mode='constant':
inverse matrix entries 1/s and -(t*(1/s)), source coordinate = m*x + b, corner pixels floor/ceil, column blend then row
blend, zero outside, truncating uint8 cast.

PINNED: tests/test_oracle_pre.py compares this file bit for bit with tests/golden/letterbox_golden.npz, outputs of the real
scikit-image (0.18.3; generator tests/golden/make_letterbox_golden.py) for seven source sizes."""
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
    inv = 1.0 / scale
    bx, by = -(tx * inv), -(ty * inv)
    out = np.zeros((dh, dw, 3), np.uint8)
    f = img.astype(np.float64)

    def px(yy, xx):
        return f[yy, xx] if (0 <= yy < sh and 0 <= xx < sw) else np.zeros(3)
    for y in range(dh):
        r = inv * y + by
        r0, r1 = math.floor(r), math.ceil(r)
        dr = r - r0
        if r1 < 0 or r0 >= sh:
            continue
        for x in range(dw):
            c = inv * x + bx
            c0, c1 = math.floor(c), math.ceil(c)
            if c1 < 0 or c0 >= sw:
                continue
            dc = c - c0
            top = (1.0 - dc) * px(r0, c0) + dc * px(r0, c1)
            bottom = (1.0 - dc) * px(r1, c0) + dc * px(r1, c1)
            out[y, x] = ((1.0 - dr) * top + dr * bottom).astype(np.uint8)
    return out
