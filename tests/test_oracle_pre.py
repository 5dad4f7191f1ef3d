"""Letterbox parity against the REAL scikit-image: tests/golden/letterbox_golden.npz holds inputs and outputs of
skimage.transform.warp called exactly as tools/utils.py:378-399 calls it (generator: tests/golden/make_letterbox_golden.py).
The oracle (oracle/preprocess_ref.py) and the host mirror (helper.letterbox_bilinear) must reproduce every byte."""
import numpy as np
import pytest

from oracle import preprocess_ref as pr
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS, letterbox_bilinear


def _cases(golden_dir):
    g = np.load(golden_dir / 'letterbox_golden.npz')
    i = 0
    while f'img{i}' in g.files:
        yield i, g[f'img{i}'], g[f'out{i}'], g[f'par{i}']
        i += 1


def test_host_mirror_bit_exact_vs_skimage(golden_dir):
    n = 0
    for i, img, ref, par in _cases(golden_dir):
        dst = (int(par[3]), int(par[4]))
        h = Helper(None, 20, VOC_ANCHORS, [list(dst)], [[7, 10], [14, 20]])
        s, t = h.letterbox_params(img.shape[:2])
        assert s[0] == par[0] and t.tolist() == [int(par[1]), int(par[2])]          # scale / translation rule, utils.py:381-385
        np.testing.assert_array_equal(letterbox_bilinear(img, dst, float(s[0]), t), ref, err_msg=f'case {i}')
        out, _ = h._process_img(img.copy(), None, is_training=False, is_resize=True)
        np.testing.assert_array_equal(out, ref / np.max(ref))                      # utils.py:405
        n += 1
    assert n >= 6


def test_oracle_bit_exact_vs_skimage(golden_dir):
    for i, img, ref, par in _cases(golden_dir):
        dst = (int(par[3]), int(par[4]))
        got = pr.letterbox(img, dst)                     # per-pixel Python loops: ~0.3 s for a 224x320 output
        np.testing.assert_array_equal(got, ref, err_msg=f'case {i}')
        assert pr.letterbox_params(img.shape[:2], dst) == (par[0], int(par[1]), int(par[2]))
