"""Pin oracle/region_layer_ref.c against the reference's own region_layer.c.

(a) committed golden vectors produced by the compiled reference (tests/golden/make_region_golden.py)
(b) the live oracle/_ref build, when present, on fresh seeded inputs (incl. hypothesis shapes)
"""
import numpy as np
import pytest

import oracle


def _cases(golden_dir):
    g = np.load(golden_dir / 'region_golden.npz')
    names = sorted({k.split('/')[0] for k in g.files if '/' in k})
    return g, names


def test_golden_cases_bit_exact(golden_dir):
    g, names = _cases(golden_dir)
    assert len(names) >= 6
    for n in names:
        W, H, A, C, li, nw, nh = g[n + '/meta']
        thr, nms = g[n + '/thr']
        out, boxes, probs = oracle.region_run(g[n + '/input'], g['anchors'][li], W, H, A, C, thr, nms, (nw, nh))
        # same glibc expf on the generating and the checking side -> bit-exact expected
        np.testing.assert_array_equal(out, g[n + '/output'], err_msg=n)
        np.testing.assert_array_equal(boxes, g[n + '/boxes'], err_msg=n)
        np.testing.assert_array_equal(probs, g[n + '/probs'], err_msg=n)
        dets = oracle.region_draw(boxes, probs, thr)
        np.testing.assert_array_equal(dets, g[n + '/dets'], err_msg=n)


@pytest.mark.skipif(not oracle.have_ref(), reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('W,H,A,C,thr,nms,net', [
    (10, 7, 3, 20, 0.6, 0.3, (320, 224)), (20, 14, 3, 20, 0.1, 0.3, (320, 224)),
    (13, 13, 3, 20, 0.2, 0.5, (416, 416)), (5, 3, 5, 2, 0.05, 0.2, (160, 224)), (1, 1, 1, 1, 0.0, 0.5, (320, 224)),
    (20, 14, 3, 1, 0.3, 0.3, (320, 224)),
])
def test_live_reference_bit_exact(W, H, A, C, thr, nms, net):
    rng = np.random.default_rng(W * 1000 + H * 10 + C)
    anchor = rng.uniform(0.05, 0.9, 2 * A).astype(np.float32)
    for scale in (6.0, 2.0):
        x = rng.uniform(-scale, scale, (A, 5 + C, H, W)).astype(np.float32)
        ro, rb, rp, rd = oracle.ref_region_run(x, anchor, W, H, A, C, thr, nms, net)
        out, boxes, probs = oracle.region_run(x, anchor, W, H, A, C, thr, nms, net)
        np.testing.assert_array_equal(out, ro)
        np.testing.assert_array_equal(boxes, rb)
        np.testing.assert_array_equal(probs, rp)
        np.testing.assert_array_equal(oracle.region_draw(boxes, probs, thr), rd)


def test_negative_coordinate_cast_is_defined():
    # region_layer.c:397-400 converts possibly-negative floats to uint32 (UB); the restatement and the
    # new ABI define it as (uint32_t)(int64_t)x.  A box hanging over the left/top edge exercises it.
    boxes = np.array([[0.01, 0.01, 0.5, 0.5]], np.float32)
    probs = np.array([[0.9, 0.9]], np.float32)
    d = oracle.region_draw(boxes, probs, 0.5)
    x1 = np.float32(0.01) * np.float32(320) - (np.float32(0.5) * np.float32(320) / np.float32(2))
    assert d[0, 0] == np.uint32(np.int64(x1) & 0xFFFFFFFF)
    assert d[0, 0] > 4_000_000_000
