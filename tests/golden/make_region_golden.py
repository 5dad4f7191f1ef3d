"""Generate tests/golden/region_*.npz by RUNNING THE REFERENCE's region_layer.c.

Requires oracle/_ref/libregion_ref.so (oracle/build_ref.sh, needs /root/reference).
The fixtures hold only data: seeded inputs and the reference's outputs
(rl->output, rl->boxes, rl->probs after NMS, and the ordered draw-callback list).

Cases (anchors = main.c:46-52, threshold/nms = main.c:280-287):
  l0_uniform / l1_uniform : U(-6,6) logits, 7x10 and 14x20 (SURVEY 8(d) 'adversarial')
  l0_typical / l1_typical : conf ~ N(-4,2) with planted confident objects
  l1_letterbox            : net 416x416 vs the hard-coded 320x224 image (region_layer.c:24-25)
                            -> non-identity correct_region_boxes
  l0_c1                   : single class (channels = 3*6), degenerate softmax
  l0_empty / l1_extreme / l0_all : nothing above the threshold / logits in +-100 / threshold 0 (edge cases)
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

ANCHORS = [[0.76120044, 0.57155991, 0.6923348, 0.88535553, 0.47163042, 0.34163313],
           [0.33340788, 0.70065861, 0.18124964, 0.38986752, 0.08497349, 0.1527057]]


def typical(rng, W, H, A, C, n_obj=5):
    x = rng.normal(0, 1, (A, 5 + C, H, W)).astype(np.float32)
    x[:, 4] = rng.normal(-4, 2, (A, H, W))
    for _ in range(n_obj):
        n, r, c, k = rng.integers(A), rng.integers(H), rng.integers(W), rng.integers(C)
        for dn in range(A):                      # neighbours so that NMS has work to do
            x[dn, 4, r, c] = rng.uniform(2, 6)
            x[dn, 5 + k, r, c] = rng.uniform(4, 8)
        x[n, 4, r, min(c + 1, W - 1)] = rng.uniform(2, 6)
        x[n, 5 + k, r, min(c + 1, W - 1)] = rng.uniform(4, 8)
    return x


def main():
    assert oracle.have_ref(), 'run oracle/build_ref.sh first (needs /root/reference)'
    rng = np.random.default_rng(20190709)
    cases = {
        'l0_uniform': (10, 7, 3, 20, 0, 0.6, 0.3, (320, 224), 'u'),
        'l1_uniform': (20, 14, 3, 20, 1, 0.6, 0.3, (320, 224), 'u'),
        'l0_typical': (10, 7, 3, 20, 0, 0.6, 0.3, (320, 224), 't'),
        'l1_typical': (20, 14, 3, 20, 1, 0.6, 0.3, (320, 224), 't'),
        'l1_letterbox': (20, 14, 3, 20, 1, 0.5, 0.45, (416, 416), 't'),
        'l0_c1': (10, 7, 3, 1, 0, 0.3, 0.3, (320, 224), 'u'),
        # edge cases (appended: the cases above keep their random draws)
        'l0_empty': (10, 7, 3, 20, 0, 0.6, 0.3, (320, 224), 'e'),        # nothing reaches the threshold: no box survives, no draw call
        'l1_extreme': (20, 14, 3, 20, 1, 0.6, 0.3, (320, 224), 'x'),     # logits U(-100, 100): expf overflow / underflow / subnormals
        'l0_all': (10, 7, 3, 20, 0, 0.0, 0.3, (320, 224), 'u'),          # threshold 0: every class of every box is a candidate
    }
    out = {}
    for name, (W, H, A, Cn, li, thr, nms, net_wh, kind) in cases.items():
        if kind == 'u':
            x = rng.uniform(-6, 6, (A, 5 + Cn, H, W)).astype(np.float32)
        elif kind == 'e':
            x = rng.normal(0, 1, (A, 5 + Cn, H, W)).astype(np.float32)
            x[:, 4] = -12.0
        elif kind == 'x':
            x = rng.uniform(-100, 100, (A, 5 + Cn, H, W)).astype(np.float32)
        else:
            x = typical(rng, W, H, A, Cn)
        o, b, p, d = oracle.ref_region_run(x, ANCHORS[li], W, H, A, Cn, thr, nms, net_wh)
        out[name + '/input'] = x
        out[name + '/meta'] = np.array([W, H, A, Cn, li, net_wh[0], net_wh[1]], np.int32)
        out[name + '/thr'] = np.array([thr, nms], np.float32)
        out[name + '/output'] = o
        out[name + '/boxes'] = b
        out[name + '/probs'] = p
        out[name + '/dets'] = d
        print(name, 'nonzero probs', int((p[:, :Cn] > 0).sum()), 'dets', len(d))
    out['anchors'] = np.array(ANCHORS, np.float32)
    np.savez_compressed(Path(__file__).with_name('region_golden.npz'), **out)


if __name__ == '__main__':
    main()
