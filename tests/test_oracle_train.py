"""oracle/train_ref.py known answers (the training-step oracle is 'parity unpinned' against TensorFlow, so it is anchored here by
hand-derived values and by finite differences of its own float64 loss)."""
import numpy as np
import torch

from k210_yolo_framework_amd import netspec as ns
from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS
from oracle import train_ref


def _tiny(seed=0, B=2):
    spec = ns.yolo_mobilev1([32, 64, 3], 3, 20, alpha=0.25)
    w = spec.init_weights(seed)
    h = Helper(None, 20, VOC_ANCHORS, [[32, 64]], [list(x) for x in spec.out_hw()])
    rng = np.random.default_rng(seed)
    ys = [[] for _ in spec.outputs]
    for b in range(B):
        boxes = np.stack([rng.integers(0, 20, 2), rng.uniform(.2, .8, 2), rng.uniform(.2, .8, 2), rng.uniform(.1, .6, 2), rng.uniform(.1, .6, 2)], 1)
        for i, lab in enumerate(h.box_to_label(boxes)):
            ys[i].append(lab)
    return spec, w, h, rng.uniform(0, 1, (B, 32, 64, 3)).astype(np.float32), [np.stack(y).astype(np.float32) for y in ys]


def test_adam_reference_two_steps_by_hand():
    """keras Adam: lr_t = lr/(1+decay*it) * sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps)."""
    opt = train_ref.AdamRef(0.1, decay=0.5)
    w = {'p': np.array([1.0, -2.0])}
    g1, g2 = np.array([0.5, -1.0]), np.array([0.25, 2.0])
    w = opt.apply(w, {'p': g1})
    m1, v1 = 0.1 * g1, 0.001 * g1 ** 2
    lr1 = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    p1 = np.array([1.0, -2.0]) - lr1 * m1 / (np.sqrt(v1) + 1e-7)
    np.testing.assert_allclose(w['p'], p1, rtol=1e-12)
    w = opt.apply(w, {'p': g2})
    m2, v2 = 0.9 * m1 + 0.1 * g2, 0.999 * v1 + 0.001 * g2 ** 2
    lr2 = 0.1 / (1 + 0.5 * 1) * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    np.testing.assert_allclose(w['p'], p1 - lr2 * m2 / (np.sqrt(v2) + 1e-7), rtol=1e-12)


def test_training_forward_uses_biased_batch_statistics_and_regulariser_covers_only_darknet_convs():
    spec, w, h, x, yt = _tiny()
    data, reg, g, stats, preds = train_ref.loss_and_grads(spec, w, x, yt, h.anchors)
    # stem: conv then BN with the batch's own mean / biased variance (recomputed here with numpy from a torch conv)
    k = torch.from_numpy(w['conv1/kernel']).double().permute(3, 2, 0, 1)
    z = torch.nn.functional.conv2d(torch.nn.functional.pad(torch.from_numpy(x).double().permute(0, 3, 1, 2), (1, 1, 1, 1)), k, stride=2).numpy()
    np.testing.assert_allclose(stats['conv1_bn'][0], z.mean((0, 2, 3)), rtol=1e-10)
    np.testing.assert_allclose(stats['conv1_bn'][1], z.var((0, 2, 3)), rtol=1e-10)              # ddof = 0
    want = sum(5e-4 * float((np.asarray(w[l.name + '/kernel'], np.float64) ** 2).sum()) for l in spec.layers if l.name.startswith('head_conv'))
    assert abs(reg - want) <= 1e-12 * want and reg > 0
    assert all(np.abs(g[l.name + '/kernel']).max() > 0 for l in spec.layers)
    assert [p.shape for p in preds] == [(2, 1, 2, 3, 25), (2, 2, 4, 3, 25)]


def test_gradients_match_central_finite_differences_of_the_float64_loss():
    spec, w, h, x, yt = _tiny(3)
    _, _, g, _, _ = train_ref.loss_and_grads(spec, w, x, yt, h.anchors)

    def total(wd):
        d, r, _, _, _ = train_ref.loss_and_grads(spec, wd, x, yt, h.anchors)
        return d + r

    rng = np.random.default_rng(0)
    for name in ['conv1/kernel', 'conv_dw_3/kernel', 'conv_pw_7_bn/gamma', 'conv_pw_13_bn/beta', 'head_conv_1/kernel', 'head_conv_2/bias',
                 'head_conv_5/kernel']:
        a = np.asarray(w[name], np.float64)
        idx = tuple(int(rng.integers(0, s)) for s in a.shape)
        eps = 1e-5 * max(1.0, abs(a[idx]))
        wp, wm = dict(w), dict(w)
        ap, am = a.copy(), a.copy()
        ap[idx] += eps
        am[idx] -= eps
        wp[name], wm[name] = ap, am
        fd = (total(wp) - total(wm)) / (2 * eps)
        assert abs(fd - g[name][idx]) <= 1e-5 * max(1.0, abs(fd)) + 1e-7, (name, idx, fd, g[name][idx])
