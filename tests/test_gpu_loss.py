"""GPU parity: yk_yolo_loss (loss terms, dL/dy_pred, ignore mask, precision/recall counters) vs oracle/loss_ref.py."""
import numpy as np
import pytest

from oracle import loss_ref
from tests.test_oracle_loss import make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('seed,layer,B,weights', [(0, 0, 4, (1, 1, 1)), (1, 1, 16, (1, 1, 1)), (2, 1, 16, (5, 0.5, 0.5)), (3, 0, 3, (5, 0.5, 0.5))])
def test_loss_grad_ignore_counts_vs_oracle(seed, layer, B, weights):
    import torch
    from k210_yolo_framework_amd import engine
    h, y_true, y_pred = make_case(seed, B=B, layer=layer)
    ow, nw, ww = weights
    ref_l, ref_g, ref_i, ref_c = loss_ref.yolo_loss(y_true, y_pred, h.anchors[layer], 0.7, 0.5, ow, nw, ww)
    counts = torch.zeros(3, device='cuda')
    loss, grad, ign = engine.yolo_loss(torch.from_numpy(y_true).cuda(), torch.from_numpy(y_pred).cuda(), h.anchors[layer],
                                       0.7, 0.5, ow, nw, ww, counts=counts, want_ignore=True)
    torch.cuda.synchronize()
    loss, grad, ign = loss.cpu().numpy(), grad.cpu().numpy(), ign.cpu().numpy()
    for k, name in enumerate(('total', 'xy', 'wh', 'obj', 'noobj', 'cls')):
        assert abs(loss[k] - ref_l[name]) <= 2e-5 * max(1.0, abs(ref_l[name])), (name, loss[k], ref_l[name])
    assert np.array_equal(ign, ref_i)                                   # index work: exact
    np.testing.assert_allclose(grad, ref_g, rtol=2e-5, atol=1e-7)
    assert tuple(int(v) for v in counts.cpu()) == ref_c                 # exact counters


def test_metrics_accumulate_like_keras_assign_add_and_empty_images():
    import torch
    from k210_yolo_framework_amd.helper import Yolo_Precision, Yolo_Recall, create_loss_fn
    h, y_true, y_pred = make_case(7, B=8, layer=1)
    y_true[3] = 0
    h.batch_size = 8
    fn = create_loss_fn(h, 0.7, 0.5, 1.0, 1.0, 1.0, 1)
    yt, yp = torch.from_numpy(y_true).cuda(), torch.from_numpy(y_pred).cuda()
    total = float(fn(yt, yp))
    ref_l, ref_g, _, (tp, fp, fnn) = loss_ref.yolo_loss(y_true, y_pred, h.anchors[1], 0.7, 0.5, 1, 1, 1)
    assert abs(total - ref_l['total']) <= 2e-5 * abs(ref_l['total'])
    np.testing.assert_allclose(fn.grad.cpu().numpy(), ref_g, rtol=2e-5, atol=1e-7)
    p, r = Yolo_Precision(0.7, name='p'), Yolo_Recall(0.7, name='r')
    for _ in range(2):                                                   # two batches accumulate
        p.update_state(yt, yp)
        r.update_state(yt, yp)
    assert abs(p.result() - tp / (tp + fp)) < 1e-6 and abs(r.result() - tp / (tp + fnn)) < 1e-6


def test_many_ground_truth_boxes_overflow_path():
    """> 1024 labelled cells in one image: the kernel scans the label tensor instead of the LDS list."""
    import torch
    from k210_yolo_framework_amd import engine
    rng = np.random.default_rng(4)
    B, hh, ww, A, C = 2, 26, 26, 3, 4
    y_true = np.zeros((B, hh, ww, A, 5 + C), np.float32)
    y_true[0, ..., 0:2] = rng.uniform(0.05, 0.95, (hh, ww, A, 2))
    y_true[0, ..., 2:4] = rng.uniform(0.02, 0.2, (hh, ww, A, 2))
    y_true[0, ..., 4] = 1
    y_true[0, ..., 5] = 1
    y_pred = rng.normal(0, 1, y_true.shape).astype(np.float32)
    anc = rng.uniform(0.05, 0.5, (A, 2)).astype(np.float32)
    ref_l, ref_g, ref_i, _ = loss_ref.yolo_loss(y_true, y_pred, anc, 0.7, 0.5, 1, 1, 1)
    loss, grad, ign = engine.yolo_loss(torch.from_numpy(y_true).cuda(), torch.from_numpy(y_pred).cuda(), anc, 0.7, 0.5, 1, 1, 1,
                                       want_ignore=True)
    torch.cuda.synchronize()
    assert np.mean(ign.cpu().numpy() == ref_i) > 0.9995               # IoU within 1 ulp of the threshold may flip
    assert abs(float(loss[0]) - ref_l['total']) <= 1e-4 * abs(ref_l['total'])


def test_calc_ignore_mask_free_function_matches_numpy_tf_iou():
    """tools/utils.py:662-705 through the Helper-API front (helper.calc_ignore_mask -> yk_yolo_loss mask output)."""
    from k210_yolo_framework_amd.helper import Helper, VOC_ANCHORS, calc_ignore_mask, tf_iou, tf_xywh_to_all
    h = Helper(None, 20, VOC_ANCHORS, [[224, 320]], [[7, 10], [14, 20]])
    rng = np.random.default_rng(8)
    B, layer = 3, 1
    gh, gw = h.out_hw[layer]
    p_xy, p_wh = rng.normal(0, 1, (B, gh, gw, 3, 2)).astype(np.float32), rng.normal(0, .5, (B, gh, gw, 3, 2)).astype(np.float32)
    t_xy, t_wh = rng.uniform(0.1, 0.9, (B, gh, gw, 3, 2)).astype(np.float32), rng.uniform(0.05, 0.5, (B, gh, gw, 3, 2)).astype(np.float32)
    mask = rng.uniform(size=(B, gh, gw, 3)) < 0.02
    mask[1] = False                                              # an image without objects: everything is "ignore = 1"
    got = calc_ignore_mask(t_xy, t_wh, p_xy, p_wh, mask, 0.3, layer, h).cpu().numpy()
    assert got.shape == (B, gh, gw, 3, 1)
    a_xy, a_wh = tf_xywh_to_all(p_xy.astype(np.float64), p_wh.astype(np.float64), layer, h)
    for b in range(B):
        if mask[b].any():
            best = tf_iou(a_xy[b], a_wh[b], t_xy[b][mask[b]], t_wh[b][mask[b]]).max(-1, keepdims=True)
        else:
            best = np.full((gh, gw, 3, 1), -np.inf)
        want = (best < 0.3).astype(np.float32)
        decided = np.abs(best - 0.3) > 1e-5                      # fp32 (GPU) vs float64 (here) only differ on the threshold itself
        assert np.array_equal(got[b][decided], want[decided])
    assert got[1].min() == 1.0 and 0 < got[0].mean() < 1
