"""GPU parity: C-mode region layer (libyolo_hip.so) vs the reference's golden vectors and the oracle.
All calls go through the C-ABI (drop-in region_layer_* and yk_region_batched)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

# Bit-exact: the device evaluates expf with glibc's own algorithm (yk_expf_glibc, csrc/yk_common.h), divisions are IEEE, the
# translation unit is built with -ffp-contract=off, so every float the reference's x86-64 build produces is reproduced.


@pytest.fixture(scope='module')
def hip():
    from k210_yolo_framework_amd import engine
    engine.require_gpu()
    return engine.lib()


def _cmp(name, got, want, thr):
    out, boxes, probs, dets = got
    wo, wb, wp, wd = want
    np.testing.assert_array_equal(out.view(np.uint32), wo.view(np.uint32), err_msg=name)
    np.testing.assert_array_equal(boxes.view(np.uint32), wb.view(np.uint32), err_msg=name)
    np.testing.assert_array_equal(probs.view(np.uint32), wp.view(np.uint32), err_msg=name)
    # draw callback rows (x1, y1, x2, y2, class, prob bits): integer work, exact, in callback order
    np.testing.assert_array_equal(dets, wd, err_msg=name)


def test_dropin_abi_on_reference_golden_vectors(hip, golden_dir):
    g = np.load(golden_dir / 'region_golden.npz')
    names = sorted({k.split('/')[0] for k in g.files if '/' in k})
    for n in names:
        W, H, A, Cn, li, nw, nh = (int(v) for v in g[n + '/meta'])
        thr, nms = (float(v) for v in g[n + '/thr'])
        got = oracle.drive_region_abi(hip, g[n + '/input'], g['anchors'][li], W, H, A, Cn, thr, nms, (nw, nh))
        _cmp(n, got, (g[n + '/output'], g[n + '/boxes'], g[n + '/probs'], g[n + '/dets']), thr)


def test_dropin_abi_main_c_call_sequence(hip):
    """main.c:278-324: two layers initialised up front, run back to back, drawn afterwards."""
    rng = np.random.default_rng(11)
    anc = [[0.76120044, 0.57155991, 0.6923348, 0.88535553, 0.47163042, 0.34163313],
           [0.33340788, 0.70065861, 0.18124964, 0.38986752, 0.08497349, 0.1527057]]
    x0 = rng.uniform(-5, 5, (3, 25, 7, 10)).astype(np.float32)
    x1 = rng.uniform(-5, 5, (3, 25, 14, 20)).astype(np.float32)
    a = oracle.drive_region_abi(hip, x0, anc[0], 10, 7, 3, 20, 0.6, 0.3)
    b = oracle.drive_region_abi(hip, x1, anc[1], 20, 14, 3, 20, 0.6, 0.3)
    for x, (W, H), an, got in ((x0, (10, 7), anc[0], a), (x1, (20, 14), anc[1], b)):
        o, bx, pr = oracle.region_run(x, an, W, H, 3, 20, 0.6, 0.3)
        _cmp('seq', got, (o, bx, pr, oracle.region_draw(bx, pr, 0.6)), 0.6)


@pytest.mark.parametrize('layout', ['chw', 'hwc'])
@pytest.mark.parametrize('W,H,A,Cn,thr,nms,net', [(10, 7, 3, 20, 0.6, 0.3, (320, 224)), (20, 14, 3, 20, 0.05, 0.3, (320, 224)),
                                                  (13, 13, 3, 20, 0.2, 0.45, (416, 416)), (5, 3, 5, 2, 0.05, 0.2, (160, 224)),
                                                  (52, 52, 3, 4, 0.01, 0.4, (416, 416))])
def test_batched_vs_oracle(hip, layout, W, H, A, Cn, thr, nms, net):
    import torch
    from k210_yolo_framework_amd import engine
    rng = np.random.default_rng(W * 100 + H)
    B = 5
    anchor = rng.uniform(0.05, 0.9, 2 * A).astype(np.float32)
    x = rng.uniform(-4, 4, (B, A, 5 + Cn, H, W)).astype(np.float32)
    if layout == 'chw':
        xin = torch.from_numpy(x.reshape(B, A * (5 + Cn), H, W)).cuda()
    else:
        xin = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 4, 1, 2)).reshape(B, H, W, A * (5 + Cn))).cuda()
    out, boxes, probs = engine.region_batched(xin, W, H, A, Cn, anchor, thr, nms, net, (320, 224), layout)
    torch.cuda.synchronize()
    for b in range(B):
        o, bx, pr = oracle.region_run(x[b], anchor, W, H, A, Cn, thr, nms, net)
        np.testing.assert_array_equal(out[b].cpu().numpy().ravel().view(np.uint32), o.view(np.uint32))
        np.testing.assert_array_equal(boxes[b].cpu().numpy().view(np.uint32), bx.view(np.uint32))
        np.testing.assert_array_equal(probs[b].cpu().numpy().view(np.uint32), pr.view(np.uint32))


def test_nms_overflow_path_many_candidates(hip):
    """> YK_NMS_MAXC (2048) candidates of one class: the global-memory path must give the same survivors."""
    import torch
    from k210_yolo_framework_amd import engine
    rng = np.random.default_rng(3)
    W, H, A, Cn = 52, 52, 3, 1
    x = rng.uniform(-1, 1, (1, A, 5 + Cn, H, W)).astype(np.float32)
    x[:, :, 4] = rng.uniform(2, 5, (1, A, H, W))          # objectness high everywhere -> 8112 candidates
    x[:, :, 2:4] = rng.uniform(-3, -2, (1, A, 2, H, W))   # small boxes: most survive NMS
    anchor = np.array([0.3, 0.3, 0.2, 0.4, 0.4, 0.2], np.float32)
    xin = torch.from_numpy(x.reshape(1, A * (5 + Cn), H, W)).cuda()
    _, boxes, probs = engine.region_batched(xin, W, H, A, Cn, anchor, 0.5, 0.3, (416, 416), (320, 224), 'chw', False)
    torch.cuda.synchronize()
    o, bx, pr = oracle.region_run(x[0], anchor, W, H, A, Cn, 0.5, 0.3, (416, 416))
    p = probs[0].cpu().numpy()
    assert (pr[:, 0] != 0).sum() > 100
    assert np.array_equal(p[:, 0] != 0, pr[:, 0] != 0)
