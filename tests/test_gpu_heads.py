"""The detection heads of the f16x2 mode as one launch of per-image workgroup clusters (csrc/yk_xheads.h; yolonet.py:23-60: Conv3x3 ->
network-output Conv1x1 pairs and the Conv1x1 in front of UpSampling2D, K split over the members of a cluster): same network outputs
as the launch-per-conv form and as the fp32 oracle, for any batch size, independent of the batch."""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = pytest.mark.gpu


def _outs(spec, w, frames, heads, max_batch=None):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_HEADS'] = '1' if heads else '0'
    os.environ['YK_FUSE_HEAD'] = '0'                                      # the reference form here is one launch per conv (tests/test_gpu_fin.py holds the fused heads)
    try:
        plan = engine.Plan(spec, w, max_batch=max_batch or len(frames), precision='f16x2')
    finally:
        os.environ.pop('YK_HEADS', None)
        os.environ.pop('YK_FUSE_HEAD', None)
    names = [l[0] for l in plan.launches()]
    plan.run_u8(torch.from_numpy(frames).cuda())
    plan.check()
    outs = [o[:len(frames)].cpu().numpy().copy() for o in plan.outputs()]
    plan.close()
    return outs, names


@pytest.mark.parametrize('B', [1, 3, 32, 40])
def test_heads_launch_matches_plain_launches_and_the_oracle(B):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(B).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
    got, names = _outs(spec, w, frames, True)
    assert sum(n.startswith('x:heads') for n in names) == 1 and names[-1].startswith('x:heads'), names
    assert not any(n.startswith('x:conv') for n in names), names          # all five head convs are inside it
    ref, names0 = _outs(spec, w, frames, False)
    assert not any(n.startswith('x:heads') for n in names0) and len(names0) == len(names) + 4
    for g, r in zip(got, ref):
        assert np.isfinite(g).all()
        assert np.abs(g - r).max() <= 2e-5 * np.abs(r).max()            # two f16x2 evaluations: rounding differs at the 2^-22 level only
    nb = min(B, 4)
    ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:nb]), emulate_f16=False, out_ids=spec.outputs)
    for g, r in zip(got, ref32):
        assert np.abs(g[:nb] - r).max() <= 1e-4 * np.abs(r).max()


def test_an_image_does_not_depend_on_its_batch_nor_on_max_batch():
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=2)
    f = np.random.default_rng(1).integers(0, 256, (9, 224, 320, 3), dtype=np.uint8)
    f[4] //= 20
    a, _ = _outs(spec, w, f, True)
    b, _ = _outs(spec, w, np.ascontiguousarray(f[::-1]), True, max_batch=16)
    c, _ = _outs(spec, w, f[4:5].copy(), True, max_batch=2)
    for x, y, z in zip(a, b, c):
        np.testing.assert_array_equal(x, y[::-1])
        np.testing.assert_array_equal(x[4:5], z)
    a2, _ = _outs(spec, w, f, True)
    for x, y in zip(a, a2):
        np.testing.assert_array_equal(x, y)                               # partial sums are added in slot order: reruns are bit-identical


def test_other_networks_and_sizes_agree_with_their_plain_launches():
    """Whatever the plan chooses for a network / image size (heads launch where every conv fits an instantiation, plain launches
    otherwise), outputs equal the YK_HEADS=0 plan's."""
    taken = 0
    for name, shape, alpha in (('yolo_mobilev1', (64, 96, 3), 0.75), ('yolo_mobilev1', (128, 160, 3), 1.0), ('yolo_mobilev1', (96, 64, 3), 0.5),
                               ('yolo_mobilev2', (224, 320, 3), 1.0), ('tiny_yolo', (416, 416, 3), 1.0), ('yolo', (96, 128, 3), 1.0)):
        spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
        w = spec.init_weights(seed=1)
        f = np.random.default_rng(0).integers(0, 256, (2, *shape), dtype=np.uint8)
        a, na = _outs(spec, w, f, True)
        b, nb = _outs(spec, w, f, False)
        taken += any(n.startswith('x:heads') for n in na)
        tol = 5e-4 if name == 'yolo_mobilev2' else 1e-4                   # (tests/test_gpu_net.py: undamped MobileNet-v2 amplifies rounding noise)
        for x, y in zip(a, b):
            assert np.isfinite(x).all(), name
            assert np.abs(x - y).max() <= tol * max(np.abs(y).max(), 1e-3), (name, shape)
    assert taken >= 2


def test_schedules_select_the_launch_form_and_agree():
    """YK_SCHEDULE_LATENCY (engine.Plan's default, Pipeline depth 1) = the two cluster launches; YK_SCHEDULE_THROUGHPUT (yk_plan_create's
    default, Pipeline depth >= 2) = one launch per layer.  Same arithmetic: outputs agree at the rounding level."""
    import torch
    from k210_yolo_framework_amd import engine
    from k210_yolo_framework_amd.helper import VOC_ANCHORS
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=3)
    f = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (5, 224, 320, 3), dtype=np.uint8)).cuda()
    outs = {}
    for sched in ('latency', 'throughput'):
        with engine.Plan(spec, w, max_batch=8, schedule=sched) as plan:
            names = [l[0] for l in plan.launches()]
            cluster = [n for n in names if n.startswith('x:persist') or n.startswith('x:heads')]
            assert len(cluster) == (2 if sched == 'latency' else 0), names
            if sched == 'throughput':
                # the five 14x20x384 blocks are ONE launch each there, all 384 output channels in one workgroup (one depthwise pass); the two
                # 7x10 blocks stay depthwise + 1x1 launches; the K-split 3x3 head convs run on two ring stages
                wide = [n for n in names if 'dw3x3s1+conv1x1_384to384' in n]
                assert len(wide) == 5 and all(',384ch,' in n for n in wide), names
                assert sum(n.startswith('x:dw3x3s') and '+conv' not in n for n in names) == 2, names
                assert all('ring2' in n for n in names if 'splitk' in n), names
            plan.run_u8(f)
            plan.check()
            outs[sched] = [o[:5].cpu().numpy().copy() for o in plan.outputs()]
    for a, b in zip(outs['latency'], outs['throughput']):
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
    with pytest.raises(engine.YkError, match='schedule'):
        engine.Plan(spec, w, max_batch=2, schedule='fast')
    p1 = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=4, depth=1)
    p3 = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=4, depth=3)
    assert (p1.schedule, p3.schedule) == ('latency', 'throughput')
    assert any(n[0].startswith('x:heads') for n in p1.plans[0].launches()) and not any(n[0].startswith('x:heads') for n in p3.plans[0].launches())
    p1.close()
    p3.close()


def test_write_through_exchange_gives_the_same_bits():
    """The cluster launches verify their placement (all members of an image on one XCD -> plain stores, the data stays in that L2) and fall
    back to write-through stores + L1-bypassing loads otherwise.  On a healthy box the fallback never runs; YK_CLUSTER_WT=1 forces it:
    same arithmetic, so the outputs must be bit-identical (and the run must not report an unassembled cluster)."""
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=4)
    f = np.random.default_rng(7).integers(0, 256, (11, 224, 320, 3), dtype=np.uint8)
    a, names = _outs(spec, w, f, True)
    assert any(n.startswith('x:persist') for n in names) and any(n.startswith('x:heads') for n in names)
    os.environ['YK_CLUSTER_WT'] = '1'
    try:
        b, _ = _outs(spec, w, f, True)
        c, _ = _outs(spec, w, f, True)
    finally:
        os.environ.pop('YK_CLUSTER_WT', None)
    for x, y, z in zip(a, b, c):
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(y, z)
