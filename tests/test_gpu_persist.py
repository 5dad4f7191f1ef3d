"""The persistent late-backbone stage of the f16x2 mode (csrc/yk_xpersist.h; keras_mobilenet.py:359-436 blocks 7-13 as ONE launch of
per-image workgroup clusters): same network outputs as the launch-per-layer form and as the fp32 oracle, for any batch size, with
several plans in flight; a cluster that cannot assemble is reported, never waited for forever."""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = pytest.mark.gpu


def _outs(spec, w, frames, persist, max_batch=None):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_PERSIST'] = '1' if persist else '0'
    try:
        plan = engine.Plan(spec, w, max_batch=max_batch or len(frames), precision='f16x2')
    finally:
        os.environ.pop('YK_PERSIST', None)
    names = [l[0] for l in plan.launches()]
    plan.run_u8(torch.from_numpy(frames).cuda())
    plan.check()
    outs = [o[:len(frames)].cpu().numpy().copy() for o in plan.outputs()]
    plan.close()
    return outs, names


@pytest.mark.parametrize('B', [1, 3, 32, 40])
def test_persistent_stage_matches_plain_launches_and_the_oracle(B):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(B).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
    got, names = _outs(spec, w, frames, True)
    assert sum('x:persist' in n for n in names) == 1, names
    assert not any(n.startswith('x:dw3x3s1_384') or 'conv1x1s1_384to384' in n or 'conv1x1s1_768to768' in n for n in names), names
    ref, names0 = _outs(spec, w, frames, False)
    # without the stage: the five 14x20x384 blocks are fused dw+pw launches, the two 7x10 blocks two launches each: nine launches for one
    assert not any('x:persist' in n for n in names0) and len(names0) == len(names) + 8, names0
    assert sum('dw3x3s1+conv1x1_384to384' in n for n in names0) == 5, names0
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


def test_four_plans_in_flight_assemble_their_clusters():
    """bench.py's shape: four independent batches on four streams; every persistent launch needs its 8 workgroups per image co-resident."""
    import torch
    from k210_yolo_framework_amd import engine
    from k210_yolo_framework_amd.helper import VOC_ANCHORS
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    B = 32
    g = torch.Generator(device='cuda').manual_seed(3)
    frames = [torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=g) for _ in range(4)]
    one = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=1, schedule='latency')
    want = []
    for f in frames:
        d, c, _ = one.submit(f)
        one.wait()
        want.append((d.cpu().numpy().copy(), c.cpu().numpy().copy()))
    one.plans[0].check()
    one.close()
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=4, schedule='latency')     # (depth >= 2 would pick the per-layer launches)
    for rnd in range(25):
        got = [pipe.submit(f) for f in frames]
        pipe.wait()
        for (d, c, _), (wd, wc) in zip(got, want):
            c = c.cpu().numpy()
            assert np.array_equal(c, wc), rnd
            for b in range(0, B, 7):
                assert np.array_equal(d[b, :c[b]].cpu().numpy(), wd[b, :wc[b]])
    for p in pipe.plans:
        p.check()
    pipe.close()


def test_other_networks_keep_their_plain_launches_or_persist_consistently():
    """yolo_mobilev2 / tiny_yolo / Darknet have no dw -> 1x1 chain of the supported form at these sizes, or a short one: whatever the plan
    chooses, outputs equal the YK_PERSIST=0 plan's."""
    for name, shape, alpha in (('yolo_mobilev2', (224, 320, 3), 1.0), ('tiny_yolo', (416, 416, 3), 1.0), ('yolo_mobilev1', (96, 64, 3), 0.5),
                               ('yolo_mobilev1', (64, 96, 3), 0.75), ('yolo_mobilev1', (128, 160, 3), 1.0)):
        spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
        w = spec.init_weights(seed=1)
        f = np.random.default_rng(0).integers(0, 256, (2, *shape), dtype=np.uint8)
        a, na = _outs(spec, w, f, True)
        b, nb = _outs(spec, w, f, False)
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-4 * max(np.abs(y).max(), 1e-3), name
