"""The fused DepthwiseConv2D + BN + act -> Conv2D 1x1 + BN + act block with its two phases on different waves (csrc/yk_xwblock.h;
keras_mobilenet.py:359-436 blocks 7-11 of yolo_mobilev1-0.75): same arithmetic in the same order as the one-role kernel, so the network
outputs are BIT-IDENTICAL to the YK_XB_WS=0 plan's, for any batch size, from run to run."""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif('dev' not in os.path.basename(os.environ.get('YK_LIB_PATH', '')),
                                 reason='the two-role block kernel is compiled into developer builds only (make -C k210_yolo_framework_amd/csrc dev; '
                                        'YK_LIB_PATH=.../libyolo_hip_dev.so): measured faster alone, slower with four batches in flight')]


def _outs(spec, w, frames, ws, max_batch=None, want=()):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_XB_WS'] = '1' if ws else '0'
    try:
        plan = engine.Plan(spec, w, max_batch=max_batch or len(frames), precision='f16x2', schedule='throughput')
    finally:
        os.environ.pop('YK_XB_WS', None)
    names = [l[0] for l in plan.launches()]
    plan.run_u8(torch.from_numpy(frames).cuda())
    plan.check()
    outs = [o[:len(frames)].cpu().numpy().copy() for o in plan.outputs()]
    mids = [plan.read_tensor(t, len(frames)) for t in want]
    plan.close()
    return outs, names, mids


@pytest.mark.parametrize('B', [1, 5, 32])
def test_two_role_blocks_are_bit_identical_to_the_one_role_kernel(B):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(B).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
    # the x1 tap (conv_pw_11: the last 14x20x384 block, stored split, read by the y2 head and by block 12) is compared as a tensor too
    x1 = [op['out'] for op in spec.ops if op.get('layer') == 'conv_pw_11']
    got, names, gm = _outs(spec, w, frames, True, want=x1)
    assert sum('2roles' in n for n in names) == 5, names
    ref, names0, rm = _outs(spec, w, frames, False, want=x1)
    assert not any('2roles' in n for n in names0) and len(names0) == len(names)
    for g, r in zip(got + gm, ref + rm):
        assert np.isfinite(g).all()
        np.testing.assert_array_equal(g, r)
    nb = min(B, 2)
    ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:nb]), emulate_f16=False, out_ids=spec.outputs)
    for g, r in zip(got, ref32):
        assert np.abs(g[:nb] - r).max() <= 1e-4 * np.abs(r).max()


def test_two_role_blocks_other_shapes_and_reruns():
    """alpha = 1.0 (512-channel blocks at 14x20: 16 steps to 512 outputs - not a <4,6> tile, stays one-role) and a small image; reruns of the
    two-role plan are bit-identical."""
    for shape, alpha in (((224, 320, 3), 0.75), ((128, 160, 3), 0.75), ((224, 320, 3), 1.0)):
        spec = ns.yolo_mobilev1(shape, 3, 20, alpha=alpha)
        w = spec.init_weights(seed=2)
        f = np.random.default_rng(1).integers(0, 256, (3, *shape), dtype=np.uint8)
        a, na, _ = _outs(spec, w, f, True)
        a2, _, _ = _outs(spec, w, f, True)
        b, _, _ = _outs(spec, w, f, False)
        for x, y, z in zip(a, a2, b):
            np.testing.assert_array_equal(x, y)
            np.testing.assert_array_equal(x, z)
