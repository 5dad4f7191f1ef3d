"""A detection head as ONE launch in the launch-per-layer schedule of the f16x2 mode (csrc/yk_xfin.h; yolonet.py:27-29, 35-38:
Conv3x3 + BN + LeakyReLU -> network-output Conv1x1): K slices reduced by the last arriving workgroup, the 75-channel conv from the
accumulator tile.  Same network outputs as the three-launch form (split-K conv, finishing pass, 1x1 conv) and as the fp32 oracle,
for any batch size, independent of the batch, bit-identical from run to run."""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = pytest.mark.gpu


def _outs(spec, w, frames, fused, max_batch=None, runs=1):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_FUSE_HEAD'] = '1' if fused else '0'
    try:
        plan = engine.Plan(spec, w, max_batch=max_batch or len(frames), precision='f16x2', schedule='throughput')
    finally:
        os.environ.pop('YK_FUSE_HEAD', None)
    names = [l[0] for l in plan.launches()]
    x = torch.from_numpy(frames).cuda()
    outs = []
    for _ in range(runs):
        plan.run_u8(x)
        plan.check()
        outs.append([o[:len(frames)].cpu().numpy().copy() for o in plan.outputs()])
    plan.close()
    return (outs[0] if runs == 1 else outs), names


@pytest.mark.parametrize('B', [1, 3, 32, 37])
def test_fused_heads_match_the_three_launch_form_and_the_oracle(B):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(B).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
    got, names = _outs(spec, w, frames, True)
    fused = [n for n in names if '+conv1x1_' in n and n.startswith('x:conv3x3')]
    assert len(fused) == 2, names                                         # y1 (768 -> 192 -> 75) and y2 (512 -> 128 -> 75)
    assert not any(n.endswith('to75[64x64,ring2]') for n in names), names
    ref, names0 = _outs(spec, w, frames, False)
    assert len(names0) == len(names) + 2 and not any('+conv1x1_' in n and n.startswith('x:conv3x3') for n in names0)
    for g, r in zip(got, ref):
        assert np.isfinite(g).all()
        assert np.abs(g - r).max() <= 2e-5 * np.abs(r).max()            # two f16x2 evaluations: only the K-slice boundaries differ
    nb = min(B, 4)
    ref32 = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:nb]), emulate_f16=False, out_ids=spec.outputs)
    for g, r in zip(got, ref32):
        assert np.abs(g[:nb] - r).max() <= 1e-4 * np.abs(r).max()


def test_fused_heads_are_per_image_and_bit_identical_from_run_to_run():
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=2)
    f = np.random.default_rng(1).integers(0, 256, (9, 224, 320, 3), dtype=np.uint8)
    f[4] //= 20
    (a, a2, a3), _ = _outs(spec, w, f, True, runs=3)
    for x, y, z in zip(a, a2, a3):                                        # whichever workgroup arrives last, the slices are added in z order
        np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(x, z)
    b, _ = _outs(spec, w, np.ascontiguousarray(f[::-1]), True, max_batch=16)
    c, _ = _outs(spec, w, f[4:5].copy(), True, max_batch=2)
    for x, y, z in zip(a, b, c):
        np.testing.assert_array_equal(x, y[::-1])
        np.testing.assert_array_equal(x[4:5], z)


def test_other_networks_and_sizes_agree_with_their_unfused_plans():
    """Whatever the plan fuses for a network / image size (128- and 192-channel heads: the MobileNet networks), outputs equal the
    YK_FUSE_HEAD=0 plan's; tiny_yolo / Darknet-53 heads (256 - 1024 channels) stay separate launches."""
    taken = 0
    for name, shape, alpha in (('yolo_mobilev1', (64, 96, 3), 0.75), ('yolo_mobilev1', (128, 160, 3), 1.0), ('yolo_mobilev1', (96, 64, 3), 0.5),
                               ('yolo_mobilev2', (224, 320, 3), 1.0), ('yolo_mobilev2', (96, 128, 3), 0.5), ('tiny_yolo', (416, 416, 3), 1.0),
                               ('yolo', (96, 128, 3), 1.0)):
        spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
        w = spec.init_weights(seed=1)
        f = np.random.default_rng(0).integers(0, 256, (3, *shape), dtype=np.uint8)
        a, na = _outs(spec, w, f, True)
        b, nb = _outs(spec, w, f, False)
        n_f = sum('+conv1x1_' in n and n.startswith('x:conv') for n in na)
        taken += n_f
        assert len(nb) == len(na) + n_f, (name, na, nb)                        # a launch entry = conv (+ its finishing pass); the 1x1 conv is the second
        tol = 5e-4 if name == 'yolo_mobilev2' else 1e-4                   # (tests/test_gpu_net.py: undamped MobileNet-v2 amplifies rounding noise)
        for x, y in zip(a, b):
            assert np.isfinite(x).all(), name
            assert np.abs(x - y).max() <= tol * np.abs(y).max(), name
    assert taken >= 6


@pytest.mark.skipif('dev' not in os.path.basename(os.environ.get('YK_LIB_PATH', '')),
                    reason='needs the slice-count override of the developer build (YK_LIB_PATH=.../libyolo_hip_dev.so)')
def test_equal_slice_counts_give_bit_identical_logits():
    """The fused head keeps the unfused path's arithmetic order per output element (k-steps ascending, three products per step, slices in z order,
    the same storage exponent for the tensor between the two convs): with the three-launch form's slice counts (7 and 4 at 32 images) its
    logits are BIT-identical to it."""
    import torch
    from k210_yolo_framework_amd import engine
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(0).integers(0, 256, (32, 224, 320, 3), dtype=np.uint8)

    def run(env):
        os.environ.update(env)
        try:
            plan = engine.Plan(spec, w, max_batch=32, precision='f16x2', schedule='throughput')
        finally:
            for k in env:
                os.environ.pop(k, None)
        plan.run_u8(torch.from_numpy(frames).cuda())
        plan.check()
        o = [x.cpu().numpy().copy() for x in plan.outputs()]
        plan.close()
        return o
    a = run({'YK_XF_SPLITK_192': '7', 'YK_XF_SPLITK_128': '4'})
    b = run({'YK_FUSE_HEAD': '0'})
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
