"""GPU parity: the conv stack (libyolo_hip.so via yk_plan_create / yk_run_*) vs oracle/yolo_net_ref.c.

Two comparisons, tolerances written out:
  * vs the oracle in fp16-storage emulation (same rounding points as the HIP engine): catches kernel
    bugs; differences come only from fp32 accumulation order (+ rare 1-ulp fp16 flips that follow).
      |y - ref| <= 1.5e-2 * max|ref| on every checked tensor, network outputs included (measured:
      1e-3..3e-3 for yolo_mobilev1; once a single fp16 rounding flips, the perturbation is re-rounded by
      every later layer, so deep tensors drift to the fp16 noise floor even against the emulation).
      The kernel-level (one layer at a time, exact inputs) comparison lives in test_gpu_layers.py.
  * vs the fp32 oracle (what the reference's Keras path computes): the north-star tolerance is on the
    DECODED quantities — scores and image-relative box coords within 1e-3 — checked in
    test_gpu_e2e.py; here the raw logits are bounded at 3e-2 * max|ref| (fp16 storage drift).
"""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = pytest.mark.gpu

TOL_EMU = 1.5e-2
TOL_F32 = 3e-2


def _run_plan(spec, w, x_u8=None, x_f32=None, fuse=True, want=(), precision='f16', splitk=True):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_FUSE_DWPW'] = '1' if fuse else '0'
    os.environ['YK_SPLITK'] = '1' if splitk else '0'
    B = (x_u8 if x_u8 is not None else x_f32).shape[0]
    plan = engine.Plan(spec, w, max_batch=B, precision=precision)
    if x_u8 is not None:
        plan.run_u8(torch.from_numpy(x_u8).cuda())
    else:
        plan.run_f32(torch.from_numpy(x_f32).cuda())
    torch.cuda.synchronize()
    outs = [o[:B].cpu().numpy() for o in plan.outputs()]
    mids = {}
    for t in want:
        try:
            mids[t] = plan.read_tensor(t, B)
        except engine.YkError:
            pass   # fused away
    names = [l[0] for l in plan.launches()]
    plan.close()
    return outs, mids, names


def _check(name, got, ref, tol):
    scale = max(float(np.abs(ref).max()), 1e-3)
    err = float(np.abs(got - ref).max())
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert np.isfinite(got).all(), name
    assert err <= tol * scale, f'{name}: max err {err:.4g} vs scale {scale:.4g} (tol {tol})'


@pytest.mark.parametrize('fuse', [False, True])
@pytest.mark.parametrize('name,shape,alpha,B', [
    ('yolo_mobilev1', (224, 320, 3), 0.75, 2), ('yolo_mobilev1', (64, 96, 3), 1.0, 3),
    ('yolo_mobilev1', (96, 64, 3), 0.5, 1), ('yolo_mobilev2', (64, 96, 3), 1.0, 2),
    ('yolo_mobilev2', (224, 320, 3), 0.75, 1)])
def test_mobilenets_u8_vs_oracle(name, shape, alpha, B, fuse):
    spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = spec.init_weights(seed=1)
    if name == 'yolo_mobilev2':
        # SURVEY 8(d) random init drives every ReLU6 of the 17 inverted-residual blocks into saturation; the
        # net then amplifies ANY perturbation ~1.5x per block (fp32-vs-fp16 oracles diverge the same way, see
        # DESIGN.md "drift").  Damped BN gains give a regime where a whole-network tolerance means something;
        # the undamped weights are covered kernel by kernel in test_gpu_layers.py.
        for k in w:
            if k.endswith('/gamma'):
                w[k] = (w[k] * 0.4).astype(np.float32)
    frames = np.random.default_rng(0).integers(0, 256, (B, *shape), dtype=np.uint8)
    x = oracle.normalise_u8(frames)
    every = [op['out'] for op in spec.ops if op['type'] in (ns.OP_CONV, ns.OP_DWCONV, ns.OP_ADD)]
    outs, mids, names = _run_plan(spec, w, x_u8=frames, fuse=fuse, want=every)
    assert any('dw3x3+' in n for n in names) == fuse
    plan = spec.compile_plan(w)
    for t, got in mids.items():
        if t in spec.outputs:
            continue
        _, ref = oracle.net_forward(plan, x, emulate_f16=True, out_ids=spec.outputs, dump_id=t)
        _check(f'{name} tensor {t}', got, ref, TOL_EMU)
    ref16 = oracle.net_forward(plan, x, emulate_f16=True, out_ids=spec.outputs)
    ref32 = oracle.net_forward(plan, x, emulate_f16=False, out_ids=spec.outputs)
    for i, (g, r16, r32) in enumerate(zip(outs, ref16, ref32)):
        _check(f'{name} y{i + 1} (fp16-emulating oracle)', g, r16, TOL_EMU)
        _check(f'{name} y{i + 1} (fp32 oracle)', g, r32, TOL_F32)


@pytest.mark.parametrize('name,shape,B', [('tiny_yolo', (224, 320, 3), 2), ('tiny_yolo', (96, 96, 3), 1),
                                          ('tiny_yolo', (416, 416, 3), 1), ('yolo', (64, 64, 3), 2), ('yolo', (96, 128, 3), 1)])
def test_darknets_f32_input_vs_oracle(name, shape, B):
    spec = ns.NETWORKS[name](shape, 3, 20)
    w = spec.init_weights(seed=2)
    if name == 'yolo':   # 23 residual Adds: damp BN gains so random-init activations stay inside fp16 range
        for k in w:
            if k.endswith('/gamma'):
                w[k] = (w[k] * 0.5).astype(np.float32)
    x = oracle.normalise_u8(np.random.default_rng(1).integers(0, 256, (B, *shape), dtype=np.uint8))
    every = [op['out'] for op in spec.ops if op['type'] in (ns.OP_CONV, ns.OP_MAXPOOL, ns.OP_ADD)][::3]
    outs, mids, _ = _run_plan(spec, w, x_f32=x, want=every)
    plan = spec.compile_plan(w)
    for t, got in mids.items():
        if t in spec.outputs:
            continue
        _, ref = oracle.net_forward(plan, x, emulate_f16=True, out_ids=spec.outputs, dump_id=t)
        _check(f'{name} tensor {t}', got, ref, TOL_EMU)
    ref16 = oracle.net_forward(plan, x, emulate_f16=True, out_ids=spec.outputs)
    for i, (g, r16) in enumerate(zip(outs, ref16)):
        _check(f'{name} y{i + 1}', g, r16, TOL_EMU)


def test_u8_and_f32_entry_points_agree():
    spec = ns.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=4)
    frames = np.random.default_rng(2).integers(0, 200, (2, 64, 96, 3), dtype=np.uint8)   # max < 255: LUT matters
    stem = spec.ops[0]['out']
    a, ma, _ = _run_plan(spec, w, x_u8=frames, want=[stem])
    b, mb, _ = _run_plan(spec, w, x_f32=oracle.normalise_u8(frames), want=[stem])
    # u8 frames take the MFMA stem, fp32 images the VALU stem: same fp16 inputs and weights, different fp32 summation order.
    # The stem outputs may therefore differ by one fp16 ulp in a few elements ...
    sa, sb = ma[stem].astype(np.float32), mb[stem].astype(np.float32)
    ulp = np.maximum(np.abs(sb), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(sa - sb) <= ulp).all() and (sa != sb).mean() < 0.02, ((sa != sb).mean(), np.abs(sa - sb).max())
    # ... and the network outputs agree like any two evaluations under the fp16 storage rule
    for x, y in zip(a, b):
        _check('u8 vs f32 entry', x, y, TOL_EMU)


def test_batch_smaller_than_max_batch_and_rerun():
    import torch
    from k210_yolo_framework_amd import engine
    spec = ns.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=4)
    rng = np.random.default_rng(3)
    plan = engine.Plan(spec, w, max_batch=8, precision='f16')
    f1 = rng.integers(0, 256, (8, 64, 96, 3), dtype=np.uint8)
    plan.run_u8(torch.from_numpy(f1).cuda())
    torch.cuda.synchronize()
    full = [o.cpu().numpy().copy() for o in plan.outputs()]
    plan.run_u8(torch.from_numpy(f1[:3].copy()).cuda())          # ragged: 3 of 8
    torch.cuda.synchronize()
    part = [o[:3].cpu().numpy() for o in plan.outputs()]
    for a, b in zip(full, part):
        np.testing.assert_array_equal(a[:3], b)                    # image i never depends on its batch mates
    plan.close()


# ---- precision 'f16x2': whole networks with the UNDAMPED SURVEY 8(d) weights vs the FP32 oracle -----------------------------
TOL_X2 = 1e-4          # of max|ref| per tensor; measured 2e-6 (mobilev1, tiny, Darknet)
TOL_X2_V2 = 5e-4       # undamped MobileNet-v2 amplifies ANY perturbation ~1.5x per inverted-residual block (17 blocks: x1000), fp32
#                        summation-order noise included - two fp32 implementations differ by this much; measured 1.1e-4


@pytest.mark.parametrize('name,shape,alpha,B,u8,fuse', [
    ('yolo_mobilev1', (224, 320, 3), 0.75, 2, True, True), ('yolo_mobilev1', (96, 64, 3), 0.5, 3, True, True),
    ('yolo_mobilev1', (224, 320, 3), 0.75, 2, True, False),     # YK_FUSE_DWPW=0 YK_SPLITK=0: stem / depthwise / 1x1 as separate launches
    ('yolo_mobilev2', (224, 320, 3), 1.0, 2, True, True), ('yolo_mobilev2', (64, 96, 3), 0.75, 1, True, True),
    ('yolo_mobilev2', (64, 96, 3), 0.75, 2, True, False),
    ('tiny_yolo', (416, 416, 3), 1.0, 1, False, True), ('yolo', (96, 128, 3), 1.0, 2, False, True),
    ('yolo', (416, 416, 3), 1.0, 1, False, True)])
def test_f16x2_whole_network_undamped_vs_fp32_oracle(name, shape, alpha, B, u8, fuse):
    """No gamma damping anywhere: MobileNet-v2 saturates its ReLU6s, Darknet-53's 23 residual adds drive activations to ~1e6 (past the
    fp16 range - the f16 plan returns inf here); the compensated operands with per-image exponents follow the fp32 path regardless."""
    spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(0).integers(0, 256, (B, *shape), dtype=np.uint8)
    x = oracle.normalise_u8(frames)
    every = [op['out'] for op in spec.ops if op['type'] in (ns.OP_CONV, ns.OP_DWCONV, ns.OP_ADD, ns.OP_MAXPOOL)]
    pick = every[::max(1, len(every) // 6)]     # each picked tensor costs one pass of the CPU oracle
    outs, mids, names = _run_plan(spec, w, x_u8=frames if u8 else None, x_f32=None if u8 else x, want=pick, precision='f16x2',
                                  fuse=fuse, splitk=fuse)
    assert all(n.startswith('x:') or n == 'u8_max' for n in names)
    assert fuse or not any('+conv' in n or '+dw' in n or 'splitk' in n for n in names), names
    tol = TOL_X2_V2 if name == 'yolo_mobilev2' else TOL_X2
    plan = spec.compile_plan(w)
    for t, got in mids.items():
        if t in spec.outputs:
            continue
        _, ref = oracle.net_forward(plan, x, emulate_f16=False, out_ids=spec.outputs, dump_id=t)
        _check(f'{name} tensor {t}', got, ref, tol)
    ref32 = oracle.net_forward(plan, x, emulate_f16=False, out_ids=spec.outputs)
    for i, (g, r) in enumerate(zip(outs, ref32)):
        _check(f'{name} y{i + 1} (fp32 oracle, f16x2 plan)', g, r, tol)


def test_f16x2_is_per_image_and_deterministic():
    """The operand exponents are taken per image: an image's outputs do not depend on its batch mates, reruns are bit-identical."""
    spec = ns.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=4)
    rng = np.random.default_rng(3)
    f = rng.integers(0, 256, (6, 64, 96, 3), dtype=np.uint8)
    f[2] //= 16                                                  # one dark image: its tensors have a much smaller max
    a, _, _ = _run_plan(spec, w, x_u8=f, precision='f16x2')
    b, _, _ = _run_plan(spec, w, x_u8=np.ascontiguousarray(f[::-1]), precision='f16x2')
    c, _, _ = _run_plan(spec, w, x_u8=f[2:3].copy(), precision='f16x2')
    for x, y, z in zip(a, b, c):
        np.testing.assert_array_equal(x, y[::-1])
        np.testing.assert_array_equal(x[2:3], z)
