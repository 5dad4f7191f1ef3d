"""GPU parity, one launch at a time: every materialised tensor of a plan is recomputed by the CPU oracle FROM THE
GPU'S OWN INPUT TENSORS of that launch (read back bit-exactly), so the comparison isolates each kernel from the
accumulated fp16 drift of the layers before it.  What is left is fp32 accumulation order, i.e. at most a 1-ulp
fp16 flip on a small fraction of elements:
    |gpu - ref| <= 2^-9 * |ref| + 1e-3 * rms(ref)      for every element   (1 fp16 ulp is <= 2^-10 * |ref|)
    and fewer than 2 % of the elements differ at all; fp32 network outputs: <= 1e-4*|ref| + 1e-4*rms(ref).
Covers: stem (u8 + normalise LUT), fused dw3x3+pw1x1 (all four tile configs), standalone dw, implicit-GEMM
1x1/3x3 incl. split-K, upsample+concat loader, residual Add epilogue, stride-2 top/left pad, max pools, fp32 heads."""
import os

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns

pytestmark = pytest.mark.gpu


def _layerwise(spec, w, B, fuse=True, splitk=True, seed=0):
    import torch
    from k210_yolo_framework_amd import engine
    os.environ['YK_FUSE_DWPW'] = '1' if fuse else '0'
    os.environ['YK_SPLITK'] = '1' if splitk else '0'
    frames = np.random.default_rng(seed).integers(0, 256, (B, *spec.in_hw, 3), dtype=np.uint8)
    plan = engine.Plan(spec, w, max_batch=B, precision='f16')
    plan.run_u8(torch.from_numpy(frames).cuda())
    torch.cuda.synchronize()
    gpu = {0: oracle.normalise_u8(frames)}
    for op in spec.ops:
        if op['type'] in (ns.OP_UPSAMPLE, ns.OP_CONCAT):
            continue
        try:
            gpu[op['out']] = plan.read_tensor(op['out'], B)
        except engine.YkError:
            pass                                   # fused away: lives only in LDS / registers
    names = [l[0] for l in plan.launches()]
    plan.close()
    cp = spec.compile_plan(w)
    producer = {op['out']: i for i, op in enumerate(spec.ops)}
    checked, worst = 0, 0.0
    for i, op in enumerate(spec.ops):
        t = op['out']
        if t not in gpu or op['type'] in (ns.OP_UPSAMPLE, ns.OP_CONCAT):
            continue
        rows, inputs = [], {}

        def need(tid):
            if tid in gpu and tid != t:
                inputs[tid] = gpu[tid]
                return
            j = producer[tid]
            o = spec.ops[j]
            need(o['in0'])
            if o['in1'] >= 0:
                need(o['in1'])
            if j not in rows:
                rows.append(j)
        need(t)
        ref = oracle.net_forward_ex(cp, inputs, sorted(rows), [t], emulate_f16=True)[0]
        got = gpu[t]
        rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
        err = np.abs(got - ref)
        is_out = t in spec.outputs                  # network outputs are fp32: no fp16 rounding to flip
        bound = (1e-4 * np.abs(ref) + 1e-4 * rms) if is_out else (2.0 ** -9 * np.abs(ref) + 1e-3 * rms)
        bad = err > bound
        frac = 0.0 if is_out else float((got != ref).mean())
        assert not bad.any(), (f'op {i} {op.get("layer")} tensor {t} ({len(rows)} ops): {int(bad.sum())} elements beyond 1 ulp; '
                               f'max err {float(err.max()):.4g}, rms {rms:.4g}')
        assert frac < 0.02, f'op {i} {op.get("layer")}: {frac:.3%} of elements differ'
        worst = max(worst, float((err / np.maximum(bound, 1e-30)).max()))
        checked += 1
    return checked, names


@pytest.mark.parametrize('fuse,splitk', [(True, True), (False, False)])
def test_yolo_mobilev1_headline_layers(fuse, splitk):
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    n, names = _layerwise(spec, spec.init_weights(seed=1), 2, fuse, splitk)
    assert n >= (19 if fuse else 32)
    assert any('splitk' in x for x in names) == splitk


def test_yolo_mobilev1_batch32_layers_use_every_fused_config():
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    n, names = _layerwise(spec, spec.init_weights(seed=1), 32)
    for cfg in ('lr_', 'wide_'):
        assert any(cfg in x for x in names), (cfg, names)


@pytest.mark.parametrize('alpha', [1.0, 0.5])
def test_yolo_mobilev1_other_widths(alpha):
    spec = ns.yolo_mobilev1((96, 128, 3), 3, 20, alpha=alpha)
    _layerwise(spec, spec.init_weights(seed=2), 3)


@pytest.mark.parametrize('fuse', [True, False])
def test_yolo_mobilev2_layers(fuse):
    spec = ns.yolo_mobilev2((224, 320, 3), 3, 20, alpha=1.0)     # K=124 padding, ReLU6, linear bottleneck + Add
    n, names = _layerwise(spec, spec.init_weights(seed=1), 2, fuse)
    assert any('+add' in x for x in names)


def test_tiny_yolo_layers():
    spec = ns.tiny_yolo((224, 320, 3), 3, 20)                    # max pools incl. the stride-1 'same' one
    _layerwise(spec, spec.init_weights(seed=1), 2)
    spec = ns.tiny_yolo((416, 416, 3), 3, 20)                    # BASELINE config 3 shape (13x13 / 26x26)
    _layerwise(spec, spec.init_weights(seed=1), 1)


def test_darknet53_layers():
    spec = ns.yolo((128, 160, 3), 3, 20)                         # stride-2 (1,0) pad, 23 residual Adds, 3 scales
    w = spec.init_weights(seed=1)
    for k in w:
        if k.endswith('/gamma'):
            w[k] = (w[k] * 0.5).astype(np.float32)               # keep random-init activations inside fp16 range
    n, names = _layerwise(spec, w, 2)
    assert sum('+add' in x for x in names) == 23
    assert sum('+upcat' in x for x in names) == 2


def test_darknet53_416_fp16_stress_config_layers():
    """BASELINE configs[4]: full YOLO (Darknet-53) 416x416, 3 scales x 3 anchors, fp16 storage — every one of the 75 convs
    at the real shape, one launch at a time (the oracle runs on the host cores of the GPU box)."""
    spec = ns.yolo((416, 416, 3), 3, 20)
    assert spec.out_hw() == [(13, 13), (26, 26), (52, 52)]
    w = spec.init_weights(seed=1)
    for k in w:
        if k.endswith('/gamma'):
            w[k] = (w[k] * 0.5).astype(np.float32)
    n, names = _layerwise(spec, w, 1)
    assert n >= 75
