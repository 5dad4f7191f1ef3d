"""The conv-stack oracle (oracle/yolo_net_ref.c) vs an independent torch-CPU float64 build of the
same Keras layers (oracle/torch_net_ref.py).  Two implementations that share no code must agree."""
import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import netspec as ns
from oracle import torch_net_ref as torch_ref

SURVEY_TABLE = {  # convs, MMAC/img, act elems, weight elems
    ('yolo_mobilev1', (224, 320, 3), 0.75): (32, 732.39, 11351690, 3839376),
    ('tiny_yolo', (416, 416, 3), 1.0): (13, 2735.8, 8737807, 8707248),
    ('yolo_mobilev2', (224, 320, 3), 1.0): (57, 747.0, 17198090, 4654876),
    ('yolo', (416, 416, 3), 1.0): (75, 32714.0, 79049919, 61573216),
}


@pytest.mark.parametrize('key', list(SURVEY_TABLE))
def test_topology_matches_survey_table(key):
    name, shape, alpha = key
    s = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    convs, mmac, act, w = SURVEY_TABLE[key]
    assert s.conv_layer_count() == convs
    assert abs(s.macs_per_image() / 1e6 - mmac) < 0.05
    assert s.act_elems_per_image() == act
    assert s.weight_elems() == w


def test_output_shapes_reference_reshape():
    # yolonet.py:40-41 hard-codes Reshape((7,10,..)),(14,20,..) for 224x320
    for n in ('yolo_mobilev1', 'yolo_mobilev2', 'tiny_yolo'):
        s = ns.NETWORKS[n]((224, 320, 3), 3, 20, alpha=0.75 if 'mobile' in n else 1.0)
        assert s.out_hw() == [(7, 10), (14, 20)]
    assert ns.yolo((416, 416, 3), 3, 20).out_hw() == [(13, 13), (26, 26), (52, 52)]


def test_mobilenet_width_rules():
    s = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=1.0)
    assert s.layers[2].kernel_shape == (1, 1, 32, 40)          # block-1 width 40 when alpha==1 (keras_mobilenet.py:217)
    s = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    assert s.layers[2].kernel_shape == (1, 1, 24, 48)
    lay = {l.name: l for l in s.layers}
    assert lay['head_conv_1'].kernel_shape == (3, 3, 768, 192)  # 192 when alpha<=0.8 (yolonet.py:28)
    assert lay['head_conv_4'].kernel_shape == (3, 3, 128 + 384, 128)
    v2 = ns.yolo_mobilev2((224, 320, 3), 3, 20, alpha=1.0)
    lay = {l.name: l for l in v2.layers}
    assert lay['Conv1'].kernel_shape == (3, 3, 3, 32)
    assert lay['block_1_expand'].kernel_shape == (1, 1, 16, 48)     # keras_mobilenet_v2.py:329
    assert lay['block_2_expand'].kernel_shape == (1, 1, 24, 124)    # :331
    assert lay['Conv_1'].kernel_shape == (1, 1, 320, 1280)


@pytest.mark.parametrize('name,shape,alpha,batch', [
    ('yolo_mobilev1', (64, 96, 3), 0.75, 2), ('yolo_mobilev2', (64, 96, 3), 1.0, 2),
    ('tiny_yolo', (96, 64, 3), 1.0, 2), ('yolo', (64, 64, 3), 1.0, 1), ('yolo_mobilev1', (224, 320, 3), 0.75, 1),
])
def test_oracle_vs_torch(name, shape, alpha, batch):
    s = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = s.init_weights(seed=1)
    plan = s.compile_plan(w)
    rng = np.random.default_rng(0)
    x = oracle.normalise_u8(rng.integers(0, 256, (batch, *shape), dtype=np.uint8))
    # outputs + a few intermediates (every 7th op output)
    want = sorted(set(s.outputs) | {op['out'] for op in s.ops[::7]})
    got = oracle.net_forward(plan, x, out_ids=want)
    ref = torch_ref.forward(s, w, x, want)
    for t, g in zip(want, got):
        r = ref[t]
        scale = max(float(np.abs(r).max()), 1e-6)
        assert g.shape == r.shape
        assert float(np.abs(g - r).max()) / scale < 2e-5, (t, float(np.abs(g - r).max()), scale)


def test_f16_emulation_rounds_storage():
    s = ns.yolo_mobilev1((32, 32, 3), 3, 2, alpha=0.5)
    w = s.init_weights(seed=3)
    plan = s.compile_plan(w)
    x = oracle.normalise_u8(np.random.default_rng(1).integers(0, 256, (1, 32, 32, 3), dtype=np.uint8))
    mid = s.ops[4]['out']
    (o16,), d16 = oracle.net_forward(plan, x, emulate_f16=True, out_ids=[s.outputs[0]], dump_id=mid)
    (o32,), d32 = oracle.net_forward(plan, x, emulate_f16=False, out_ids=[s.outputs[0]], dump_id=mid)
    assert np.array_equal(d16, oracle.f16_round(d16))           # stored activations are fp16 values
    assert not np.array_equal(d32, oracle.f16_round(d32))
    assert np.abs(o16 - o32).max() < 0.05 * max(1.0, np.abs(o32).max())


def test_f16_round_matches_numpy():
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-6, 500), rng.normal(0, 3e4, 500),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8]]).astype(np.float32)
    L = oracle.lib()
    got = np.array([L.yk_ref_f16_round(float(a)) for a in v], np.float32)
    np.testing.assert_array_equal(got, oracle.f16_round(v))


def test_normalise_matches_numpy_float64_division():
    rng = np.random.default_rng(2)
    f = rng.integers(0, 200, (3, 8, 8, 3), dtype=np.uint8)
    want = np.stack([(im / np.max(im)).astype(np.float32) for im in f])   # tools/utils.py:405
    np.testing.assert_array_equal(oracle.normalise_u8(f), want)
