"""the K210 kmodel-v3 reader (k210_yolo_framework_amd/kmodel.py) and the only REAL-WEIGHTS known answer in the
reference tree: yolo3_frame_test_public/kfpkg/kpu_yolov3.kfpkg -> yolo.kmodel on the demo picture (aiimg.h = kfpkg/dog.jpg resized).

Three layers of evidence, all on CPU (the GPU run is tests/test_gpu_kmodel.py):
  * the container parses into the graph of models/yolonet.py:27-43 (32 KPU convs + the main-memory head ops);
  * oracle/kpu_ref.py (the KPU integer pipeline) on the demo picture + the reference's C region layer at main.c's thresholds gives
    the bicycle and the car of asset/k210_res.jpg (the regression fixture pins the boxes);
  * the dequantised float network agrees with the integer pipeline to rounding on the first layer, to 8-bit noise down the backbone,
    and - run through the fp32 oracle - finds the dog, the bicycle and the car of asset/dog_res.jpg (README.md:121-128).
"""
from pathlib import Path

import numpy as np
import pytest

import oracle
from k210_yolo_framework_amd import kmodel, netspec as ns
from oracle import kpu_ref

GOLD = Path(__file__).resolve().parent / 'golden'
VOC = ['aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog', 'horse', 'motorbike',
       'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


@pytest.fixture(scope='module')
def km():
    return kmodel.parse((GOLD / 'yolo.kmodel').read_bytes())


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD / 'kmodel_dog_golden.npz')


def test_fixture_is_the_references_file():
    ref = Path('/root/reference/yolo3_frame_test_public/kfpkg/kpu_yolov3.kfpkg')
    if not ref.exists():
        pytest.skip('reference tree not present (GPU box)')
    assert kmodel.read_kfpkg(ref) == (GOLD / 'yolo.kmodel').read_bytes()


def test_container_parses_into_the_yolo_mobilev1_graph(km):
    assert km.version == 3 and len(km.layers) == 39 and len(km.outputs) == 2
    assert [s for _, s in km.outputs] == [75 * 7 * 10 * 4, 75 * 14 * 20 * 4]
    convs = km.convs
    assert len(convs) == 32 == len(kmodel.YOLO_MOBILEV1_ORDER)
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    for c, layer in zip(convs, spec.layers):
        k = layer.kernel_shape
        assert c.ksize == k[0] and c.depthwise == (layer.kind == 'dwconv'), layer.name
        assert (c.in_ch, c.out_ch) == ((k[2], k[2]) if layer.kind == 'dwconv' else (k[2], k[3])), layer.name
    # stride 2 is a stride-1 conv + "left-top" 2x2 subsampling on the KPU; the depthwise convs hand it to the pointwise conv behind them
    assert [c.pool_type for c in convs].count(kmodel.POOL_LEFT_TOP_2_S2) == 5
    assert (convs[0].in_h, convs[0].in_w) == (224, 320)                        # the network tensor is 224 x 320
    types = [l.type for l in km.layers if isinstance(l, kmodel.MemLayer)]
    assert types == [12, 23, 13, 13, 17, 10243, 12]


def test_bad_files_are_refused():
    with pytest.raises(kmodel.KmodelError):
        kmodel.parse(b'\x04\x00\x00\x00' + b'\x00' * 60)                       # version 4
    with pytest.raises(kmodel.KmodelError):
        kmodel.parse(b'\x03\x00')
    data = bytearray((GOLD / 'yolo.kmodel').read_bytes()[:5000])
    with pytest.raises(kmodel.KmodelError):
        kmodel.parse(bytes(data))                                                # truncated bodies


def test_integer_pipeline_on_the_demo_picture_gives_the_k210_result(km, gold):
    outs = kpu_ref.run(km, gold['image'])
    np.testing.assert_array_equal(outs[0], gold['y1_q'])
    np.testing.assert_array_equal(outs[1], gold['y2_q'])
    dets = []
    for li, (W, H) in enumerate([(10, 7), (20, 14)]):
        x = outs[li].reshape(3, 25, H, W).astype(np.float32)
        _, bx, pr = oracle.region_run(x, gold['anchors'][li], W, H, 3, 20, 0.6, 0.3)          # main.c:280-288
        dets.append(oracle.region_draw(bx, pr, 0.6).reshape(-1, 6))
    dets = np.concatenate(dets, 0)
    np.testing.assert_array_equal(dets, gold['dets'])
    assert sorted(VOC[int(c)] for c in dets[:, 4]) == ['bicycle', 'car']       # asset/k210_res.jpg: the large bicycle box, the car top right
    car = dets[dets[:, 4] == 6][0]
    assert 170 < car[0] < 200 and 15 < car[1] < 40 and 290 < car[2] <= 320 and 55 < car[3] < 85


def test_dequantised_network_follows_the_integer_pipeline_and_finds_dog_bicycle_car(km, gold):
    w, rep = kmodel.to_float_weights(km)
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    assert set(w) == set(spec.init_weights(seed=0))
    # the activation tables are LeakyReLU(0.3) after conv1 / pointwise, ReLU after depthwise, LeakyReLU(0.1) in the head
    # (keras_mobilenet.py:356,428,436; yolonet.py:260), linear on the two outputs
    for name, r in rep['layers'].items():
        want = 1.0 if name in ('head_conv_2', 'head_conv_5') else 0.1 if name.startswith('head') else 0.0 if '_dw_' in name else 0.3
        assert abs(r['alpha'] - want) < 2e-3, (name, r['alpha'])
    keep = {}
    kpu_ref.run(km, gold['image'], keep)
    x = oracle.normalise_u8(gold['image'].transpose(1, 2, 0)[None].copy())
    plan = spec.compile_plan(w)
    # first layer: the float value and the 8-bit one differ by rounding only
    _, t1 = oracle.net_forward(plan, x, False, spec.outputs, dump_id=spec.ops[0]['out'])
    q1 = keep[0].astype(np.float64).transpose(1, 2, 0)[None] - rep['layers']['conv1']['y0']
    assert np.abs(t1 - q1).max() <= 0.5 + 5e-3          # + the floor of the two integer shifts (bn >> 15, table >> 20)
    # down the backbone: 8-bit noise (mean error in q units stays small; clipped entries excluded)
    last = [op for op in spec.ops if op.get('layer') == 'conv_pw_13'][0]['out']
    _, t27 = oracle.net_forward(plan, x, False, spec.outputs, dump_id=last)
    q27 = keep[26].astype(np.float64).transpose(1, 2, 0)[None]
    ok = (q27 > 0) & (q27 < 255)
    assert np.abs(t27 - (q27 - rep['layers']['conv_pw_13']['y0']))[ok].mean() < 5.0
    # the float network on the demo picture: dog (11), bicycle (1), car (6) - asset/dog_res.jpg
    outs = oracle.net_forward(plan, x, False, spec.outputs)
    found = {}
    for li, (W, H) in enumerate([(10, 7), (20, 14)]):
        xx = outs[li][0].transpose(2, 0, 1).reshape(3, 25, H, W).astype(np.float32).copy()
        _, bx, pr = oracle.region_run(xx, gold['anchors'][li], W, H, 3, 20, 0.5, 0.3)
        for r in oracle.region_draw(bx, pr, 0.5).reshape(-1, 6):
            found[VOC[int(r[4])]] = r[:4]
    assert set(found) == {'dog', 'bicycle', 'car'}, found
    dog = found['dog']
    assert 30 < dog[0] < 80 and 120 < dog[2] < 180 and dog[3] > 180          # lower left of the picture


def test_malformed_files_raise_kmodel_errors_not_struct_errors():
    """The parser trusts no offset of the file: truncations and wild offsets are KmodelErrors with the layer named."""
    import struct
    from k210_yolo_framework_amd import kmodel
    data = (GOLD / 'yolo.kmodel').read_bytes() if 'GOLD' in globals() else open('tests/golden/yolo.kmodel', 'rb').read()
    km = kmodel.parse(data)
    for cut in (10, 27, 60, 400, len(data) // 2, len(data) - 5):
        with pytest.raises(kmodel.KmodelError):
            kmodel.to_float_weights(kmodel.parse(data[:cut]))
    # wild offsets inside the first conv layer's body: register block, weights, BatchNorm, activation table
    ver, fl, arch, nl, ms, mm, nout = struct.unpack_from('<7I', data, 0)
    pos = 28 + 8 * nout + 8 * nl
    hdrs = [struct.unpack_from('<2I', data, 28 + 8 * nout + 8 * i) for i in range(nl)]
    for ty, sz in hdrs:
        if ty == kmodel.KL_K210_CONV:
            break
        pos += sz
    for field in (2, 3, 4, 5):                       # layer_offset, weights_offset, bn_offset, act_offset of kpu_model_conv_layer_argument_t
        bad = bytearray(data)
        struct.pack_into('<I', bad, pos + 4 * field, 0xFFFFFF00)
        with pytest.raises(kmodel.KmodelError):
            kmodel.parse(bytes(bad))
    # a network with the same number of convs but another shape must not be dequantised into yolo_mobilev1's names
    c = km.convs[3]
    c.out_ch += 8
    with pytest.raises(kmodel.KmodelError, match='yolo_mobilev1'):
        kmodel.to_float_weights(km)
