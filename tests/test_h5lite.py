"""h5lite (dependency-free HDF5 subset) and keras_io (Keras checkpoint <-> NetSpec weights), SURVEY 8(f) row N1.

Pinned against the real libhdf5: tests/golden/keras_mini_*.h5 were written by h5py 3.3 / HDF5 1.10.6 (generator
tests/golden/make_h5_golden.py) in the layout Keras' save_weights / save_model produce; where an interpreter with h5py exists
(/opt/conda/bin/python3.9 in the build container) the files written by h5lite are also read back by the real library."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from k210_yolo_framework_amd import h5lite, keras_io, netspec as ns

sys.path.insert(0, str(Path(__file__).resolve().parent))
from mini_net import mini_spec as _mini   # noqa: E402
H5PY_PYTHON = '/opt/conda/bin/python3.9'


def test_reader_on_real_hdf5_weight_file(golden_dir):
    f = h5lite.File(golden_dir / 'keras_mini_weights.h5')
    assert f.version == 0                                               # h5py default: superblock v0, symbol-table groups
    names = [n.decode() for n in f.attrs['layer_names']]
    assert names[:4] == ['conv1', 'conv1_bn', 'conv_dw_1', 'conv_dw_1_bn'] and 'conv_pw_1_relu' in names
    assert f.attrs['backend'] == b'tensorflow' and f.attrs['keras_version'] == b'2.2.4-tf'
    assert [w.decode() for w in f['conv_dw_1'].attrs['weight_names']] == ['conv_dw_1/depthwise_kernel:0']
    assert len(f['conv_pw_1_relu'].attrs['weight_names']) == 0
    exp = np.load(golden_dir / 'keras_mini_expected.npz')
    k = f['conv1/conv1/kernel:0']
    assert k.shape == (3, 3, 3, 8) and k.dtype == np.float32
    np.testing.assert_array_equal(k.read(), exp['conv1/kernel'])
    np.testing.assert_array_equal(f['conv_dw_2/conv_dw_2/depthwise_kernel:0'].read(), exp['conv_dw_2/kernel'])
    np.testing.assert_array_equal(f['conv1_bn']['conv1_bn/moving_variance:0'][...], exp['conv1_bn/moving_variance'])
    assert sum(1 for _ in f.visit_datasets()) == 10 + 2 + 8 * 4      # kernels + head biases + 8 BatchNorms x 4
    with pytest.raises(KeyError):
        f['conv1/nope']


def test_reader_on_real_full_model_file_vlen_chunked_gzip(golden_dir):
    f = h5lite.File(golden_dir / 'keras_mini_model.h5')
    assert isinstance(f.attrs['model_config'], str) and '"class_name": "Model"' in f.attrs['model_config']   # vlen string
    mw = f['model_weights']
    assert list(mw.attrs['layer_names'][:2]) == ['conv1', 'conv1_bn']
    spec80 = _mini(80)
    w80 = spec80.init_weights(seed=5)
    np.testing.assert_array_equal(mw['conv_pw_2/conv_pw_2/kernel:0'].read(), w80['conv_pw_2/kernel'])      # chunked+gzip+shuffle
    np.testing.assert_array_equal(mw['conv1_bn/conv1_bn/gamma:0'].read(), w80['conv1_bn/gamma'])
    assert mw['conv2d_18/conv2d_18/kernel:0'].shape == (1, 1, 24, 255)


def test_keras_io_loads_by_name_and_creation_order(golden_dir):
    spec = _mini(20)
    exp = dict(np.load(golden_dir / 'keras_mini_expected.npz'))
    w, rep = keras_io.load_keras_weights(spec, golden_dir / 'keras_mini_weights.h5')
    assert set(w) == set(exp)
    for k in exp:
        np.testing.assert_array_equal(w[k], exp[k], err_msg=k)
    assert 'conv_dw_1_bn' in rep['by_name'] and ('head_conv_1', 'conv2d') in rep['by_order'] and not rep['cut']
    assert ('head_conv_5', 'conv2d_4') in rep['by_order'] and ('head_conv_4_bn', 'batch_normalization_2') in rep['by_order']


def test_keras_io_head_surgery_255_to_75(golden_dir):
    """models/yolonet.py:146-156,182-189: a COCO (80-class) checkpoint loaded into a 20-class model keeps the leading channels."""
    spec = _mini(20)
    exp = dict(np.load(golden_dir / 'keras_mini_expected.npz'))
    w, rep = keras_io.load_keras_weights(spec, golden_dir / 'keras_mini_model.h5')   # auto names start at conv2d_17 here
    for k in exp:
        np.testing.assert_array_equal(w[k], exp[k], err_msg=k)
    cut = {(c[1], c[2]) for c in rep['cut']}
    assert ((1, 1, 24, 255), (1, 1, 24, 75)) in cut and ((255,), (75,)) in cut and len(rep['cut']) == 4
    with pytest.raises(ValueError):                      # the other way round (file smaller than the model) must not pass silently
        keras_io.load_keras_weights(_mini(100), golden_dir / 'keras_mini_weights.h5')


def test_writer_roundtrip_all_four_networks(tmp_path):
    for name, shape, alpha in (('yolo_mobilev1', (64, 96, 3), 0.5), ('yolo_mobilev2', (64, 96, 3), 0.5), ('tiny_yolo', (64, 64, 3), 1.0)):
        spec = ns.NETWORKS[name](shape, 3, 2, alpha=alpha)
        w = spec.init_weights(seed=3)
        p = tmp_path / f'{name}.h5'
        keras_io.save_keras_weights(spec, w, p)
        back, rep = keras_io.load_keras_weights(spec, p)
        assert set(back) == set(w) and not rep['cut'] and not rep['unused'] and not rep['missing']
        for k in w:
            np.testing.assert_array_equal(back[k], w[k], err_msg=f'{name} {k}')
        f = h5lite.File(p)
        names = [n.decode() for n in f.attrs['layer_names']]
        if name == 'yolo_mobilev2':
            assert {'bn_Conv1', 'expanded_conv_depthwise_BN', 'block_16_project_BN', 'Conv_1_bn'} <= set(names)   # keras_mobilenet_v2.py names
        if name == 'tiny_yolo':
            assert names[:4] == ['conv2d', 'batch_normalization', 'conv2d_1', 'batch_normalization_1']           # tf.keras 1.14 auto names


def test_partial_backbone_file_by_name(tmp_path):
    """models/yolonet.py:16-21: base_model.load_weights('data/mobilenet_v1_base_7.h5') covers only the backbone."""
    spec = ns.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.5)
    w = spec.init_weights(seed=9)
    tree, order = {}, []
    for l in spec.layers:
        if l.name.startswith('head_conv'):
            continue
        kn = 'depthwise_kernel:0' if l.kind == 'dwconv' else 'kernel:0'
        tree[l.name] = ({l.name: {kn: w[l.name + '/kernel']}}, {'weight_names': np.array([f'{l.name}/{kn}'.encode()])})
        bn = keras_io.keras_bn_name(l)
        tree[bn] = ({bn: {f'{k}:0': w[f'{l.bn_name}/{k}'] for k in keras_io.BN_KEYS}},
                    {'weight_names': np.array([f'{bn}/{k}:0'.encode() for k in keras_io.BN_KEYS])})
        order += [l.name, bn]
    p = tmp_path / 'base.h5'
    h5lite.write(p, tree, {'layer_names': np.array([n.encode() for n in order])})
    base = spec.init_weights(seed=1)
    with pytest.raises(ValueError):
        keras_io.load_keras_weights(spec, p, base=base, strict=True)
    got, rep = keras_io.load_keras_weights(spec, p, base=base, strict=False)
    assert rep['missing'] and all(m.startswith('head_conv') for m in rep['missing'])
    np.testing.assert_array_equal(got['conv_pw_13/kernel'], w['conv_pw_13/kernel'])
    np.testing.assert_array_equal(got['head_conv_1/kernel'], base['head_conv_1/kernel'])


def test_not_hdf5_and_truncated_files_fail_loudly(tmp_path, golden_dir):
    with pytest.raises(h5lite.H5Error):
        h5lite.File(b'PK\x03\x04' + b'\0' * 100)
    data = (golden_dir / 'keras_mini_weights.h5').read_bytes()
    with pytest.raises((h5lite.H5Error, ValueError, KeyError)):
        f = h5lite.File(data[:len(data) // 3])
        for _, d in f.visit_datasets():
            d.read()


@pytest.mark.skipif(not os.path.exists(H5PY_PYTHON), reason='no interpreter with h5py here (build container only)')
def test_files_written_by_h5lite_are_read_by_the_real_libhdf5(tmp_path):
    spec = ns.tiny_yolo((64, 64, 3), 3, 2)
    w = spec.init_weights(seed=4)
    p = tmp_path / 'w.h5'
    keras_io.save_keras_weights(spec, w, p)
    np.save(tmp_path / 'k.npy', w['conv2d_3/kernel'])
    code = f"""
import h5py, numpy as np
f = h5py.File(r'{p}', 'r')
names = [n.decode() for n in f.attrs['layer_names']]
assert names[:2] == ['conv2d', 'batch_normalization'], names
k = f['conv2d_2/conv2d_2/kernel:0'][...]
assert np.array_equal(k, np.load(r'{tmp_path / 'k.npy'}')), 'kernel differs'
n = []
f.visititems(lambda name, o: n.append(o[...].size) if isinstance(o, h5py.Dataset) else None)
assert len(n) == {sum(1 + (1 if l.use_bias else 0) + (4 if l.bn_name else 0) for l in spec.layers)}, len(n)
assert [w.decode() for w in f['batch_normalization_1'].attrs['weight_names']][0] == 'batch_normalization_1/gamma:0'
print('h5py-ok')
"""
    r = subprocess.run([H5PY_PYTHON, '-W', 'ignore', '-c', code], capture_output=True, text=True)
    assert 'h5py-ok' in r.stdout, r.stderr[-2000:]


# ---- the full-model file of keras_train.py:105-109 ------------------------------------------------------------------------------------
def _walk_shapes(cfg: dict):
    """Shape inference over a Keras functional `model_config` (channels_last): {layer name: (h, w, c)}."""
    out = {}
    for l in cfg['config']['layers']:
        c, cls = l['config'], l['class_name']
        src = [out[n[0]] for n in l['inbound_nodes'][0]] if l['inbound_nodes'] else []
        if cls == 'InputLayer':
            out[l['name']] = tuple(c['batch_input_shape'][1:])
            continue
        h, w, ch = src[0]
        if cls == 'ZeroPadding2D':
            (pt, pb), (pl, pr) = c['padding']
            out[l['name']] = (h + pt + pb, w + pl + pr, ch)
        elif cls in ('Conv2D', 'DepthwiseConv2D', 'MaxPooling2D'):
            k = c.get('kernel_size', c.get('pool_size'))[0]
            st = c['strides'][0]
            if c.get('padding', 'valid') == 'same':
                ho, wo = -(-h // st), -(-w // st)
            else:
                ho, wo = (h - k) // st + 1, (w - k) // st + 1
            out[l['name']] = (ho, wo, c['filters'] if cls == 'Conv2D' else ch)
        elif cls == 'UpSampling2D':
            out[l['name']] = (2 * h, 2 * w, ch)
        elif cls == 'Concatenate':
            assert all(s[:2] == src[0][:2] for s in src)
            out[l['name']] = (h, w, sum(s[2] for s in src))
        elif cls == 'Add':
            assert all(s == src[0] for s in src)
            out[l['name']] = src[0]
        elif cls in ('BatchNormalization', 'LeakyReLU', 'ReLU'):
            out[l['name']] = src[0]
        else:
            raise AssertionError(cls)
    return out


@pytest.mark.parametrize('name,shape,alpha', [('yolo_mobilev1', (224, 320, 3), 0.75), ('yolo_mobilev2', (224, 320, 3), 1.0),
                                              ('tiny_yolo', (416, 416, 3), 1.0), ('yolo', (416, 416, 3), 1.0)])
def test_save_model_layout_round_trip(tmp_path, name, shape, alpha):
    """keras.models.save_model's file: `model_config` + `/model_weights` (keras_train.py:105-109).  The weights come back bit for bit
    through the nested layout, and the generated architecture is a consistent Keras functional graph whose shapes are the network's."""
    from k210_yolo_framework_amd import keras_io, netspec as ns
    spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = spec.init_weights(seed=3)
    path = tmp_path / 'yolo_model.h5'
    keras_io.save_keras_model(spec, w, path)
    f = h5lite.File(path)
    assert 'model_weights' in f and 'layer_names' not in f.attrs and 'layer_names' in f['model_weights'].attrs
    assert bytes(np.asarray(f.attrs['keras_version']).item()).startswith(b'2.2.4')
    got, rep = keras_io.load_keras_weights(spec, path)
    assert set(got) == set(w) and all(np.array_equal(got[k], w[k]) for k in w)
    cfg = keras_io.read_model_config(path)
    assert cfg['class_name'] == 'Model'
    layers = cfg['config']['layers']
    names = [l['name'] for l in layers]
    assert len(set(names)) == len(names)                                          # Keras requires unique layer names
    seen = set()
    for l in layers:                                                              # topological order, every inbound layer defined before use
        for node in l['inbound_nodes']:
            for src, ni, ti, kw in node:
                assert src in seen and ni == 0 and ti == 0 and kw == {}
        seen.add(l['name'])
    file_layers = [n.decode() if isinstance(n, bytes) else str(n) for n in np.asarray(f['model_weights'].attrs['layer_names']).ravel()]
    weighted = [l['name'] for l in layers if l['class_name'] in ('Conv2D', 'DepthwiseConv2D', 'BatchNormalization')]
    assert sorted(weighted) == sorted(file_layers)                               # every weighted layer of the graph has its group, and vice versa
    shapes = _walk_shapes(cfg)
    outs = [shapes[o[0]] for o in cfg['config']['output_layers']]
    assert outs == [spec.tensors[t] for t in spec.outputs]
    n_conv = sum(1 for l in layers if l['class_name'] in ('Conv2D', 'DepthwiseConv2D'))
    assert n_conv == spec.conv_layer_count()
    if name == 'yolo_mobilev1':                                                   # spot checks against models/keras_mobilenet.py:340-436
        by = {l['name']: l for l in layers}
        assert by['conv1']['config']['padding'] == 'valid' and by['conv1_pad']['config']['padding'] == [[1, 1], [1, 1]]     # :343
        assert by['conv_dw_1']['class_name'] == 'DepthwiseConv2D' and by['conv_dw_1']['config']['padding'] == 'same'
        assert by['conv_pw_1_leaky']['config']['alpha'] == pytest.approx(0.3) and 'max_value' not in by['conv_dw_1_relu']['config']
        assert by['conv1_bn']['config']['epsilon'] == 0.001


def test_plugin_save_writes_the_full_model_file(tmp_path):
    from k210_yolo_framework_amd import keras_io, yolonet
    m, _ = yolonet.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.5)
    m.save(str(tmp_path / 'm.h5'))
    assert keras_io.read_model_config(tmp_path / 'm.h5')['config']['layers'][0]['config']['batch_input_shape'] == [None, 64, 96, 3]
    m.save_weights(str(tmp_path / 'w.h5'))
    assert keras_io.read_model_config(tmp_path / 'w.h5') is None
    m2, _ = yolonet.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.5)
    m2.load_weights(str(tmp_path / 'm.h5'))
    assert all(np.array_equal(m2.get_weights()[k], v) for k, v in m.get_weights().items())
