"""Keras-HDF5 checkpoints <-> the flat weight dict of a NetSpec.

The reference moves weights around as Keras HDF5 files:
  * `yolo_model_warpper.load_weights(ckpt)`            keras_inference.py:80, keras_train.py:52-57
  * `keras.models.save_model(yolo_model, ckpt)`        keras_train.py:105-109
  * backbone pre-train files (`base_model.load_weights('data/mobilenet_v1_base_7.h5')`)      models/yolonet.py:16-21,55-62
  * COCO checkpoints whose 255-channel heads are cut down to A*(5+C) channels               models/yolonet.py:146-156,182-189

File layout (what `save_weights_to_hdf5_group` of Keras 2.x / tf.keras 1.14 writes, read here with h5lite):
  root (or group 'model_weights' of a full-model file)  attrs  layer_names = [b'conv1', b'conv1_bn', ...]   (model.layers order)
  group <layer>                                          attrs  weight_names = [b'conv1/kernel:0', ...]      (layer.weights order)
  dataset <layer>/<weight name>                          float32, Keras layouts: Conv2D HWIO, DepthwiseConv2D [3,3,C,1],
                                                         BatchNormalization gamma, beta, moving_mean, moving_variance

Assignment rule.  Keras loads topologically (layer i of the file -> layer i of the model).  The same pairing is rebuilt
here without a Keras graph: layers whose names the reference fixes in code (MobileNet: 'conv1', 'conv_dw_3_bn', 'bn_Conv1',
'block_7_project_BN', ...) are matched by name; the auto-named rest ('conv2d_17', 'batch_normalization_4': the YOLO heads and
all of Darknet) are matched per class in creation order (numeric suffix), which is the order models/yolonet.py creates them in
and the order netspec.py mirrors.  Every pairing is shape-checked; a file tensor that is larger than the model's along an axis
(COCO 255-channel heads) is cut with the reference's own rule `new[slices_of_min_shape] = old[slices_of_min_shape]`.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import h5lite
from . import netspec as ns

BN_KEYS = ('gamma', 'beta', 'moving_mean', 'moving_variance')


def keras_bn_name(layer: ns.Layer) -> Optional[str]:
    """The BatchNormalization layer name Keras has for this conv (keras_mobilenet.py:355,427,435; keras_mobilenet_v2.py:322,381,447,467,481)."""
    if not layer.bn_name:
        return None
    n = layer.name
    if n == 'Conv1':
        return 'bn_Conv1'
    if n == 'Conv_1':
        return 'Conv_1_bn'
    if n.startswith(('block_', 'expanded_conv_')):
        return n + '_BN'
    return n + '_bn'


def _s(x) -> str:
    return x.decode('utf8') if isinstance(x, (bytes, np.bytes_)) else str(x)


def _attr_list(attrs: dict, key: str) -> List[str]:
    """An attribute Keras may have split into `key0, key1, ...` chunks when it exceeded the 64 KiB header limit."""
    if key in attrs:
        return [_s(v) for v in np.asarray(attrs[key]).ravel()]
    out, i = [], 0
    while f'{key}{i}' in attrs:
        out += [_s(v) for v in np.asarray(attrs[f'{key}{i}']).ravel()]
        i += 1
    return out


def read_keras_h5(path) -> List[Tuple[str, List[Tuple[str, np.ndarray]]]]:
    """-> [(layer name, [(weight short name, array)])] in the file's layer order; layers without weights are dropped."""
    f = h5lite.File(path)
    root = f['model_weights'] if 'layer_names' not in f.attrs and 'model_weights' in f else f
    names = _attr_list(root.attrs, 'layer_names')
    if not names:
        raise h5lite.H5Error(f'{path}: no layer_names attribute - not a Keras weight file')
    out = []
    for ln in names:
        g = root[ln]
        ws = []
        for wn in _attr_list(g.attrs, 'weight_names'):
            short = wn.split('/')[-1].split(':')[0]
            ws.append((short, np.asarray(g[wn].read())))
        if ws:
            out.append((ln, ws))
    return out


def _creation_index(name: str, pos: int) -> Tuple[int, int]:
    if name.startswith(('conv2d', 'batch_normalization', 'depthwise_conv2d')):
        m = re.search(r'_(\d+)$', name)
        return (int(m.group(1)) if m else 0, pos)
    return (1 << 30, pos)                                # unknown naming scheme: keep file order, after the auto-named ones


def _fit(name: str, have: np.ndarray, want_shape: Tuple[int, ...], report: dict) -> np.ndarray:
    have = np.asarray(have)
    if have.shape == tuple(want_shape):
        return have.astype(np.float32)
    if have.ndim != len(want_shape) or any(h < w for h, w in zip(have.shape, want_shape)):
        raise ValueError(f'{name}: file tensor {have.shape} does not fit model tensor {tuple(want_shape)}')
    report['cut'].append((name, have.shape, tuple(want_shape)))
    return have[tuple(slice(0, w) for w in want_shape)].astype(np.float32)      # models/yolonet.py:146-156,182-189


def assign(spec: ns.NetSpec, file_layers, base: Optional[Dict[str, np.ndarray]] = None, strict: bool = True):
    """Pair file layers with the NetSpec's parameters.  -> (weights dict, report).  With strict=False a file that covers only
    part of the network (a backbone pre-train file) updates just those layers of `base`."""
    want = spec.init_weights(seed=0)
    weights = dict(base) if base is not None else {}
    report = {'by_name': [], 'by_order': [], 'cut': [], 'unused': [], 'missing': []}
    convs = [(l.name, l) for l in spec.layers]
    bns = [(keras_bn_name(l), l) for l in spec.layers if l.bn_name]
    fconv, fbn = {}, {}
    for pos, (ln, ws) in enumerate(file_layers):
        d = dict(ws)
        if 'kernel' in d or 'depthwise_kernel' in d:
            fconv[ln] = (pos, d)
        elif 'gamma' in d or 'moving_mean' in d:
            fbn[ln] = (pos, d)
        else:
            report['unused'].append(ln)

    def put_conv(l: ns.Layer, ln: str, d: dict):
        k = d.get('depthwise_kernel') if l.kind == 'dwconv' else d.get('kernel')
        if k is None:
            raise ValueError(f'{ln}: expected a {"depthwise_" if l.kind == "dwconv" else ""}kernel for {l.name}')
        weights[l.name + '/kernel'] = _fit(ln + '/kernel', k, want[l.name + '/kernel'].shape, report)
        if l.use_bias:
            if 'bias' not in d:
                raise ValueError(f'{ln}: no bias for {l.name}')
            weights[l.name + '/bias'] = _fit(ln + '/bias', d['bias'], want[l.name + '/bias'].shape, report)

    def put_bn(l: ns.Layer, ln: str, d: dict):
        for key in BN_KEYS:
            tgt = f'{l.bn_name}/{key}'
            if key not in d:
                if key == 'gamma':                        # BatchNormalization(scale=False)
                    weights[tgt] = np.ones(want[tgt].shape, np.float32)
                    continue
                if key == 'beta':                         # center=False
                    weights[tgt] = np.zeros(want[tgt].shape, np.float32)
                    continue
                raise ValueError(f'{ln}: BatchNormalization without {key}')
            weights[tgt] = _fit(f'{ln}/{key}', d[key], want[tgt].shape, report)

    def auto_named(l: ns.Layer) -> bool:               # layers Keras numbers itself: never matched by name (the counter
        return l.name.startswith(('head_conv_', 'conv2d_'))   # start differs between Keras flavours and between processes)

    todo_c, todo_b = [], []
    for name, l in convs:
        if not auto_named(l) and name in fconv:
            put_conv(l, name, fconv.pop(name)[1])
            report['by_name'].append(name)
        else:
            todo_c.append(l)
    for name, l in bns:
        if not auto_named(l) and name in fbn:
            put_bn(l, name, fbn.pop(name)[1])
            report['by_name'].append(name)
        else:
            todo_b.append(l)
    rest_c = sorted(fconv.items(), key=lambda kv: _creation_index(kv[0], kv[1][0]))
    rest_b = sorted(fbn.items(), key=lambda kv: _creation_index(kv[0], kv[1][0]))
    for l, (ln, (_, d)) in zip(todo_c, rest_c):
        put_conv(l, ln, d)
        report['by_order'].append((l.name, ln))
    for l, (ln, (_, d)) in zip(todo_b, rest_b):
        put_bn(l, ln, d)
        report['by_order'].append((l.bn_name, ln))
    report['unused'] += [ln for ln, _ in rest_c[len(todo_c):]] + [ln for ln, _ in rest_b[len(todo_b):]]
    report['missing'] = [l.name for l in todo_c[len(rest_c):]] + [l.bn_name for l in todo_b[len(rest_b):]]
    if strict and (report['missing'] or report['unused']):
        raise ValueError(f'checkpoint does not match {spec.name}: layers without file weights {report["missing"][:4]}, '
                         f'file layers without a place {report["unused"][:4]}')
    if strict:
        for k, v in want.items():
            if k not in weights:
                raise ValueError(f'{k} not set by the checkpoint')
    return weights, report


def load_keras_weights(spec: ns.NetSpec, path, base: Optional[Dict[str, np.ndarray]] = None, strict: bool = True):
    return assign(spec, read_keras_h5(path), base=base, strict=strict)


def save_keras_weights(spec: ns.NetSpec, weights: Dict[str, np.ndarray], path, tf_keras_names: bool = True) -> None:
    """Write `weights` in Keras' `save_weights` HDF5 layout (root attribute `layer_names`, one group per layer with `weight_names`).
    NOTE the format: the reference's keras_train.py:105-109 calls `keras.models.save_model`, whose file additionally carries the
    architecture (`model_config`) and nests the same tree under `/model_weights`; this writer produces the WEIGHTS-ONLY file - what
    `model.load_weights(path)` reads on the reference side (keras_inference.py:80 does exactly that), not what `load_model` /
    `TFLiteConverter.from_keras_model_file` (keras_freeze.py:15-23) need.  The reader (`read_keras_h5`) accepts both layouts.
    Layers are written in creation order; auto-named layers get tf.keras 1.14 names ('conv2d', 'conv2d_1', ...)."""
    tree, order = {}, []
    n_conv = n_bn = 0

    def auto(prefix: str, i: int) -> str:
        return prefix if (i == 0 and tf_keras_names) else f'{prefix}_{i if tf_keras_names else i + 1}'

    def fixed(name: str) -> bool:
        return not name.startswith(('head_conv_', 'conv2d_'))
    for l in spec.layers:
        if fixed(l.name):
            cname = l.name
        else:
            cname = auto('conv2d', n_conv)
            n_conv += 1
        ws = {('depthwise_kernel:0' if l.kind == 'dwconv' else 'kernel:0'): np.asarray(weights[l.name + '/kernel'], np.float32)}
        wn = [f'{cname}/{"depthwise_kernel" if l.kind == "dwconv" else "kernel"}:0']
        if l.use_bias:
            ws['bias:0'] = np.asarray(weights[l.name + '/bias'], np.float32)
            wn.append(f'{cname}/bias:0')
        tree[cname] = ({cname: ws}, {'weight_names': np.array([w.encode() for w in wn])})
        order.append(cname)
        if l.bn_name:
            if fixed(l.name):
                bname = keras_bn_name(l)
            else:
                bname = auto('batch_normalization', n_bn)
                n_bn += 1
            tree[bname] = ({bname: {f'{k}:0': np.asarray(weights[f'{l.bn_name}/{k}'], np.float32) for k in BN_KEYS}},
                           {'weight_names': np.array([f'{bname}/{k}:0'.encode() for k in BN_KEYS])})
            order.append(bname)
    h5lite.write(path, tree, {'layer_names': np.array([n.encode() for n in order]), 'backend': np.bytes_(b'tensorflow'),
                              'keras_version': np.bytes_(b'2.2.4-tf')})
