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


def _weights_tree(spec: ns.NetSpec, weights: Dict[str, np.ndarray], tf_keras_names: bool = True):
    """The `save_weights` tree: ({layer: ({layer: {weight: array}}, {'weight_names'})}, root attributes)."""
    tree, order = {}, []
    names = _file_layer_names(spec, tf_keras_names)
    for l in spec.layers:
        cname, bname = names[l.name]
        ws = {('depthwise_kernel:0' if l.kind == 'dwconv' else 'kernel:0'): np.asarray(weights[l.name + '/kernel'], np.float32)}
        wn = [f'{cname}/{"depthwise_kernel" if l.kind == "dwconv" else "kernel"}:0']
        if l.use_bias:
            ws['bias:0'] = np.asarray(weights[l.name + '/bias'], np.float32)
            wn.append(f'{cname}/bias:0')
        tree[cname] = ({cname: ws}, {'weight_names': np.array([w.encode() for w in wn])})
        order.append(cname)
        if bname:
            tree[bname] = ({bname: {f'{k}:0': np.asarray(weights[f'{l.bn_name}/{k}'], np.float32) for k in BN_KEYS}},
                           {'weight_names': np.array([f'{bname}/{k}:0'.encode() for k in BN_KEYS])})
            order.append(bname)
    return tree, {'layer_names': np.array([n.encode() for n in order]), 'backend': np.bytes_(b'tensorflow'),
                  'keras_version': np.bytes_(b'2.2.4-tf')}


def save_keras_weights(spec: ns.NetSpec, weights: Dict[str, np.ndarray], path, tf_keras_names: bool = True) -> None:
    """Write `weights` in Keras' `save_weights` HDF5 layout (root attribute `layer_names`, one group per layer with `weight_names`):
    the WEIGHTS-ONLY file, what `model.load_weights(path)` reads on the reference side (keras_inference.py:80, keras_train.py:52).
    The full-model file of keras_train.py:105-109 (`save_model`: the same tree under `/model_weights` plus `model_config`) is
    `save_keras_model`.  The reader (`read_keras_h5`) accepts both layouts.
    Layers are written in creation order; auto-named layers get tf.keras 1.14 names ('conv2d', 'conv2d_1', ...)."""
    tree, attrs = _weights_tree(spec, weights, tf_keras_names)
    h5lite.write(path, tree, attrs)


# ---- the full-model file of keras_train.py:105-109 (`keras.models.save_model(yolo_model, ckpt)`) -------------------------------------
def keras_model_config(spec: ns.NetSpec) -> dict:
    """The functional-model description Keras 2.2.4-tf stores in the `model_config` attribute of a `save_model` file, rebuilt from
    the op list: InputLayer, [ZeroPadding2D +] Conv2D / DepthwiseConv2D, BatchNormalization, LeakyReLU / ReLU, MaxPooling2D,
    UpSampling2D, Concatenate, Add, with every layer's inbound node.  Weighted layers carry the names `save_keras_weights` writes
    (creation-order `conv2d_N` / `batch_normalization_N` for the head, the backbone's own names elsewhere); helper layers get derived
    names (`<conv>_pad`, `<conv>_relu`).  Every layer config lists the constructor arguments that DIFFER from Keras' defaults
    (`from_config` is `cls(**config)`): the attribute has to fit HDF5's 64 KiB object-header message, which a fully spelled-out
    Darknet-53 (252 layers) does not."""
    layers = [{'name': 'input_1', 'class_name': 'InputLayer', 'inbound_nodes': [],
               'config': {'batch_input_shape': [None, int(spec.in_hw[0]), int(spec.in_hw[1]), 3], 'dtype': 'float32', 'name': 'input_1'}}]
    producer = {0: 'input_1'}                                   # tensor id -> name of the Keras layer whose output it is
    names = _file_layer_names(spec)
    counters = {}

    def fresh(prefix):
        i = counters.get(prefix, 0)
        counters[prefix] = i + 1
        return prefix if i == 0 else f'{prefix}_{i}'

    def add(name, cls, cfg, inbound):
        layers.append({'name': name, 'class_name': cls, 'config': {'name': name, **cfg}, 'inbound_nodes': [[[src, 0, 0, {}] for src in inbound]]})
        return name
    darknet = spec.name in ('yolo', 'tiny_yolo')
    bn_momentum = 0.999 if spec.name == 'yolo_mobilev2' else 0.99           # keras_mobilenet_v2.py:319; Keras default elsewhere
    for op in spec.ops:
        ty, src = op['type'], producer[op['in0']]
        co = spec.tensors[op['out']][2]
        if ty in (ns.OP_CONV, ns.OP_DWCONV):
            l = op['layer']
            cname, bname = names[l]
            k, st = op['k'], op['stride']
            pad = [[int(op['pad_t']), int(op['pad_b'])], [int(op['pad_l']), int(op['pad_r'])]]
            # the reference pads every strided conv explicitly (ZeroPadding2D + 'valid': keras_mobilenet.py:343,418, keras_mobilenet_v2.py:312,
            # 455, yolonet.py:197) and uses padding='same' at stride 1
            same = st == 1 and pad == [[(k - 1) // 2] * 2] * 2
            if not same and any(v for row in pad for v in row):
                src = add(f'{cname}_pad', 'ZeroPadding2D', {'padding': pad}, [src])
            layer = next(x for x in spec.layers if x.name == l)
            common = {'kernel_size': [k, k], 'strides': [st, st], 'padding': 'same' if same else 'valid', 'use_bias': bool(layer.use_bias)}
            if ty == ns.OP_CONV:
                cfg = {**common, 'filters': int(co)}
                if darknet:
                    cfg['kernel_regularizer'] = {'class_name': 'L1L2', 'config': {'l1': 0.0, 'l2': 0.0005}}      # yolonet.py:243
                src = add(cname, 'Conv2D', cfg, [src])
            else:
                src = add(cname, 'DepthwiseConv2D', common, [src])
            if bname:
                src = add(bname, 'BatchNormalization', {'axis': [3], 'momentum': bn_momentum, 'epsilon': 0.001}, [src])
            if op['act'] == ns.ACT_LEAKY:
                src = add(f'{cname}_leaky', 'LeakyReLU', {'alpha': float(np.float32(op['alpha']))}, [src])
            elif op['act'] in (ns.ACT_RELU, ns.ACT_RELU6):
                src = add(f'{cname}_relu', 'ReLU', {'max_value': 6.0} if op['act'] == ns.ACT_RELU6 else {}, [src])
        elif ty == ns.OP_MAXPOOL:
            src = add(fresh('max_pooling2d'), 'MaxPooling2D', {'pool_size': [2, 2], 'padding': 'same', 'strides': [op['stride']] * 2}, [src])
        elif ty == ns.OP_UPSAMPLE:
            src = add(fresh('up_sampling2d'), 'UpSampling2D', {'size': [2, 2]}, [src])
        elif ty == ns.OP_CONCAT:
            src = add(fresh('concatenate'), 'Concatenate', {'axis': -1}, [src, producer[op['in1']]])
        elif ty == ns.OP_ADD:
            src = add(fresh('add'), 'Add', {}, [src, producer[op['in1']]])
        else:
            raise ValueError(f'op type {ty}')
        producer[op['out']] = src
    return {'class_name': 'Model', 'config': {'name': 'model', 'layers': layers, 'input_layers': [['input_1', 0, 0]],
                                              'output_layers': [[producer[t], 0, 0] for t in spec.outputs]}}


def _file_layer_names(spec: ns.NetSpec, tf_keras_names: bool = True) -> Dict[str, Tuple[str, Optional[str]]]:
    """conv layer name of the spec -> (Keras conv layer name, Keras BatchNormalization layer name | None), as save_keras_weights numbers them."""
    out, n_conv, n_bn = {}, 0, 0

    def auto(prefix: str, i: int) -> str:
        return prefix if (i == 0 and tf_keras_names) else f'{prefix}_{i if tf_keras_names else i + 1}'
    for l in spec.layers:
        fixed = not l.name.startswith(('head_conv_', 'conv2d_'))
        if fixed:
            cname = l.name
        else:
            cname = auto('conv2d', n_conv)
            n_conv += 1
        bname = None
        if l.bn_name:
            if fixed:
                bname = keras_bn_name(l)
            else:
                bname = auto('batch_normalization', n_bn)
                n_bn += 1
        out[l.name] = (cname, bname)
    return out


def save_keras_model(spec: ns.NetSpec, weights: Dict[str, np.ndarray], path) -> None:
    """The file `keras.models.save_model(yolo_model, path)` writes (keras_train.py:105-109; what `load_model`,
    `TFLiteConverter.from_keras_model_file` (keras_freeze.py:15-23) and `load_weights` all read): root attributes `keras_version`,
    `backend`, `model_config` (JSON, keras_model_config) and the `save_weights` tree nested under `/model_weights`.  No optimizer
    state (`include_optimizer` has nothing to include: Adam's moments live in train.Trainer and are not part of the reference's
    inference hand-off)."""
    import json
    tree, attrs = _weights_tree(spec, weights)
    cfg = json.dumps(keras_model_config(spec), separators=(',', ':')).encode('utf8')
    h5lite.write(path, {'model_weights': (tree, attrs)},
                 {'keras_version': np.bytes_(b'2.2.4-tf'), 'backend': np.bytes_(b'tensorflow'), 'model_config': np.bytes_(cfg)})


def read_model_config(path) -> Optional[dict]:
    """`model_config` of a full-model file as a dict (None for a weights-only file)."""
    import json
    f = h5lite.File(path)
    raw = f.attrs.get('model_config')
    if raw is None:
        return None
    if isinstance(raw, np.ndarray):
        raw = raw.item() if raw.shape == () else raw.ravel()[0]
    return json.loads(raw.decode('utf8') if isinstance(raw, (bytes, np.bytes_)) else str(raw))
