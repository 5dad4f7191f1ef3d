"""Golden Keras-layout HDF5 files written by the REAL libhdf5 (h5py), to pin k210_yolo_framework_amd/h5lite.py + keras_io.py.

Run with an interpreter that has h5py (here: /opt/conda/bin/python3.9, h5py 3.3.0 / HDF5 1.10.6):
    /opt/conda/bin/python3.9 tests/golden/make_h5_golden.py
It writes, for a small detector built from the same layer kinds and NAMES the reference's Keras models have,
  keras_mini_weights.h5   what `model.save_weights()` produces: root attrs layer_names/backend/keras_version, one group per
                          layer with attr weight_names and datasets <layer>/<weight>:0 (save_weights_to_hdf5_group layout);
                          contiguous float32 datasets, fixed-length byte-string attributes (h5py 2.x style)
  keras_mini_model.h5     what `keras.models.save_model()` produces: the same tree under /model_weights, a JSON model_config
                          attribute, variable-length string attributes (h5py 3.x style), the two head convs with 255 output
                          channels (COCO) so that loading into a 20-class model exercises the cut of yolonet.py:146-156,
                          and two datasets stored chunked + gzip + shuffle
  keras_mini_expected.npz the arrays, keyed by this repo's weight names, for the 20-class model
Only h5py calls and seeded random data: nothing of the reference's source."""
import json
import sys
from pathlib import Path

import h5py
import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mini_net import mini_spec                                # noqa: E402  (tests/mini_net.py, pure numpy)
from k210_yolo_framework_amd.keras_io import keras_bn_name, BN_KEYS   # noqa: E402


def keras_layers(spec, w, first_auto):
    """[(keras layer name, [(weight name, array)])] in creation order, auto-named layers numbered from `first_auto`."""
    out, nc, nb = [], first_auto, first_auto

    def auto(prefix, i):
        return prefix if i == 0 else f'{prefix}_{i}'
    for l in spec.layers:
        fixed = not l.name.startswith('head_conv_')
        cname = l.name if fixed else auto('conv2d', nc)
        nc += 0 if fixed else 1
        ws = [(f'{cname}/{"depthwise_kernel" if l.kind == "dwconv" else "kernel"}:0', w[l.name + '/kernel'])]
        if l.use_bias:
            ws.append((f'{cname}/bias:0', w[l.name + '/bias']))
        out.append((cname, ws))
        if l.bn_name:
            bname = keras_bn_name(l) if fixed else auto('batch_normalization', nb)
            nb += 0 if fixed else 1
            out.append((bname, [(f'{bname}/{k}:0', w[f'{l.bn_name}/{k}']) for k in BN_KEYS]))
        if l.name == 'conv_pw_1':
            out.append(('conv_pw_1_relu', []))           # weight-less layers are listed too
    return out


def write_group(g, layers, vlen, compressed=()):
    names = [n for n, _ in layers]
    g.attrs['layer_names'] = names if vlen else np.array([n.encode() for n in names])
    g.attrs['backend'] = 'tensorflow' if vlen else np.bytes_(b'tensorflow')
    g.attrs['keras_version'] = '2.2.4-tf' if vlen else np.bytes_(b'2.2.4-tf')
    for n, ws in layers:
        lg = g.create_group(n)
        wn = [k for k, _ in ws]
        lg.attrs['weight_names'] = wn if (vlen and wn) else np.array([k.encode() for k in wn], dtype='S' if wn else 'S1')
        for k, a in ws:
            if k in compressed:
                lg.create_dataset(k, data=a, chunks=tuple(max(1, d // 2) for d in a.shape), compression='gzip', shuffle=True)
            else:
                d = lg.create_dataset(k, a.shape, dtype=a.dtype)
                d[...] = a


here = Path(__file__).resolve().parent
spec20, spec80 = mini_spec(20), mini_spec(80)
w80 = spec80.init_weights(seed=5)
w20 = {k: v.copy() for k, v in w80.items()}
for l in spec20.layers:                                     # the 20-class model's view of the COCO file: leading 75 of 255 channels
    if l.use_bias:
        w20[l.name + '/kernel'] = w80[l.name + '/kernel'][..., :75]
        w20[l.name + '/bias'] = w80[l.name + '/bias'][:75]
with h5py.File(here / 'keras_mini_weights.h5', 'w') as f:
    write_group(f, keras_layers(spec20, w20, first_auto=0), vlen=False)
with h5py.File(here / 'keras_mini_model.h5', 'w') as f:
    f.attrs['model_config'] = json.dumps({'class_name': 'Model', 'config': {'name': 'mini', 'layers': len(spec80.layers)}})
    f.attrs['keras_version'] = '2.2.4-tf'
    f.attrs['backend'] = 'tensorflow'
    write_group(f.create_group('model_weights'), keras_layers(spec80, w80, first_auto=17), vlen=True,
                compressed=('conv_pw_2/kernel:0', 'conv1_bn/gamma:0'))
np.savez(here / 'keras_mini_expected.npz', **w20)
print('wrote keras_mini_weights.h5, keras_mini_model.h5, keras_mini_expected.npz')
