"""Host-side mirror of the reference's model plugin API (`models/yolonet.py`).

The reference selects a network by name (`eval(model_def)`, keras_inference.py:77-78, keras_train.py:49-50)
and calls it as  net([H,W,3], anchor_num, class_num, alpha=...) -> (yolo_model, yolo_model_warpper)  where
`yolo_model` emits [N,h,w,A*(5+C)] per scale and `yolo_model_warpper` the same data reshaped to
[N,h,w,A,5+C] (yolonet.py:40-44).  The same four names are registered here; the objects returned expose
the methods the reference scripts use — `load_weights`, `predict`, `save_weights` — with numpy in / list of
numpy out and the reference's shapes and (h, w, anchor, entry) order.

All arithmetic happens in libyolo_hip.so (engine.Plan); without a GPU `predict` raises.  The models default to the 'f16x2'
precision mode (fp16 MFMA with compensated operands: the fp32-class results a Keras user expects, scores within 1e-3 and the same
boxes); pass precision='f16' (or set `model.precision`) for the throughput mode used by bench.py.

Weights: Keras HDF5 files (`.h5`, read and written by keras_io / h5lite without h5py: the reference's checkpoint format,
including the 255->A*(5+C) head cut of yolonet.py:146-156,182-189) or a flat `.npz` with Keras-layout arrays
(`<layer>/kernel` HWIO, `<layer>/bias`, `<bn>/gamma|beta|moving_mean|moving_variance`).
Unlike the reference (yolonet.py:16-21,146,182) building a model does NOT require pre-train files: it
starts from seeded random weights until `load_weights` is called; `load_weights(path, by_name=True)` loads a backbone-only
pre-train file the way `base_model.load_weights('data/mobilenet_v1_base_7.h5')` does.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import netspec as ns


class YoloModel:
    """`wrapped=False` -> yolo_model ([N,h,w,A*(5+C)]), `wrapped=True` -> yolo_model_warpper ([N,h,w,A,5+C])."""

    def __init__(self, spec: ns.NetSpec, shared: dict, wrapped: bool):
        self.spec = spec
        self._s = shared          # weights + lazily built engine plan, shared by both views of one network
        self.wrapped = wrapped

    @property
    def precision(self) -> str:
        return self._s.get('precision', 'f16x2')

    @precision.setter
    def precision(self, value: str) -> None:
        if value != self.precision:
            self._s['precision'] = value
            self._drop_plan()

    # -- Keras-like surface -------------------------------------------------------------------------
    @property
    def input_shape(self):
        return (None, *self.spec.in_hw, 3)

    @property
    def output_shape(self):
        e = 5 + self.spec.class_num
        a = self.spec.anchor_num
        return [(None, h, w, a, e) if self.wrapped else (None, h, w, a * e) for (h, w) in self.spec.out_hw()]

    def get_weights(self) -> Dict[str, np.ndarray]:
        return self._s['weights']

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        want = self.spec.init_weights(seed=0)
        for k, v in want.items():
            if k not in weights:
                raise KeyError(f'missing weight {k}')
            if tuple(weights[k].shape) != tuple(v.shape):
                raise ValueError(f'{k}: shape {weights[k].shape} != {v.shape}')
        self._s['weights'] = {k: np.asarray(weights[k], np.float32) for k in want}
        self._drop_plan()

    def load_weights(self, path: str, by_name: bool = False) -> None:
        """keras_inference.py:80 / keras_train.py:52-57.  `.h5`/`.hdf5`: a Keras weight or full-model file; `.kmodel`/`.kfpkg`: the
        K210 demo's quantised yolo_mobilev1-0.75 (kmodel.py); otherwise `.npz`.
        by_name=True accepts a file that covers only part of the network (backbone pre-train files, yolonet.py:16-21)."""
        path = str(path)
        if path.endswith(('.h5', '.hdf5', '.keras')):
            from . import keras_io
            w, self.last_load_report = keras_io.load_keras_weights(self.spec, path, base=self._s['weights'], strict=not by_name)
            self.set_weights(w)
            return
        if path.endswith(('.kmodel', '.kfpkg')):
            # the K210 demo's 8-bit model (yolo3_frame_test_public/kfpkg/kpu_yolov3.kfpkg -> yolo.kmodel, main.c:57,213,274), dequantised
            from . import kmodel
            data = kmodel.read_kfpkg(path) if path.endswith('.kfpkg') else open(path, 'rb').read()
            w, self.last_load_report = kmodel.to_float_weights(kmodel.parse(data))
            self.set_weights(w)
            return
        with np.load(path) as z:
            have = {k: z[k] for k in z.files}
        if by_name:
            have = {**self._s['weights'], **have}
        self.set_weights(have)

    def save(self, path: str) -> None:
        """`keras.models.save_model(yolo_model, path)` (keras_train.py:105-109): the full-model HDF5 file - `model_config` (the
        architecture as Keras JSON) plus the weights under `/model_weights`."""
        from . import keras_io
        keras_io.save_keras_model(self.spec, self._s['weights'], str(path))

    def save_weights(self, path: str) -> None:
        """Weights only: `.h5` -> Keras `save_weights` layout (what the reference's `load_weights` reads), else `.npz`.  The full-model
        file of keras_train.py:105-109 is `save`."""
        path = str(path)
        if path.endswith(('.h5', '.hdf5')):
            from . import keras_io
            keras_io.save_keras_weights(self.spec, self._s['weights'], path)
        else:
            np.savez(path, **self._s['weights'])

    def _drop_plan(self):
        p = self._s.pop('plan', None)
        if p is not None:
            p.close()

    def _plan(self, batch: int):
        from . import engine
        p = self._s.get('plan')
        if p is None or p.max_batch < batch:
            self._drop_plan()
            p = engine.Plan(self.spec, self._s['weights'], max_batch=max(batch, 1), precision=self.precision)
            self._s['plan'] = p
        return p

    def predict(self, x: np.ndarray) -> List[np.ndarray]:
        """keras_inference.py:88.  x: [N,H,W,3] float (already `img / np.max(img)`) or uint8 frames
        (then the normalisation of tools/utils.py:405 is done on the GPU)."""
        import torch
        x = np.asarray(x)
        if x.ndim == 3:
            x = x[None]
        n = x.shape[0]
        plan = self._plan(n)
        if x.dtype == np.uint8:
            plan.run_u8(torch.from_numpy(np.ascontiguousarray(x)).cuda())
        else:
            plan.run_f32(torch.from_numpy(np.ascontiguousarray(x, np.float32)).cuda())
        torch.cuda.synchronize()
        outs = [o[:n].cpu().numpy() for o in plan.outputs()]
        if self.wrapped:
            e = 5 + self.spec.class_num
            outs = [o.reshape(n, o.shape[1], o.shape[2], self.spec.anchor_num, e) for o in outs]
        return outs


def _build(name: str, input_shape, anchor_num: int, class_num: int, **kwargs) -> Tuple[YoloModel, YoloModel]:
    alpha = kwargs.get('alpha', 1.0)
    spec = ns.NETWORKS[name](list(input_shape), anchor_num, class_num, alpha=alpha)
    shared = {'weights': spec.init_weights(seed=1), 'precision': kwargs.get('precision', 'f16x2')}
    return YoloModel(spec, shared, False), YoloModel(spec, shared, True)


def yolo_mobilev1(input_shape, anchor_num, class_num, **kwargs):
    """models/yolonet.py:12-46."""
    return _build('yolo_mobilev1', input_shape, anchor_num, class_num, **kwargs)


def yolo_mobilev2(input_shape, anchor_num, class_num, **kwargs):
    """models/yolonet.py:49-104."""
    return _build('yolo_mobilev2', input_shape, anchor_num, class_num, **kwargs)


def tiny_yolo(input_shape, anchor_num, class_num, **kwargs):
    """models/yolonet.py:107-158."""
    return _build('tiny_yolo', input_shape, anchor_num, class_num, **kwargs)


def yolo(input_shape, anchor_num, class_num, **kwargs):
    """models/yolonet.py:161-191."""
    return _build('yolo', input_shape, anchor_num, class_num, **kwargs)


# the reference uses eval(model_def); a dict keeps the same names without eval
MODEL_DEFS = {'yolo_mobilev1': yolo_mobilev1, 'yolo_mobilev2': yolo_mobilev2, 'tiny_yolo': tiny_yolo, 'yolo': yolo}
