"""train_ref.py — CPU ORACLE for the training step (test infrastructure, NOT the product path).

torch-CPU float64 autograd restatement of what `train_model.fit` computes per step in the reference:

  keras_train.py:73-76      Adam(lr, decay) over the summed per-layer losses
  tools/utils.py:708-793    create_loss_fn (ignore mask without gradient, utils.py:704)
  models/yolonet.py:245-260 DarknetConv2D: l2(5e-4) kernel regulariser; BatchNormalization in training mode
  models/keras_mobilenet.py / keras_mobilenet_v2.py backbones (no regulariser)

PARITY UNPINNED against the reference itself: TensorFlow 1.14 is not installable here and the reference holds no
test or golden vector for its training step.  This file is the independent arbiter instead — it shares no code
with csrc/yk_train.hip (autograd derives every gradient the HIP kernels implement by hand).

The floating-point kernels are compared with this float64 reference within the tolerances written in
tests/test_gpu_train.py."""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from k210_yolo_framework_amd import netspec as ns

L2_WEIGHT = 5e-4


def _is_darknet_conv(name):
    return name.startswith('head_conv') or name.startswith('conv2d_')


def forward_train(spec: ns.NetSpec, params: Dict[str, torch.Tensor], x_nhwc: torch.Tensor, stats: dict = None):
    """Training-mode forward (batch statistics, biased variance).  params: Keras-layout float64 leaf tensors."""
    lay = {l.name: l for l in spec.layers}
    T = {0: x_nhwc.permute(0, 3, 1, 2)}
    for idx, op in enumerate(spec.ops):
        x = T[op['in0']]
        t = op['type']
        if t in (ns.OP_CONV, ns.OP_DWCONV):
            l = lay[op['layer']]
            k = params[l.name + '/kernel']
            ho, wo, _ = spec.tensors[op['out']]
            kk, st = op['k'], op['stride']
            pb = (ho - 1) * st + kk - x.shape[2] - op['pad_t']
            pr = (wo - 1) * st + kk - x.shape[3] - op['pad_l']
            xp = F.pad(x, (op['pad_l'], max(pr, 0), op['pad_t'], max(pb, 0)))
            if t == ns.OP_CONV:
                y = F.conv2d(xp, k.permute(3, 2, 0, 1), params.get(l.name + '/bias') if l.use_bias else None, stride=st)
            else:
                y = F.conv2d(xp, k.permute(2, 3, 0, 1), None, stride=st, groups=x.shape[1])
            y = y[:, :, :ho, :wo]
            if l.bn_name:
                mu = y.mean((0, 2, 3), keepdim=True)
                var = ((y - mu) ** 2).mean((0, 2, 3), keepdim=True)
                if stats is not None:
                    stats[l.bn_name] = (mu.detach().flatten().numpy(), var.detach().flatten().numpy())
                y = (y - mu) / torch.sqrt(var + ns.BN_EPS) * params[l.bn_name + '/gamma'].view(1, -1, 1, 1) \
                    + params[l.bn_name + '/beta'].view(1, -1, 1, 1)
                if stats is not None and stats.get('__want_pre__'):
                    stats[l.name + '/pre'] = y.detach().permute(0, 2, 3, 1).numpy()
            a = op['act']
            if a == ns.ACT_RELU:
                y = F.relu(y)
            elif a == ns.ACT_RELU6:
                y = torch.clamp(y, 0, 6)
            elif a == ns.ACT_LEAKY:
                y = F.leaky_relu(y, op['alpha'])
        elif t == ns.OP_MAXPOOL:
            ho, wo, _ = spec.tensors[op['out']]
            st = op['stride']
            pb = max((ho - 1) * st + 2 - x.shape[2], 0)
            pr = max((wo - 1) * st + 2 - x.shape[3], 0)
            y = F.max_pool2d(F.pad(x, (0, pr, 0, pb), value=float('-inf')), 2, st)
            if stats is not None and stats.get('__want_pre__'):
                stats[f'op{idx}/pool_in'] = x.detach().permute(0, 2, 3, 1).numpy()
        elif t == ns.OP_UPSAMPLE:
            y = F.interpolate(x, scale_factor=2, mode='nearest')
        elif t == ns.OP_CONCAT:
            y = torch.cat([x, T[op['in1']]], 1)
        elif t == ns.OP_ADD:
            y = x + T[op['in1']]
        else:
            raise ValueError(t)
        T[op['out']] = y
    e = 5 + spec.class_num
    return [T[o].permute(0, 2, 3, 1).reshape(x_nhwc.shape[0], *spec.tensors[o][:2], spec.anchor_num, e) for o in spec.outputs]


def yolo_loss_torch(yt, yp, anchors, obj_thresh, iou_thresh, ow, nw, ww, batch_size):
    """tools/utils.py:741-791 for one layer, float64 torch, differentiable in yp."""
    B, h, w, A, E = yp.shape
    anc = torch.as_tensor(np.asarray(anchors), dtype=torch.float64)
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    off = torch.stack([gx, gy], -1)[:, :, None, :].double()
    whv = torch.tensor([w, h]).double()
    obj = yt[..., 4:5]
    ob = yt[..., 4] > obj_thresh
    with torch.no_grad():
        axy = (torch.sigmoid(yp[..., 0:2]) + off) / whv
        awh = torch.exp(yp[..., 2:4]) * anc
        ign = torch.ones(B, h, w, A).double()
        for b in range(B):
            gxy, gwh = yt[b][..., 0:2][ob[b]], yt[b][..., 2:4][ob[b]]
            if len(gxy):
                p1, p2 = axy[b][..., None, :] - awh[b][..., None, :] / 2, axy[b][..., None, :] + awh[b][..., None, :] / 2
                g1, g2 = gxy - gwh / 2, gxy + gwh / 2
                iw = (torch.minimum(p2, g2) - torch.maximum(p1, g1)).clamp(min=0)
                inter = iw[..., 0] * iw[..., 1]
                iou = inter / (awh[b][..., None, 0] * awh[b][..., None, 1] + gwh[:, 0] * gwh[:, 1] - inter)
                ign[b] = (iou.max(-1).values < iou_thresh).double()
    gtxy = yt[..., 0:2] * whv - off
    gtwh = torch.where(ob[..., None], torch.log(yt[..., 2:4].clamp(min=1e-30) / anc), torch.zeros(1).double())
    cw = 2 - yt[..., 2:3] * yt[..., 3:4]
    bce = lambda z, x: F.binary_cross_entropy_with_logits(x, z, reduction='none')
    xy = (obj * cw * bce(gtxy, yp[..., 0:2])).sum() / batch_size
    wh = (obj * cw * ww * (gtwh - yp[..., 2:4]) ** 2).sum() / batch_size
    bc = bce(yt[..., 4:5], yp[..., 4:5])
    ol = ow * (obj * bc).sum() / batch_size
    nl = nw * ((1 - obj) * ign[..., None] * bc).sum() / batch_size
    cl = (obj * bce(yt[..., 5:], yp[..., 5:])).sum() / batch_size
    return ol + nl + cl + xy + wh


def loss_and_grads(spec: ns.NetSpec, weights: Dict[str, np.ndarray], x_nhwc: np.ndarray, y_true: Sequence[np.ndarray], anchors,
                   obj_thresh=0.7, iou_thresh=0.5, obj_weight=1.0, noobj_weight=1.0, wh_weight=1.0, batch_size=None, want_pre=False, with_reg=True):
    """-> (data_loss, reg_loss, grads dict in Keras layout, bn batch stats)."""
    trainable = [k for k in weights if not k.endswith(('/moving_mean', '/moving_variance'))]
    params = {k: torch.from_numpy(np.asarray(weights[k], np.float64)).requires_grad_(True) for k in trainable}
    x = torch.from_numpy(np.asarray(x_nhwc, np.float64))
    stats = {'__want_pre__': True} if want_pre else {}
    preds = forward_train(spec, params, x, stats)
    bs = batch_size if batch_size else x.shape[0]
    data = sum(yolo_loss_torch(torch.from_numpy(np.asarray(yt, np.float64)), yp, anchors[i], obj_thresh, iou_thresh, obj_weight,
                               noobj_weight, wh_weight, bs) for i, (yt, yp) in enumerate(zip(y_true, preds)))
    reg = sum(L2_WEIGHT * (params[l.name + '/kernel'] ** 2).sum() for l in spec.layers if l.kind == 'conv' and _is_darknet_conv(l.name))
    (data + reg if with_reg else data).backward()
    stats.pop('__want_pre__', None)
    grads = {k: (p.grad.numpy() if p.grad is not None else np.zeros(p.shape)) for k, p in params.items()}
    return float(data.detach()), float(reg.detach()), grads, stats, [p.detach().numpy() for p in preds]


class AdamRef:
    """keras.optimizers.Adam(lr, decay) update rule in float64 (keras/optimizers.py get_updates)."""

    def __init__(self, lr, decay=0.0, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.decay, self.b1, self.b2, self.eps, self.it = lr, decay, b1, b2, eps, 0
        self.m: Dict[str, np.ndarray] = {}
        self.v: Dict[str, np.ndarray] = {}

    def apply(self, weights: Dict[str, np.ndarray], grads: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
        lr = self.lr / (1.0 + self.decay * self.it)
        t = self.it + 1
        lr_t = lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        out = dict(weights)
        for k, g in grads.items():
            m = self.b1 * self.m.get(k, 0.0) + (1 - self.b1) * g
            v = self.b2 * self.v.get(k, 0.0) + (1 - self.b2) * g * g
            self.m[k], self.v[k] = m, v
            out[k] = np.asarray(weights[k], np.float64) - lr_t * m / (np.sqrt(v) + self.eps)
        self.it += 1
        return out
