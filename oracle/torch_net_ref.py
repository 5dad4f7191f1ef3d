"""torch_net_ref.py — CPU ORACLE (test infrastructure, NOT the product path).

Independent torch-CPU build of the conv stack (models/yolonet.py, keras_mobilenet*.py layers): the forward pass straight from
Keras-layout parameters with torch.nn.functional ops and UNFOLDED BatchNorm, so it shares no code with netspec.compile_plan's
folding, with oracle/yolo_net_ref.c, or with the HIP engine.  Two uses:
  * float64: the second arbiter tests/test_oracle_net.py checks yolo_net_ref.c against (the TF-1.14 layers are un-vendored, parity
    with Keras itself is unpinned);
  * float32: bench.py's `cpu_baseline` leg - the same graph Keras would run on the host cores, on oneDNN."""
import numpy as np
import torch
import torch.nn.functional as F

from k210_yolo_framework_amd import netspec as ns


def forward(spec: ns.NetSpec, weights, x_nhwc: np.ndarray, want=None, dtype=torch.float64, store_hook=None):
    """-> dict tensor_id -> NHWC fp32 numpy for ids in `want` (default: spec.outputs).
    store_hook(tensor_id, y_nchw) -> y_nchw: applied to every op's result before it is stored - a storage-format model (e.g. rounding to
    p significant bits) for pricing inter-layer formats (tools/r05_format_pricing.py); None = exact."""
    want = list(spec.outputs if want is None else want)
    lay = {l.name: l for l in spec.layers}
    T = {0: torch.from_numpy(np.ascontiguousarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).to(dtype)}
    with torch.no_grad():
        for op in spec.ops:
            x = T[op['in0']]
            t = op['type']
            if t in (ns.OP_CONV, ns.OP_DWCONV):
                l = lay[op['layer']]
                k = torch.from_numpy(weights[l.name + '/kernel']).to(dtype)
                hi, wi = x.shape[2], x.shape[3]
                ho, wo, _ = spec.tensors[op['out']]
                kk, st = op['k'], op['stride']
                pb = (ho - 1) * st + kk - hi - op['pad_t']
                pr = (wo - 1) * st + kk - wi - op['pad_l']
                xp = F.pad(x, (op['pad_l'], max(pr, 0), op['pad_t'], max(pb, 0)))
                if t == ns.OP_CONV:
                    w = k.permute(3, 2, 0, 1)                       # HWIO -> OIHW
                    b = torch.from_numpy(weights[l.name + '/bias']).to(dtype) if l.use_bias else None
                    y = F.conv2d(xp, w, b, stride=st)
                else:
                    w = k.permute(2, 3, 0, 1)                       # [3,3,C,1] -> [C,1,3,3]
                    y = F.conv2d(xp, w, None, stride=st, groups=x.shape[1])
                y = y[:, :, :ho, :wo]
                if l.bn_name:
                    g, bt, mu, var = (torch.from_numpy(weights[l.bn_name + s]).to(dtype)
                                      for s in ('/gamma', '/beta', '/moving_mean', '/moving_variance'))
                    y = F.batch_norm(y, mu, var, g, bt, training=False, eps=ns.BN_EPS)
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
            elif t == ns.OP_UPSAMPLE:
                y = F.interpolate(x, scale_factor=2, mode='nearest')
            elif t == ns.OP_CONCAT:
                y = torch.cat([x, T[op['in1']]], 1)
            elif t == ns.OP_ADD:
                y = x + T[op['in1']]
            else:
                raise ValueError(t)
            if store_hook is not None:
                y = store_hook(op['out'], y)
            T[op['out']] = y
    return {i: T[i].permute(0, 2, 3, 1).float().numpy() for i in want}


class Prepared:
    """The same graph with its parameters converted ONCE (torch tensors, OIHW, optional channels_last): what a framework holds after
    model load.  bench.py's cpu_baseline times this form - `forward()` above re-converts every numpy weight on every call, which is
    fine for a checker and unfair to a baseline."""

    def __init__(self, spec: ns.NetSpec, weights, dtype=torch.float32, channels_last: bool = True):
        self.spec, self.dtype, self.cl = spec, dtype, channels_last
        self.p = {}
        for l in spec.layers:
            k = torch.from_numpy(np.asarray(weights[l.name + '/kernel'])).to(dtype)
            w = (k.permute(3, 2, 0, 1) if l.kind == 'conv' else k.permute(2, 3, 0, 1)).contiguous()
            if channels_last:
                w = w.contiguous(memory_format=torch.channels_last)
            e = {'w': w, 'b': torch.from_numpy(np.asarray(weights[l.name + '/bias'])).to(dtype) if l.use_bias else None}
            if l.bn_name:
                e['bn'] = tuple(torch.from_numpy(np.asarray(weights[l.bn_name + s])).to(dtype)
                                for s in ('/moving_mean', '/moving_variance', '/gamma', '/beta'))
            self.p[l.name] = e

    def __call__(self, x_nhwc: np.ndarray):
        spec = self.spec
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).to(self.dtype)
        if self.cl:
            x = x.contiguous(memory_format=torch.channels_last)
        T = {0: x}
        with torch.no_grad():
            for op in spec.ops:
                x = T[op['in0']]
                t = op['type']
                if t in (ns.OP_CONV, ns.OP_DWCONV):
                    e = self.p[op['layer']]
                    ho, wo, _ = spec.tensors[op['out']]
                    kk, st = op['k'], op['stride']
                    pb = (ho - 1) * st + kk - x.shape[2] - op['pad_t']
                    pr = (wo - 1) * st + kk - x.shape[3] - op['pad_l']
                    if op['pad_t'] or op['pad_l'] or pb > 0 or pr > 0:
                        x = F.pad(x, (op['pad_l'], max(pr, 0), op['pad_t'], max(pb, 0)))
                    y = F.conv2d(x, e['w'], e['b'], stride=st, groups=x.shape[1] if t == ns.OP_DWCONV else 1)[:, :, :ho, :wo]
                    if 'bn' in e:
                        mu, var, g, bt = e['bn']
                        y = F.batch_norm(y, mu, var, g, bt, training=False, eps=ns.BN_EPS)
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
                elif t == ns.OP_UPSAMPLE:
                    y = F.interpolate(x, scale_factor=2, mode='nearest')
                elif t == ns.OP_CONCAT:
                    y = torch.cat([x, T[op['in1']]], 1)
                elif t == ns.OP_ADD:
                    y = x + T[op['in1']]
                else:
                    raise ValueError(t)
                T[op['out']] = y
        return [T[i].permute(0, 2, 3, 1).float().numpy() for i in spec.outputs]
