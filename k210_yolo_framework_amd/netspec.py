"""Static layer plans for the four detectors of the reference.

This module restates the *topology* that the reference builds with tf.keras
(`models/yolonet.py:12-260`, `models/keras_mobilenet.py:215-229,291-436`,
`models/keras_mobilenet_v2.py:118-125,311-382,426-485`) as a flat list of ops
over numbered tensors.  Nothing here computes: a plan is data.  It is consumed
by

  * the HIP engine (`csrc/yk_engine.hip`, through `yk_plan_create`), which
    fuses and launches gfx950 kernels from it, and
  * the CPU oracle (`oracle/yolo_net_ref.c`), test infrastructure only.

Layouts: activations NHWC; conv kernels are stored in the plan blob as
[Cout][kh][kw][Cin] fp32 (Keras HWIO transposed so the reduction axis is
contiguous), depthwise kernels as [kh][kw][C]; inference BatchNorm is folded
to a per-channel fp32 (scale, bias) pair applied after the accumulation:
y = acc*scale + bias  with  scale = gamma/sqrt(var+eps), bias = beta-mean*scale
(Keras BN inference formula; eps=1e-3 everywhere in the reference).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# ---- op codes shared with include/yolo_hip.h ------------------------------
OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_UPSAMPLE, OP_CONCAT, OP_ADD = 1, 2, 3, 4, 5, 6
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_LEAKY = 0, 1, 2, 3
OP_FIELDS = 24
FLAG_NET_OUTPUT = 1

BN_EPS = 1e-3  # keras BatchNormalization default, also explicit in keras_mobilenet_v2.py:320


def _f2i(x: float) -> int:
    return struct.unpack('<i', struct.pack('<f', float(x)))[0]


@dataclass
class Layer:
    """One parameterised layer (conv / depthwise conv) with Keras-style params."""
    name: str
    kind: str                    # 'conv' | 'dwconv'
    kernel_shape: Tuple[int, ...]  # HWIO for conv, [3,3,C,1] for dwconv
    use_bias: bool
    bn_name: Optional[str]


@dataclass
class NetSpec:
    name: str
    in_hw: Tuple[int, int]
    tensors: List[Tuple[int, int, int]] = field(default_factory=list)   # (h, w, c)
    ops: List[dict] = field(default_factory=list)
    layers: List[Layer] = field(default_factory=list)
    outputs: List[int] = field(default_factory=list)     # tensor ids of y1, y2[, y3]
    anchor_num: int = 3
    class_num: int = 20

    # ------------------------------------------------------------------ build
    def _new_tensor(self, h, w, c) -> int:
        self.tensors.append((int(h), int(w), int(c)))
        return len(self.tensors) - 1

    def conv(self, x: int, cout: int, k: int, stride: int = 1, pad=None, *, bn: bool = True,
             bias: bool = False, act=(ACT_NONE, 0.0), name: str, net_output: bool = False) -> int:
        h, w, cin = self.tensors[x]
        if pad is None:  # Keras 'same' at stride 1
            assert stride == 1
            p = (k - 1) // 2
            pad = (p, p, p, p)
        pt, pb, pl, pr = pad
        ho = (h + pt + pb - k) // stride + 1
        wo = (w + pl + pr - k) // stride + 1
        y = self._new_tensor(ho, wo, cout)
        self.layers.append(Layer(name, 'conv', (k, k, cin, cout), bias, name + '_bn' if bn else None))
        self.ops.append(dict(type=OP_CONV, in0=x, in1=-1, out=y, cin=cin, cout=cout, k=k, stride=stride,
                             pad_t=pt, pad_l=pl, pad_b=pb, pad_r=pr, act=act[0], alpha=act[1], layer=name,
                             flags=FLAG_NET_OUTPUT if net_output else 0))
        return y

    def dwconv(self, x: int, stride: int, pad, *, act, name: str) -> int:
        h, w, c = self.tensors[x]
        pt, pb, pl, pr = pad
        ho = (h + pt + pb - 3) // stride + 1
        wo = (w + pl + pr - 3) // stride + 1
        y = self._new_tensor(ho, wo, c)
        self.layers.append(Layer(name, 'dwconv', (3, 3, c, 1), False, name + '_bn'))
        self.ops.append(dict(type=OP_DWCONV, in0=x, in1=-1, out=y, cin=c, cout=c, k=3, stride=stride,
                             pad_t=pt, pad_l=pl, pad_b=pb, pad_r=pr, act=act[0], alpha=act[1], layer=name, flags=0))
        return y

    def maxpool(self, x: int, stride: int) -> int:
        """2x2 max pool, Keras padding='same' (extra row/col on the bottom/right, -inf)."""
        h, w, c = self.tensors[x]
        ho, wo = -(-h // stride), -(-w // stride)
        y = self._new_tensor(ho, wo, c)
        self.ops.append(dict(type=OP_MAXPOOL, in0=x, in1=-1, out=y, cin=c, cout=c, k=2, stride=stride,
                             pad_t=0, pad_l=0, act=0, alpha=0.0, layer=None, flags=0))
        return y

    def upsample(self, x: int) -> int:
        h, w, c = self.tensors[x]
        y = self._new_tensor(2 * h, 2 * w, c)
        self.ops.append(dict(type=OP_UPSAMPLE, in0=x, in1=-1, out=y, cin=c, cout=c, k=1, stride=1,
                             pad_t=0, pad_l=0, act=0, alpha=0.0, layer=None, flags=0))
        return y

    def concat(self, a: int, b: int) -> int:
        ha, wa, ca = self.tensors[a]
        hb, wb, cb = self.tensors[b]
        assert (ha, wa) == (hb, wb), (self.tensors[a], self.tensors[b])
        y = self._new_tensor(ha, wa, ca + cb)
        self.ops.append(dict(type=OP_CONCAT, in0=a, in1=b, out=y, cin=ca + cb, cout=ca + cb, k=1, stride=1,
                             pad_t=0, pad_l=0, act=0, alpha=0.0, layer=None, flags=0))
        return y

    def add(self, a: int, b: int) -> int:
        assert self.tensors[a] == self.tensors[b]
        h, w, c = self.tensors[a]
        y = self._new_tensor(h, w, c)
        self.ops.append(dict(type=OP_ADD, in0=a, in1=b, out=y, cin=c, cout=c, k=1, stride=1,
                             pad_t=0, pad_l=0, act=0, alpha=0.0, layer=None, flags=0))
        return y

    # ------------------------------------------------------------ statistics
    def out_hw(self) -> List[Tuple[int, int]]:
        return [self.tensors[t][:2] for t in self.outputs]

    def macs_per_image(self) -> int:
        n = 0
        for op in self.ops:
            _, _, _ = self.tensors[op['in0']]
            ho, wo, _ = self.tensors[op['out']]
            if op['type'] == OP_CONV:
                n += ho * wo * op['k'] ** 2 * op['cin'] * op['cout']
            elif op['type'] == OP_DWCONV:
                n += ho * wo * 9 * op['cin']
        return n

    def act_elems_per_image(self) -> int:
        """in+out activation elements of every conv layer (SURVEY 8(d) byte model)."""
        n = 0
        for op in self.ops:
            if op['type'] in (OP_CONV, OP_DWCONV):
                hi, wi, ci = self.tensors[op['in0']]
                ho, wo, co = self.tensors[op['out']]
                n += hi * wi * ci + ho * wo * co
        return n

    def weight_elems(self) -> int:
        return sum(int(np.prod(l.kernel_shape)) for l in self.layers)

    def conv_layer_count(self) -> int:
        return sum(1 for op in self.ops if op['type'] in (OP_CONV, OP_DWCONV))

    # --------------------------------------------------------------- weights
    def init_weights(self, seed: int = 1, conf_bias: float = -4.0) -> Dict[str, np.ndarray]:
        """Seeded synthetic weights: He-normal kernels, BN gamma~U(.5,1.5),
        beta,mean~N(0,.1), var~U(.5,1.5); head bias ~N(0,.1) with the conf logit shifted."""
        rng = np.random.default_rng(seed)
        w: Dict[str, np.ndarray] = {}
        e = 5 + self.class_num
        for l in self.layers:
            kh, kw, ci, co = l.kernel_shape
            fan_in = kh * kw * (ci if l.kind == 'conv' else 1)
            w[l.name + '/kernel'] = rng.normal(0.0, np.sqrt(2.0 / fan_in), l.kernel_shape).astype(np.float32)
            if l.use_bias:
                b = rng.normal(0.0, 0.1, (co,)).astype(np.float32)
                if co == self.anchor_num * e:
                    b[4::e] += conf_bias
                w[l.name + '/bias'] = b
            if l.bn_name:
                c = co if l.kind == 'conv' else ci
                w[l.bn_name + '/gamma'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
                w[l.bn_name + '/beta'] = rng.normal(0.0, 0.1, (c,)).astype(np.float32)
                w[l.bn_name + '/moving_mean'] = rng.normal(0.0, 0.1, (c,)).astype(np.float32)
                w[l.bn_name + '/moving_variance'] = rng.uniform(0.5, 1.5, (c,)).astype(np.float32)
        return w

    def init_keras_default(self, seed: int = 1) -> Dict[str, np.ndarray]:
        """What a freshly built Keras model holds (the starting point of keras_train.py when no checkpoint is given):
        glorot_uniform kernels (Conv2D / DepthwiseConv2D defaults), zero biases, BatchNormalization gamma=1, beta=0,
        moving_mean=0, moving_variance=1.  init_weights() is the benchmark/test initialiser and is not used for training."""
        rng = np.random.default_rng(seed)
        w: Dict[str, np.ndarray] = {}
        for l in self.layers:
            kh, kw, ci, co = l.kernel_shape
            lim = np.sqrt(6.0 / (kh * kw * ci + kh * kw * co))           # fan_in = rf*shape[-2], fan_out = rf*shape[-1]
            w[l.name + '/kernel'] = rng.uniform(-lim, lim, l.kernel_shape).astype(np.float32)
            if l.use_bias:
                w[l.name + '/bias'] = np.zeros((co,), np.float32)
            if l.bn_name:
                c = co if l.kind == 'conv' else ci
                w[l.bn_name + '/gamma'] = np.ones((c,), np.float32)
                w[l.bn_name + '/beta'] = np.zeros((c,), np.float32)
                w[l.bn_name + '/moving_mean'] = np.zeros((c,), np.float32)
                w[l.bn_name + '/moving_variance'] = np.ones((c,), np.float32)
        return w

    # ------------------------------------------------------------- serialise
    def compile_plan(self, weights: Dict[str, np.ndarray]):
        """-> (ops int32 [n,OP_FIELDS], tensors int32 [t,4], blob float32 [n]) for yk_plan_create.

        BN folding is done here in float64 and rounded once to fp32."""
        lay = {l.name: l for l in self.layers}
        blob: List[np.ndarray] = []
        off = 0

        def push(a: np.ndarray) -> int:
            nonlocal off
            a = np.ascontiguousarray(a, dtype=np.float32).ravel()
            pad = (-a.size) % 8          # keep every segment 32-byte aligned
            blob.append(a)
            if pad:
                blob.append(np.zeros(pad, np.float32))
            o = off
            off += a.size + pad
            return o

        rows = np.zeros((len(self.ops), OP_FIELDS), np.int32)
        for i, op in enumerate(self.ops):
            w_off = s_off = b_off = -1
            if op['type'] in (OP_CONV, OP_DWCONV):
                l = lay[op['layer']]
                k = weights[l.name + '/kernel'].astype(np.float64)
                if l.kind == 'conv':
                    kk = np.transpose(k, (3, 0, 1, 2))       # HWIO -> OHWI
                    c = l.kernel_shape[3]
                else:
                    kk = k[..., 0]                            # [3,3,C]
                    c = l.kernel_shape[2]
                if l.bn_name:
                    g = weights[l.bn_name + '/gamma'].astype(np.float64)
                    bt = weights[l.bn_name + '/beta'].astype(np.float64)
                    mu = weights[l.bn_name + '/moving_mean'].astype(np.float64)
                    var = weights[l.bn_name + '/moving_variance'].astype(np.float64)
                    scale = g / np.sqrt(var + BN_EPS)
                    bias = bt - mu * scale
                else:
                    scale = np.ones(c)
                    bias = weights[l.name + '/bias'].astype(np.float64) if l.use_bias else np.zeros(c)
                w_off, s_off, b_off = push(kk), push(scale), push(bias)
            hi, wi, _ = self.tensors[op['in0']]
            ho, wo, _ = self.tensors[op['out']]
            rows[i, :20] = [op['type'], op['in0'], op['in1'], op['out'], op['cin'], op['cout'], op['k'],
                            op['stride'], op['pad_t'], op['pad_l'], op['act'], _f2i(op['alpha']),
                            w_off, s_off, b_off, op['flags'], hi, wi, ho, wo]
        tens = np.zeros((len(self.tensors), 4), np.int32)
        for i, (h, w_, c) in enumerate(self.tensors):
            tens[i] = [h, w_, c, 1 if i == 0 else 0]
        b = np.concatenate(blob) if blob else np.zeros(0, np.float32)
        return rows, tens, b


# =============================================================================
# builders (reference plugin API: net(input_shape, anchor_num, class_num, alpha=..))
# =============================================================================
LEAKY01 = (ACT_LEAKY, 0.1)     # DarknetConv2D_BN_Leaky, yolonet.py:260
LEAKY03 = (ACT_LEAKY, 0.3)     # keras LeakyReLU() default alpha, keras_mobilenet.py:356,436
RELU = (ACT_RELU, 0.0)         # keras_mobilenet.py:428 (plain ReLU, not ReLU6)
RELU6 = (ACT_RELU6, 6.0)
K210_S2_PAD = (1, 1, 1, 1)     # ZeroPadding2D(((1,1),(1,1))) + 'valid', keras_mobilenet.py:343,418
SAME3 = (1, 1, 1, 1)


def _head(s: NetSpec, x1: int, x2: int, y1_mid: int, y2_mid: int, up_mid: int, out_c: int, idx: List[int]):
    """y1/y2 heads shared by yolo_mobilev1 / yolo_mobilev2 / tiny_yolo (yolonet.py:27-38,86-96,128-138)."""
    def nm(prefix):
        idx[0] += 1
        return f'{prefix}_{idx[0]}'
    t = s.conv(x2, y1_mid, 3, act=LEAKY01, name=nm('head_conv'))
    y1 = s.conv(t, out_c, 1, bn=False, bias=True, name=nm('head_conv'), net_output=True)
    t = s.conv(x2, up_mid, 1, act=LEAKY01, name=nm('head_conv'))
    t = s.upsample(t)
    t = s.concat(t, x1)                       # upsampled channels FIRST (yolonet.py:38)
    t = s.conv(t, y2_mid, 3, act=LEAKY01, name=nm('head_conv'))
    y2 = s.conv(t, out_c, 1, bn=False, bias=True, name=nm('head_conv'), net_output=True)
    s.outputs = [y1, y2]


def yolo_mobilev1(input_shape, anchor_num: int, class_num: int, alpha: float = 1.0) -> NetSpec:
    """models/yolonet.py:12-46 over models/keras_mobilenet.py:215-229."""
    H, W, C = input_shape
    assert C == 3
    s = NetSpec('yolo_mobilev1', (H, W), anchor_num=anchor_num, class_num=class_num)
    x = s._new_tensor(H, W, 3)
    x = s.conv(x, int(32 * alpha), 3, 2, K210_S2_PAD, act=LEAKY03, name='conv1')
    cfg = [(40 if alpha == 1.0 else 64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2),
           (512, 1), (512, 1), (512, 1), (512, 1), (512, 1), (1024, 2), (1024, 1)]
    x1 = None
    for i, (f, st) in enumerate(cfg, start=1):
        x = s.dwconv(x, st, K210_S2_PAD if st == 2 else SAME3, act=RELU, name=f'conv_dw_{i}')
        x = s.conv(x, int(f * alpha), 1, act=LEAKY03, name=f'conv_pw_{i}')
        if i == 11:
            x1 = x                             # conv_pw_11_relu, yolonet.py:23
    _head(s, x1, x, 128 if alpha > 0.8 else 192, 128, 128, anchor_num * (class_num + 5), [0])
    return s


def _make_divisible(v, divisor, min_value=None):
    """models/keras_mobilenet_v2.py:118-125."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def yolo_mobilev2(input_shape, anchor_num: int, class_num: int, alpha: float = 1.0) -> NetSpec:
    """models/yolonet.py:49-104 over models/keras_mobilenet_v2.py:311-382,426-485."""
    H, W, C = input_shape
    assert C == 3
    s = NetSpec('yolo_mobilev2', (H, W), anchor_num=anchor_num, class_num=class_num)
    x = s._new_tensor(H, W, 3)
    x = s.conv(x, 32, 3, 2, K210_S2_PAD, act=RELU6, name='Conv1')   # fixed 32 ch (:313)
    x1 = None

    def block(x, filters, stride, expansion, block_id, expand_channel=None):
        nonlocal x1
        cin = s.tensors[x][2]
        pw = _make_divisible(int(filters * alpha), 8)
        inp = x
        prefix = f'block_{block_id}_' if block_id else 'expanded_conv_'
        if block_id:
            x = s.conv(x, expand_channel if expand_channel else expansion * cin, 1, act=RELU6,
                       name=prefix + 'expand')
            if block_id == 13:
                x1 = x                         # block_13_expand_relu, yolonet.py:83
        x = s.dwconv(x, stride, K210_S2_PAD if stride == 2 else SAME3, act=RELU6, name=prefix + 'depthwise')
        x = s.conv(x, pw, 1, act=(ACT_NONE, 0.0), name=prefix + 'project')
        if cin == pw and stride == 1:
            x = s.add(inp, x)
        return x

    x = block(x, 16, 1, 1, 0)
    x = block(x, 24, 2, 6, 1, 48 if alpha > .6 else None)
    x = block(x, 24, 1, 6, 2, 124 if alpha > .6 else None)
    for bid, (f, st) in zip(range(3, 17), [(32, 2), (32, 1), (32, 1), (64, 2), (64, 1), (64, 1), (64, 1),
                                           (96, 1), (96, 1), (96, 1), (160, 2), (160, 1), (160, 1), (320, 1)]):
        x = block(x, f, st, 6, bid)
    last = _make_divisible(1280 * alpha, 8) if alpha > 1.0 else 1280
    x = s.conv(x, last, 1, act=RELU6, name='Conv_1')
    mid = 128 if alpha > 0.7 else 192
    _head(s, x1, x, mid, mid, 128, anchor_num * (class_num + 5), [0])
    return s


def tiny_yolo(input_shape, anchor_num: int, class_num: int, alpha: float = 1.0) -> NetSpec:
    """models/yolonet.py:107-143 (output reshape derived from out_hw)."""
    H, W, C = input_shape
    s = NetSpec('tiny_yolo', (H, W), anchor_num=anchor_num, class_num=class_num)
    x = s._new_tensor(H, W, 3)
    n = [0]

    def cbl(x, f, k):
        n[0] += 1
        return s.conv(x, f, k, act=LEAKY01, name=f'conv2d_{n[0]}')
    for f in (16, 32, 64, 128):
        x = cbl(x, f, 3)
        x = s.maxpool(x, 2)
    x1 = cbl(x, 256, 3)
    x = s.maxpool(x1, 2)
    x = cbl(x, 512, 3)
    x = s.maxpool(x, 1)
    x = cbl(x, 1024, 3)
    x2 = cbl(x, 256, 1)
    out_c = anchor_num * (class_num + 5)
    t = cbl(x2, 512, 3)
    n[0] += 1
    y1 = s.conv(t, out_c, 1, bn=False, bias=True, name=f'conv2d_{n[0]}', net_output=True)
    t = cbl(x2, 128, 1)
    t = s.upsample(t)
    t = s.concat(t, x1)
    t = cbl(t, 256, 3)
    n[0] += 1
    y2 = s.conv(t, out_c, 1, bn=False, bias=True, name=f'conv2d_{n[0]}', net_output=True)
    s.outputs = [y1, y2]
    return s


def yolo(input_shape, anchor_num: int, class_num: int, alpha: float = 1.0) -> NetSpec:
    """Darknet-53 YOLOv3, models/yolonet.py:161-229."""
    H, W, C = input_shape
    s = NetSpec('yolo', (H, W), anchor_num=anchor_num, class_num=class_num)
    x = s._new_tensor(H, W, 3)
    n = [0]

    def cbl(x, f, k, stride=1):
        n[0] += 1
        pad = (1, 0, 1, 0) if stride == 2 else None      # ZeroPadding2D(((1,0),(1,0))) + valid (:197)
        return s.conv(x, f, k, stride, pad, act=LEAKY01, name=f'conv2d_{n[0]}')

    def resblock(x, f, nb):
        x = cbl(x, f, 3, 2)
        for _ in range(nb):
            y = cbl(x, f // 2, 1)
            y = cbl(y, f, 3)
            x = s.add(x, y)
        return x

    def last_layers(x, f, out_c):
        x = cbl(x, f, 1)
        x = cbl(x, f * 2, 3)
        x = cbl(x, f, 1)
        x = cbl(x, f * 2, 3)
        x = cbl(x, f, 1)
        y = cbl(x, f * 2, 3)
        n[0] += 1
        y = s.conv(y, out_c, 1, bn=False, bias=True, name=f'conv2d_{n[0]}', net_output=True)
        return x, y

    x = cbl(x, 32, 3)
    x = resblock(x, 64, 1)
    x = resblock(x, 128, 2)
    r52 = x = resblock(x, 256, 8)     # darknet.layers[92]  (52x52x256 @416)
    r26 = x = resblock(x, 512, 8)     # darknet.layers[152] (26x26x512 @416)
    x = resblock(x, 1024, 4)
    out_c = anchor_num * (class_num + 5)
    x, y1 = last_layers(x, 512, out_c)
    x = s.concat(s.upsample(cbl(x, 256, 1)), r26)
    x, y2 = last_layers(x, 256, out_c)
    x = s.concat(s.upsample(cbl(x, 128, 1)), r52)
    x, y3 = last_layers(x, 128, out_c)
    s.outputs = [y1, y2, y3]
    return s


NETWORKS = {'yolo_mobilev1': yolo_mobilev1, 'yolo_mobilev2': yolo_mobilev2,
            'tiny_yolo': tiny_yolo, 'yolo': yolo}
