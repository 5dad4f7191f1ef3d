"""K210 kmodel (v3) reader: recover the trained, 8-bit quantised yolo_mobilev1-0.75 of the reference's K210 demo as Keras-named
float weights this framework can run.

The only trained weights in the reference tree are inside `yolo3_frame_test_public/kfpkg/kpu_yolov3.kfpkg` (a zip): `yolo.kmodel`,
3 926 440 bytes, flashed at 0x00A00000 and run by `main.c:274,303` through the Kendryte SDK's `kpu_load_kmodel / kpu_run_kmodel`.
The container is nncase v0.1's "kmodel v3" (third party, un-vendored; restated here from the published Kendryte standalone SDK
`kpu.h` / `kpu.c` and nncase's K210 kernels):

    header   7 x u32   version(3) flags arch layers_length max_start_address main_mem_usage output_count
    outputs  output_count x (address, size)                  in main memory
    layers   layers_length x (type, body_size)               then the bodies, back to back
    K210 conv body (type 10240): flags, main_mem_out_address, layer_offset, weights_offset, bn_offset, act_offset (file offsets)
        layer:   twelve 64-bit KPU registers (kpu_layer_argument_t): channels-1, sizes-1, kernel 1x1/3x3, pool type, pad value, depthwise
                 bit, and the zero-point terms arg_x >> shr_x, arg_w >> shr_w, arg_add
        weights: uint8 [oc][ic][kh*kw] (depthwise: [c][kh*kw])
        bn:      one u64 per output channel: mul:24 add:32 shift:4
        act:     16 segments (shift:8, y_mul:16, x_start:36 signed) + 16 result biases: a piecewise-linear table in the integer domain
    main-memory layers: DEQUANTIZE(12), REQUANTIZE(13, 256-entry table), QUANTIZED_CONCAT(17), QUANTIZED_RESIZE_NEAREST_NEIGHBOR(23),
        K210_UPLOAD(10243)

The KPU computes, per output (nncase `kpu_conv2d`):
    acc = sum(x*w) + (arg_x * sum(x) >> shr_x) + (arg_w * sum(w) >> shr_w) + arg_add * in_channels_per_group     x, w uint8; pad = pad_value
    z   = (acc * bn.mul >> bn.shift) + bn.add
    y   = clamp(((z - seg.x_start) * seg.y_mul >> seg.shift) + seg.bias, 0, 255)        seg = last segment with z > x_start
which is the asymmetric-quantised form of  y_real = act(bn(sum(x_real * w_real)))  with x_real = s_x (x - zp_x), w_real = s_w (w - zp_w),
zp_x = -arg_w / 2^shr_w and zp_w = -arg_x / 2^shr_x.  `to_float_weights` re-expresses every layer in CENTRED integer units
(x~ = x - zp_x, w~ = w - zp_w: real zero is zero, so Keras zero padding is right) and folds the BN multiplier, the activation table's
slope and the requantisation factors into a per-channel (scale, bias) + LeakyReLU(alpha) - exactly the layer form of
`models/yolonet.py:12-43` / `keras_mobilenet.py:291-436`, so the result loads into `yolonet.yolo_mobilev1` under the Keras names.
What is lost: the 8-bit rounding and range clipping of every activation (the float network is the un-clipped one).

`oracle/kpu_ref.py` (test infrastructure) runs the integer pipeline above bit by bit; `tests/test_kmodel.py` checks this module
against it and against the known answer of the demo picture (`kfpkg/dog.jpg` -> dog, bicycle, car; README.md:121-128,166).
"""
from __future__ import annotations

import io
import struct
import zipfile
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

KL_DEQUANTIZE, KL_REQUANTIZE, KL_QUANTIZED_CONCAT, KL_QUANTIZED_RESIZE_NN = 12, 13, 17, 23
KL_K210_CONV, KL_K210_UPLOAD = 10240, 10243
KLF_MAIN_MEM_OUT = 1
POOL_BYPASS, POOL_LEFT_TOP_2_S2 = 0, 5


class KmodelError(ValueError):
    pass


def _bits(v: int, lo: int, n: int, signed: bool = False) -> int:
    x = (v >> lo) & ((1 << n) - 1)
    if signed and x >> (n - 1):
        x -= 1 << n
    return x


@dataclass
class ConvLayer:
    index: int
    flags: int
    main_mem_out: int
    src_addr: int
    dst_addr: int
    in_ch: int
    out_ch: int
    in_w: int
    in_h: int
    out_w: int
    out_h: int
    ksize: int                    # 1 or 3
    pool_type: int
    pad_value: int
    depthwise: bool
    shr_w: int
    shr_x: int
    arg_w: int
    arg_x: int
    arg_add: int
    weights: np.ndarray           # uint8 [oc][ic][k*k] (depthwise: [c][1][k*k])
    bn_mul: np.ndarray            # int64 [oc]
    bn_add: np.ndarray
    bn_shift: np.ndarray
    act_start: np.ndarray         # int64 [16]
    act_mul: np.ndarray
    act_shift: np.ndarray
    act_bias: np.ndarray          # int64 [16]

    @property
    def zp_x(self) -> float:
        return -self.arg_w / float(1 << self.shr_w)

    @property
    def zp_w(self) -> float:
        return -self.arg_x / float(1 << self.shr_x)


@dataclass
class MemLayer:
    index: int
    type: int
    fields: dict


@dataclass
class Kmodel:
    version: int
    main_mem_usage: int
    outputs: List[Tuple[int, int]]
    layers: List[object] = field(default_factory=list)

    @property
    def convs(self) -> List[ConvLayer]:
        return [l for l in self.layers if isinstance(l, ConvLayer)]


def read_kfpkg(path) -> bytes:
    """The `yolo.kmodel` member of a .kfpkg (a zip with flash-list.json)."""
    with zipfile.ZipFile(path) as z:
        names = [n for n in z.namelist() if n.endswith('.kmodel')]
        if not names:
            raise KmodelError(f'{path}: no .kmodel member')
        return z.read(names[0])


def parse(data: bytes) -> Kmodel:
    if len(data) < 28:
        raise KmodelError('kmodel: truncated header')
    ver, _flags, _arch, nl, _maxstart, mainmem, nout = struct.unpack_from('<7I', data, 0)
    if ver != 3:
        raise KmodelError(f'kmodel: version {ver} (only v3, nncase 0.1, is understood)')
    off = 28
    if nout > 64 or nl > 4096 or off + 8 * (nout + nl) > len(data):
        raise KmodelError(f'kmodel: header announces {nout} outputs and {nl} layers, the file has {len(data)} bytes')
    outs = [struct.unpack_from('<2I', data, off + 8 * i) for i in range(nout)]
    off += 8 * nout
    hdrs = [struct.unpack_from('<2I', data, off + 8 * i) for i in range(nl)]
    pos = off + 8 * nl
    km = Kmodel(ver, mainmem, [(int(a), int(s)) for a, s in outs])
    for i, (ty, sz) in enumerate(hdrs):
        if pos + sz > len(data):
            raise KmodelError(f'kmodel: layer {i} runs past the end of the file')
        body = data[pos:pos + sz]
        need = {KL_K210_CONV: 24, KL_DEQUANTIZE: 24, KL_REQUANTIZE: 272, KL_QUANTIZED_CONCAT: 12, KL_QUANTIZED_RESIZE_NN: 36, KL_K210_UPLOAD: 24}.get(ty)
        if need is not None and sz < need:
            raise KmodelError(f'kmodel: layer {i} (type {ty}) has a {sz}-byte body, at least {need} expected')
        if ty == KL_K210_CONV:
            km.layers.append(_parse_conv(i, data, body))
        elif ty == KL_DEQUANTIZE:
            fl, src, dst, cnt = struct.unpack_from('<4I', body, 0)
            sc, bs = struct.unpack_from('<2f', body, 16)
            km.layers.append(MemLayer(i, ty, dict(flags=fl, src=src, dst=dst, count=cnt, scale=sc, bias=bs)))
        elif ty == KL_REQUANTIZE:
            fl, src, dst, cnt = struct.unpack_from('<4I', body, 0)
            km.layers.append(MemLayer(i, ty, dict(flags=fl, src=src, dst=dst, count=cnt, table=np.frombuffer(body, np.uint8, 256, 16).copy())))
        elif ty == KL_QUANTIZED_CONCAT:
            fl, dst, n = struct.unpack_from('<3I', body, 0)
            if 12 + 8 * n > sz:
                raise KmodelError(f'kmodel: layer {i}: concat of {n} inputs does not fit its {sz}-byte body')
            ins = [struct.unpack_from('<2I', body, 12 + 8 * k) for k in range(n)]
            km.layers.append(MemLayer(i, ty, dict(flags=fl, dst=dst, inputs=[(int(a), int(s)) for a, s in ins])))
        elif ty == KL_QUANTIZED_RESIZE_NN:
            fl, src, dst, w, h, c, ow, oh, align = struct.unpack_from('<9I', body, 0)
            km.layers.append(MemLayer(i, ty, dict(flags=fl, src=src, dst=dst, in_w=w, in_h=h, channels=c, out_w=ow, out_h=oh, align=align)))
        elif ty == KL_K210_UPLOAD:
            fl, src, kpu, w, h, c = struct.unpack_from('<6I', body, 0)
            km.layers.append(MemLayer(i, ty, dict(flags=fl, src=src, kpu_addr=kpu, width=w, height=h, channels=c)))
        else:
            raise KmodelError(f'kmodel: layer {i} has type {ty}, which this reader does not know')
        pos += sz
    return km


def _parse_conv(index: int, data: bytes, body: bytes) -> ConvLayer:
    fl, mmout, lo, wo, bo, ao = struct.unpack_from('<6I', body, 0)
    # every offset comes out of an untrusted file: check each region against the file before touching it
    n = len(data)
    if lo + 96 > n:
        raise KmodelError(f'kmodel: conv layer {index}: register block at {lo} runs past the end of the file ({n} bytes)')
    if not (wo <= bo <= n):
        raise KmodelError(f'kmodel: conv layer {index}: weight / BatchNorm offsets {wo}, {bo} are out of order or past the end of the file')
    if ao + 128 + 16 > n:
        raise KmodelError(f'kmodel: conv layer {index}: activation table at {ao} runs past the end of the file')
    r = struct.unpack_from('<12Q', data, lo)
    depthwise = bool(_bits(r[0], 3, 1))
    src, dst = _bits(r[1], 0, 15), _bits(r[1], 32, 15)
    ic, oc = _bits(r[2], 0, 10) + 1, _bits(r[2], 32, 10) + 1
    iw, ih = _bits(r[3], 0, 10) + 1, _bits(r[3], 10, 9) + 1
    ow, oh = _bits(r[3], 32, 10) + 1, _bits(r[3], 42, 9) + 1
    ktype, pool, padv = _bits(r[4], 0, 3), _bits(r[4], 4, 4), _bits(r[4], 24, 8)
    ks = 3 if ktype == 1 else 1
    shr_w, shr_x = _bits(r[9], 0, 4), _bits(r[9], 4, 4)
    arg_w, arg_x = _bits(r[9], 8, 24, True), _bits(r[9], 32, 24, True)
    arg_add = _bits(r[10], 0, 40, True)
    nw = oc * ks * ks * (1 if depthwise else ic)
    # (kernel_load_cfg.para_size is the bytes of ONE parameter load; big layers are loaded in several passes of o_ch_num_coef channels)
    if bo - wo < nw:
        raise KmodelError(f'kmodel: conv layer {index}: {bo - wo} weight bytes in the file, {nw} expected (16-bit weights are not supported)')
    if bo + 8 * oc > n:
        raise KmodelError(f'kmodel: conv layer {index}: BatchNorm table of {oc} channels at {bo} runs past the end of the file')
    w = np.frombuffer(data, np.uint8, nw, wo).reshape(oc, 1 if depthwise else ic, ks * ks).copy()
    bn = np.frombuffer(data, '<u8', oc, bo)
    bn_mul = np.array([_bits(int(v), 0, 24, True) for v in bn], np.int64)
    bn_add = np.array([_bits(int(v), 24, 32, True) for v in bn], np.int64)
    bn_shift = np.array([_bits(int(v), 56, 4) for v in bn], np.int64)
    act = np.frombuffer(data, '<u8', 16, ao)
    a_shift = np.array([_bits(int(v), 0, 8) for v in act], np.int64)
    a_mul = np.array([_bits(int(v), 8, 16, True) for v in act], np.int64)
    a_start = np.array([_bits(int(v), 24, 36, True) for v in act], np.int64)
    a_bias = np.frombuffer(data, np.int8, 16, ao + 128).astype(np.int64)
    return ConvLayer(index, fl, mmout, src, dst, ic, oc, iw, ih, ow, oh, ks, pool, padv, depthwise, shr_w, shr_x, arg_w, arg_x, arg_add, w,
                     bn_mul, bn_add, bn_shift, a_start, a_mul, a_shift, a_bias)


# ---- dequantisation into Keras-named float weights ---------------------------------------------------------------------------------
def _act_fit(c: ConvLayer) -> Tuple[float, float, float, float]:
    """The activation table as  y = y0 + s_pos * (z - z0)  for z >= z0,  y0 + s_neg * (z - z0)  below: returns (z0, y0, s_pos, s_neg).
    The table has 16 segments; nncase emits at most one kink inside the un-clamped range (ReLU / LeakyReLU / linear) plus the two
    clamping ends, which the float network does not need."""
    segs = []
    for k in range(16):
        lo = float(c.act_start[k])
        hi = float(c.act_start[k + 1]) if k + 1 < 16 else np.inf
        if hi <= lo:
            continue
        slope = float(c.act_mul[k]) / float(1 << int(c.act_shift[k]))
        segs.append((lo, hi, slope, float(c.act_bias[k])))
    # value of the table at a point
    def f(z):
        for lo, hi, s, b in reversed(segs):
            if z > lo:
                return (z - lo) * s + b
        lo, hi, s, b = segs[0]
        return (z - lo) * s + b
    live = [sg for sg in segs if sg[2] != 0.0]
    if not live:
        raise KmodelError(f'kmodel: conv layer {c.index}: activation table has no live segment')
    slopes = sorted({round(sg[2], 12) for sg in live})
    s_pos = max(slopes)
    s_neg = min(slopes) if len(slopes) > 1 else s_pos
    # the kink: where the steepest segment family starts
    pos = [sg for sg in live if abs(sg[2] - s_pos) < 1e-12]
    z0 = min(sg[0] for sg in pos)
    if s_neg == s_pos:                                         # linear, or ReLU whose flat part is the zero-slope (clamped) segment
        flat = [sg for sg in segs if sg[2] == 0.0 and sg[1] <= z0 + 1e-9]
        if flat:
            s_neg = 0.0
    return z0, f(z0 + 1e-9), s_pos, s_neg


def _requant_factor(table: np.ndarray) -> Tuple[float, float]:
    """A REQUANTIZE table is round(a * q + b) clipped to 0..255: least-squares (a, b) over its un-clipped part."""
    q = np.arange(256, dtype=np.float64)
    t = table.astype(np.float64)
    ok = (t > 0) & (t < 255)
    if ok.sum() < 8:
        ok = np.ones(256, bool)
    a, b = np.polyfit(q[ok], t[ok], 1)
    return float(a), float(b)


YOLO_MOBILEV1_ORDER = (['conv1'] + [n for i in range(1, 14) for n in (f'conv_dw_{i}', f'conv_pw_{i}')] +
                       ['head_conv_1', 'head_conv_2', 'head_conv_3', 'head_conv_4', 'head_conv_5'])


def to_float_weights(km: Kmodel) -> Tuple[Dict[str, np.ndarray], dict]:
    """Float parameters of `yolonet.yolo_mobilev1(alpha=0.75)` under this framework's (Keras layer) names + a report (activation slopes found, zero points, the
    requantisation factors).  Conv kernels HWIO, depthwise [3,3,C,1], BatchNorm as gamma / beta / moving_mean 0 / moving_variance 1-eps
    so that Keras' inference formula reproduces (scale, bias) exactly; the two output convs carry a bias."""
    convs = km.convs
    if len(convs) != len(YOLO_MOBILEV1_ORDER):
        raise KmodelError(f'kmodel: {len(convs)} KPU conv layers; the yolo_mobilev1 graph has {len(YOLO_MOBILEV1_ORDER)}')
    # every conv against the layer of yolo_mobilev1-0.75 it is mapped onto (kernel size, depthwise, channels, no KPU pooling): a kmodel
    # of another network with the same NUMBER of convs must not be dequantised into this one's names
    from . import netspec as ns
    spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
    want = {l.name: l for l in spec.layers}
    for name, c in zip(YOLO_MOBILEV1_ORDER, convs):
        kh, _, ci, co = want[name].kernel_shape
        dw = want[name].kind == 'dwconv'
        exp = (kh, dw, ci, ci if dw else co)
        got = (c.ksize, bool(c.depthwise), c.in_ch, c.out_ch)
        if got != exp:
            raise KmodelError(f'kmodel: conv {c.index} is (k, depthwise, in, out) = {got}; yolo_mobilev1-0.75 layer {name} is {exp}')
        if c.pool_type not in (0, 5):          # 5 = keep the top-left sample of every 2x2 window: how the KPU runs a stride-2 conv
            raise KmodelError(f'kmodel: conv {c.index} ({name}) uses KPU pooling type {c.pool_type}; the graph has no pooling layers')
    mem = [l for l in km.layers if isinstance(l, MemLayer)]
    deq = {l.fields['src']: l.fields for l in mem if l.type == KL_DEQUANTIZE}
    req = [l.fields for l in mem if l.type == KL_REQUANTIZE]
    cat = [l.fields for l in mem if l.type == KL_QUANTIZED_CONCAT]
    if len(req) != 2 or len(cat) != 1 or len(deq) != 2:
        raise KmodelError('kmodel: expected the yolo head (two dequantised outputs, one requantised concat)')
    # which requantise feeds which half of the concat (inputs are listed in concat order: [upsampled head branch, backbone])
    (c0_addr, c0_size), (c1_addr, c1_size) = cat[0]['inputs']
    r_first = next(r for r in req if r['dst'] == c0_addr)
    r_second = next(r for r in req if r['dst'] == c1_addr)
    a_first, _ = _requant_factor(r_first['table'])
    a_second, _ = _requant_factor(r_second['table'])
    by_name = dict(zip(YOLO_MOBILEV1_ORDER, convs))
    out: Dict[str, np.ndarray] = {}
    report = {'layers': {}, 'requant': {'upsampled_branch': a_first, 'backbone_branch': a_second}}
    eps = 1e-3
    for name, c in by_name.items():
        z0, y0, s_pos, s_neg = _act_fit(c)
        alpha = s_neg / s_pos
        wq = c.weights.astype(np.float64) - c.zp_w                                   # centred weights [oc][ic][kk]
        mul = c.bn_mul.astype(np.float64) / np.exp2(c.bn_shift.astype(np.float64))
        add = c.bn_add.astype(np.float64)
        scale = s_pos * mul
        bias = s_pos * (add - z0)
        in_gain = np.ones(wq.shape[1])
        if name == 'conv1':
            if abs(c.zp_x) > 0.5:
                raise KmodelError(f'kmodel: the input zero point is {c.zp_x}, expected 0 (raw u8 pixels)')
            in_gain[:] = 255.0                                                       # the engine feeds img / max(img) in [0, 1]
        if name == 'head_conv_4':                                                    # concat input: [requantised up(head_conv_3) | requantised conv_pw_11]
            n_first = by_name['head_conv_3'].out_ch
            in_gain[n_first:] = a_second                                             # conv_pw_11 keeps its own domain (conv_dw_12 reads it too)
        if name == 'head_conv_3':
            scale, bias = scale * a_first, bias * a_first                           # LeakyReLU is positively homogeneous
        if name in ('head_conv_2', 'head_conv_5'):                                   # network outputs: dequantised to real logits
            dq = deq.get(c.main_mem_out)
            if dq is None:
                raise KmodelError(f'kmodel: output conv {name} is not followed by a DEQUANTIZE')
            # the output convs are linear in Keras; their table is the identity above z0 and flat below it, which is only the lower end
            # of the 8-bit range (real = q * scale + bias, q = 0 is the calibrated minimum): not an activation
            if abs(alpha) > 1e-6 and abs(alpha - 1.0) > 1e-6:
                raise KmodelError(f'kmodel: output conv {name} has a non-linear activation table')
            alpha = 1.0
            scale = dq['scale'] * s_pos * mul
            bias = dq['scale'] * (y0 + s_pos * (add - z0)) + dq['bias']
        wf = wq * in_gain[None, :, None]
        k = c.ksize
        if c.depthwise:
            out[f'{name}/kernel'] = wf.reshape(c.out_ch, k, k).transpose(1, 2, 0)[..., None].astype(np.float32)
        else:
            out[f'{name}/kernel'] = wf.reshape(c.out_ch, c.in_ch, k, k).transpose(2, 3, 1, 0).astype(np.float32)
        if name in ('head_conv_2', 'head_conv_5'):
            # Conv2D(use_bias=True) without BN: fold the per-channel scale into the kernel
            out[f'{name}/kernel'] = (out[f'{name}/kernel'].astype(np.float64) * scale[None, None, None, :]).astype(np.float32)
            out[f'{name}/bias'] = bias.astype(np.float32)
        else:
            bn = f'{name}_bn'
            out[f'{bn}/gamma'] = (scale * np.sqrt(1.0)).astype(np.float32)
            out[f'{bn}/beta'] = bias.astype(np.float32)
            out[f'{bn}/moving_mean'] = np.zeros(c.out_ch, np.float32)
            out[f'{bn}/moving_variance'] = np.full(c.out_ch, 1.0 - eps, np.float32)   # gamma / sqrt(var + eps) = gamma
        report['layers'][name] = dict(alpha=alpha, zp_x=c.zp_x, zp_w=c.zp_w, z0=z0, y0=y0, slope=s_pos, pool=c.pool_type, pad_value=c.pad_value)
    return out, report
