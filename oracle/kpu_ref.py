"""kpu_ref.py — CPU ORACLE (test infrastructure, NOT the product path): the K210 KPU's integer pipeline for a kmodel v3, bit by bit.

Restated from the published algorithm of the un-vendored third-party runtime the reference's K210 demo links against
(`yolo3_frame_test_public/main.c:274,303`: `kpu_load_kmodel`, `kpu_run_kmodel`; Kendryte standalone SDK `kpu.c`, nncase v0.1 K210
kernels `kpu_conv2d`):
    acc = sum(x*w) + (arg_x*sum(x) >> shr_x) + (arg_w*sum(w) >> shr_w) + arg_add * in_channels_per_group     x, w uint8, pad = pad_value
    z   = (acc * bn.mul >> bn.shift) + bn.add
    y   = clamp(carry_shift((z - seg.start) * seg.mul, seg.shift) + seg.bias, 0, 255)      seg = last segment with z > start
    pooling `left_top_2_s2` (how the KPU does a stride-2 conv), DEQUANTIZE / REQUANTIZE / QUANTIZED_CONCAT / RESIZE_NEAREST / UPLOAD in
    main memory.
PARITY UNPINNED against the silicon (no K210 here, the SDK is not in /root/reference); pinned instead by the only known answer the
reference publishes for these weights: `kfpkg/dog.jpg` through this emulator + the COMPILED reference region layer gives the dog /
bicycle / car of `asset/k210_res.jpg` (tests/test_kmodel.py).  Consumes the parse of k210_yolo_framework_amd/kmodel.py (product code:
the file-format reader); everything arithmetic is here.
"""
import numpy as np

from k210_yolo_framework_amd import kmodel as km_


def _carry_shift(v: np.ndarray, s: int) -> np.ndarray:
    if s <= 0:
        return v
    v = v >> (s - 1)
    odd = (v & 1) != 0
    half = v >> 1
    return np.where(odd, np.where(v < 0, half - 1 + 1, half + 1), half)      # round half up on the shifted value (nncase carry_shift)


def conv(c: km_.ConvLayer, x: np.ndarray) -> np.ndarray:
    """x uint8 [C][H][W] -> uint8 [OC][OH][OW]"""
    C, H, W = x.shape
    assert C == c.in_ch and H == c.in_h and W == c.in_w, (c.index, x.shape, (c.in_ch, c.in_h, c.in_w))
    k = c.ksize
    p = (k - 1) // 2
    xp = np.full((C, H + 2 * p, W + 2 * p), c.pad_value, np.int64)
    xp[:, p:p + H, p:p + W] = x
    # im2col [C][k*k][H][W]
    cols = np.stack([xp[:, ky:ky + H, kx:kx + W] for ky in range(k) for kx in range(k)], 1)
    w = c.weights.astype(np.int64)                                            # [oc][ic|1][kk]
    if c.depthwise:
        sum_xw = np.einsum('ckhw,ck->chw', cols, w[:, 0, :])
        sum_x = cols.sum(1)                                                   # [C][H][W]
        g_ic = 1
    else:
        sum_xw = (w.reshape(c.out_ch, -1).astype(np.float64) @ cols.reshape(C * k * k, H * W).astype(np.float64))
        sum_xw = np.rint(sum_xw).astype(np.int64).reshape(c.out_ch, H, W)     # exact: |sum| < 2^53
        sum_x = cols.sum((0, 1))[None]                                        # [1][H][W]
        g_ic = C
    sum_w = w.reshape(c.out_ch, -1).sum(1)[:, None, None]
    acc = sum_xw + ((c.arg_x * sum_x) >> c.shr_x) + ((c.arg_w * sum_w) >> c.shr_w) + c.arg_add * g_ic
    z = ((acc * c.bn_mul[:, None, None]) >> c.bn_shift[:, None, None]) + c.bn_add[:, None, None]
    y = np.zeros_like(z)
    seg = np.zeros(z.shape, np.int64)
    for s in range(16):                                                       # last segment whose start is below z
        seg = np.where(z > c.act_start[s], s, seg)
    for s in range(16):
        m = seg == s
        if m.any():
            v = _carry_shift((z[m] - c.act_start[s]) * c.act_mul[s], int(c.act_shift[s])) + c.act_bias[s]
            y[m] = v
    y = np.clip(y, 0, 255).astype(np.uint8)
    if c.pool_type == km_.POOL_LEFT_TOP_2_S2:
        y = y[:, ::2, ::2]
    elif c.pool_type != km_.POOL_BYPASS:
        raise NotImplementedError(f'KPU pool type {c.pool_type}')
    assert y.shape == (c.out_ch, c.out_h, c.out_w), (c.index, y.shape, (c.out_ch, c.out_h, c.out_w))
    return y


def run(model: km_.Kmodel, image_chw_u8: np.ndarray, keep=None):
    """Run every layer; returns the float outputs ([C][H][W] each) and, when `keep` is a dict, every conv layer's uint8 output."""
    kpu = {}                                                                  # KPU RAM address -> tensor
    mem = {}                                                                  # main-memory address -> tensor (uint8 or float32)
    first = True
    for l in model.layers:
        if isinstance(l, km_.ConvLayer):
            x = image_chw_u8 if first else kpu[l.src_addr]
            first = False
            y = conv(l, x)
            kpu[l.dst_addr] = y
            if l.flags & km_.KLF_MAIN_MEM_OUT:
                mem[l.main_mem_out] = y
            if keep is not None:
                keep[l.index] = y
            continue
        f = l.fields
        if l.type == km_.KL_DEQUANTIZE:
            src = mem[f['src']]
            assert src.size == f['count']
            mem[f['dst']] = src.astype(np.float32) * np.float32(f['scale']) + np.float32(f['bias'])
        elif l.type == km_.KL_REQUANTIZE:
            src = mem[f['src']]
            assert src.size == f['count']
            mem[f['dst']] = f['table'][src]
        elif l.type == km_.KL_QUANTIZED_RESIZE_NN:
            src = mem[f['src']]
            assert src.shape == (f['channels'], f['in_h'], f['in_w'])
            ys = (np.arange(f['out_h']) * f['in_h']) // f['out_h']
            xs = (np.arange(f['out_w']) * f['in_w']) // f['out_w']
            mem[f['dst']] = src[:, ys][:, :, xs]
        elif l.type == km_.KL_QUANTIZED_CONCAT:
            parts = [mem[a] for a, _ in f['inputs']]
            for p_, (_, sz) in zip(parts, f['inputs']):
                assert p_.size == sz
            mem[f['dst']] = np.concatenate(parts, 0)
        elif l.type == km_.KL_K210_UPLOAD:
            src = mem[f['src']]
            assert src.shape == (f['channels'], f['height'], f['width'])
            kpu[f['kpu_addr']] = src
        else:
            raise NotImplementedError(l.type)
    return [mem[a] for a, _ in model.outputs]
