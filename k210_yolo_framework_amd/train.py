"""Training step on MI355X (keras_train.py:73-98).

The reference's `train_model.fit` runs, per step: forward in training mode (batch-statistics BatchNorm), the YOLO
loss of every output layer (tools/utils.py:708-793) plus the l2(5e-4) kernel regulariser of every DarknetConv2D
(models/yolonet.py:245-250), TF autodiff, and keras Adam(lr, decay) (keras_train.py:73-76).  All of that is
TensorFlow in the reference; here each arithmetic op is a HIP kernel of libyolo_hip.so (csrc/yk_train.hip,
csrc/yk_loss.hip) called through the C-ABI, and this module is only the tape: it walks the NetSpec forwards and
backwards.  torch supplies device buffers, views/concat (data movement) and the gradient all-reduce (RCCL).

Data-parallel: one process per GPU, each with `per_rank_batch` images.  The loss divisor is the GLOBAL batch (what
Helper.batch_size is in the single-process reference), gradients are SUM-all-reduced in one flat bucket, BatchNorm
statistics stay per replica.  With world_size 1 this is exactly the reference's step.

fp32 storage and fp32 MFMA throughout (TF1.14's default for this model)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import engine
from . import netspec as ns

L2_WEIGHT = 5e-4                 # keras.regularizers.l2(5e-4), yolonet.py:247
BN_MOMENTUM = 0.99               # keras BatchNormalization default (v1 backbone, every DarknetConv2D_BN_Leaky)
BN_MOMENTUM_V2 = 0.999           # MobileNetV2 backbone, keras_mobilenet_v2.py:320


def _is_darknet_conv(name: str) -> bool:
    """Layers built by DarknetConv2D (yolonet.py:245) carry the l2 regulariser; the MobileNet backbones do not."""
    return name.startswith('head_conv') or name.startswith('conv2d_')


class Trainer:
    def __init__(self, spec: ns.NetSpec, weights: Dict[str, np.ndarray], anchors: np.ndarray, per_rank_batch: int,
                 obj_thresh: float = 0.7, iou_thresh: float = 0.5, obj_weight: float = 1.0, noobj_weight: float = 1.0,
                 wh_weight: float = 1.0, lr: float = 5e-4, decay: float = 0.0, device: int = 0, process_group=None,
                 world_size: int = 1, use_graph: bool = True):
        import torch
        engine.require_gpu()
        self.torch = torch
        self.spec, self.B = spec, int(per_rank_batch)
        self.dev = torch.device('cuda', device)
        self.anchors = np.asarray(anchors, np.float32)
        self.hyper = dict(obj_thresh=obj_thresh, iou_thresh=iou_thresh, obj_weight=obj_weight, noobj_weight=noobj_weight,
                          wh_weight=wh_weight)
        self.lr, self.decay, self.iterations = float(lr), float(decay), 0
        self.pg, self.world = process_group, int(world_size)
        self.lay = {l.name: l for l in spec.layers}
        self.L = engine.lib()

        # ---- flat parameter / gradient / Adam buffers; conv kernels as [Cout][kh*kw*Cin], depthwise as [9][C]
        self.slots: Dict[str, tuple] = {}
        off = 0
        for l in spec.layers:
            kh, kw, ci, co = l.kernel_shape
            shape = (co, kh * kw * ci) if l.kind == 'conv' else (9, ci)
            items = [(l.name + '/kernel', shape)]
            if l.use_bias:
                items.append((l.name + '/bias', (co,)))
            if l.bn_name:
                c = co if l.kind == 'conv' else ci
                items += [(l.bn_name + '/gamma', (c,)), (l.bn_name + '/beta', (c,))]
            for nm, shp in items:
                n = int(np.prod(shp))
                self.slots[nm] = (off, shp)
                off += (n + 7) // 8 * 8
        self.n_params = off
        self.P = torch.zeros(off, dtype=torch.float32, device=self.dev)
        self.G = torch.zeros_like(self.P)
        self.m = torch.zeros_like(self.P)
        self.v = torch.zeros_like(self.P)
        self.moving: Dict[str, "torch.Tensor"] = {}
        self.load_weights(weights)
        self.counts = [torch.zeros(3, dtype=torch.float32, device=self.dev) for _ in spec.outputs]
        self.saved: Dict[int, dict] = {}
        self.T: Dict[int, "torch.Tensor"] = {}
        # forward + loss + backward is ~800 launches whose arguments never change from step to step: after one eager step it is
        # captured as a HIP graph and replayed (same kernels, same order, same buffers -> bitwise the same results; measured 7.8 ->
        # 6.7 ms per step for yolo_mobilev2-1.0 at 16 images).  Adam stays outside: its step counter is a launch argument.
        self.use_graph = bool(use_graph)
        # weight gradients (and the regulariser's dot products) depend on nothing the backward chain waits for: they can run on a second
        # stream - inside a captured step a parallel branch of the graph - joined before the exchange / update (~170 of the step's ~700 launches
        # off the critical path).  A cross-stream edge inside a replayed hipGraph is expensive on this runtime: one fork per layer (~120 edges)
        # measured 5.59 -> 10.37 ms per step (gpurun_out/r5c3), so the deferred launches are flushed to the side stream in
        # YK_TRAIN_WSTREAM = N chunks (N forks + one join; 0 = everything on one stream, round 4's form).
        import os as _os
        self.wgrad_chunks = max(0, int(_os.environ.get('YK_TRAIN_WSTREAM', '0') or 0))
        self.wgrad_stream = self.wgrad_chunks > 0
        # ... what does pay (round 6): the 1x1 convs' weight gradients are independent GEMMs of 1-30 tiles each; collected during the backward
        # walk and issued as ONE grouped launch at its end (yk_gemm_f32_grouped: bitwise the same results), their tiles fill the chip
        # together - 35 GEMMs + 35 slice-adding launches become 2, the depthwise weight gradients (17 + 17) 2 more.  YK_TRAIN_GROUP_WGRAD=0: one launch
        # per layer as before.
        self.group_wgrad = (_os.environ.get('YK_TRAIN_GROUP_WGRAD', '1') or '1') != '0' and not self.wgrad_stream
        self.implicit3x3 = (_os.environ.get('YK_TRAIN_IMPLICIT3X3', '1') or '1') != '0'     # 3x3 convs as implicit GEMMs (0: im2col + GEMM + col2im, rounds 2-5)
        self._ws = None
        self._l2_seg = None
        self._fa = None
        self._graph = None
        self._gx = self._gy = self._gres = self._side = None
        self._eager_steps = 0
        self._captured_key = None

    # ------------------------------------------------------------------ parameters
    def view(self, buf, name):
        off, shp = self.slots[name]
        return buf[off:off + int(np.prod(shp))].view(*shp)

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        torch = self.torch
        for l in self.spec.layers:
            k = np.asarray(weights[l.name + '/kernel'], np.float32)
            if l.kind == 'conv':
                k = np.transpose(k, (3, 0, 1, 2)).reshape(l.kernel_shape[3], -1)          # HWIO -> O,(HWI)
            else:
                k = k[..., 0].reshape(9, -1)
            self.view(self.P, l.name + '/kernel').copy_(torch.from_numpy(np.ascontiguousarray(k)))
            if l.use_bias:
                self.view(self.P, l.name + '/bias').copy_(torch.from_numpy(np.asarray(weights[l.name + '/bias'], np.float32)))
            if l.bn_name:
                for s in ('/gamma', '/beta'):
                    self.view(self.P, l.bn_name + s).copy_(torch.from_numpy(np.asarray(weights[l.bn_name + s], np.float32)))
                for s in ('/moving_mean', '/moving_variance'):
                    src = torch.from_numpy(np.asarray(weights[l.bn_name + s], np.float32).copy())
                    cur = self.moving.get(l.bn_name + s)
                    if cur is None:
                        self.moving[l.bn_name + s] = src.to(self.dev)
                    else:
                        # IN PLACE: a captured step holds the raw pointer of this tensor (yk_bn_train_fwd updates the moving statistics
                        # inside the replayed graph); rebinding the name would leave the graph writing a freed buffer while
                        # export_weights / validate read a tensor nobody updates
                        cur.copy_(src)

    def export_weights(self) -> Dict[str, np.ndarray]:
        """Keras-layout arrays again (what keras.models.save_model would hold, keras_train.py:107)."""
        out = {}
        for l in self.spec.layers:
            kh, kw, ci, co = l.kernel_shape
            k = self.view(self.P, l.name + '/kernel').cpu().numpy()
            out[l.name + '/kernel'] = (np.transpose(k.reshape(co, kh, kw, ci), (1, 2, 3, 0)) if l.kind == 'conv'
                                       else k.reshape(3, 3, ci)[..., None]).copy()
            if l.use_bias:
                out[l.name + '/bias'] = self.view(self.P, l.name + '/bias').cpu().numpy().copy()
            if l.bn_name:
                for s in ('/gamma', '/beta'):
                    out[l.bn_name + s] = self.view(self.P, l.bn_name + s).cpu().numpy().copy()
                for s in ('/moving_mean', '/moving_variance'):
                    out[l.bn_name + s] = self.moving[l.bn_name + s].cpu().numpy().copy()
        return out

    def grads(self) -> Dict[str, np.ndarray]:
        """Gradients of the last step in Keras layout (for parity tests)."""
        out = {}
        for nm, (off, shp) in self.slots.items():
            g = self.G[off:off + int(np.prod(shp))].view(*shp).cpu().numpy()
            if nm.endswith('/kernel'):
                l = self.lay[nm[:-7]]
                kh, kw, ci, co = l.kernel_shape
                g = np.transpose(g.reshape(co, kh, kw, ci), (1, 2, 3, 0)) if l.kind == 'conv' else g.reshape(3, 3, ci)[..., None]
            out[nm] = g.copy()
        return out

    # ------------------------------------------------------------------ kernel calls
    def _s(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _ck(self, rc, what):
        engine._check(rc, what)

    def _new(self, *shape):
        return self.torch.empty(shape, dtype=self.torch.float32, device=self.dev)

    def gemm(self, tA, tB, M, N, K, A, lda, Bm, ldb, Cm, ldc, alpha=1.0, beta=0.0):
        self._ck(self.L.yk_gemm_f32(C.c_int(tA), C.c_int(tB), C.c_int(M), C.c_int(N), C.c_int(K), C.c_float(alpha), engine._ptr(A),
                                    C.c_int(lda), engine._ptr(Bm), C.c_int(ldb), C.c_float(beta), engine._ptr(Cm), C.c_int(ldc),
                                    self._s()), 'yk_gemm_f32')

    def _geom(self, op):
        hi, wi, ci = self.spec.tensors[op['in0']]
        ho, wo, _ = self.spec.tensors[op['out']]
        return [C.c_int(v) for v in (self.B, hi, wi, ci, ho, wo, op['stride'], op['pad_t'], op['pad_l'])]

    def _axpy(self, a, x, y):
        self._ck(self.L.yk_axpy_f32(C.c_longlong(x.numel()), C.c_float(a), engine._ptr(x), engine._ptr(y), self._s()), 'yk_axpy_f32')

    def _fused_adds(self):
        """conv op index -> (index of the Add that consumes ONLY-there its output, the Add's other input).  The Add is folded into that
        conv's BatchNorm apply pass when the conv has BatchNorm, its output has no other reader and is not a network output, and the Add
        follows before anything else needs the output (residual blocks: keras_mobilenet_v2.py:483-484, yolonet.py:194-204)."""
        if self._fa is None:
            ops, fa = self.spec.ops, {}
            readers = {}
            for k, o in enumerate(ops):
                for key in ('in0', 'in1'):
                    if o.get(key, -1) is not None and o.get(key, -1) >= 0:
                        readers.setdefault(o[key], []).append(k)
            for k, o in enumerate(ops):
                if o['type'] != ns.OP_ADD:
                    continue
                for mine, other in ((o['in1'], o['in0']), (o['in0'], o['in1'])):
                    prod = [j for j, q in enumerate(ops) if q['out'] == mine and q['type'] in (ns.OP_CONV, ns.OP_DWCONV)]
                    if (len(prod) == 1 and self.lay[ops[prod[0]]['layer']].bn_name and readers.get(mine, []) == [k] and mine not in self.spec.outputs
                            and mine != other and all(q['out'] != other for q in ops[prod[0]:k])):
                        fa[prod[0]] = (k, other)
                        break
            self._fa = fa
        return self._fa

    # ------------------------------------------------------------------ forward (training mode)
    def forward(self, x_nhwc) -> List["torch.Tensor"]:
        """x_nhwc: cuda fp32 [B,H,W,3] already normalised (Helper._process_img output).  Saves the tape."""
        torch = self.torch
        assert x_nhwc.is_cuda and x_nhwc.dtype == torch.float32 and tuple(x_nhwc.shape[:1]) == (self.B,)
        T, S = {0: x_nhwc.contiguous()}, {}
        fused_add = self._fused_adds()                               # conv op index -> (Add op index, the Add's other input)
        done = set()
        for i, op in enumerate(self.spec.ops):
            if i in done:
                continue
            x, t = T[op['in0']], op['type']
            ho, wo, co = self.spec.tensors[op['out']]
            M = self.B * ho * wo
            if t in (ns.OP_CONV, ns.OP_DWCONV):
                l = self.lay[op['layer']]
                w = self.view(self.P, l.name + '/kernel')
                z = self._new(self.B, ho, wo, co)
                a = kk = None
                implicit = False                                   # 3x3 conv as an implicit GEMM: no column matrix (needs Cin % 4 == 0: not the 3-channel stem)
                if t == ns.OP_CONV:
                    ci, k = op['cin'], op['k']
                    if k == 1 and op['stride'] == 1:
                        a, kk = x, ci
                    else:
                        assert k == 3
                        implicit = self.implicit3x3 and ci % 4 == 0 and bool(l.bn_name)
                        if not implicit:
                            a, kk = self._new(M, 9 * ci), 9 * ci
                            self._ck(self.L.yk_im2col3x3_f32(engine._ptr(x), *self._geom(op), engine._ptr(a), self._s()), 'yk_im2col3x3_f32')
                if l.bn_name:
                    # convolution + batch statistics + apply in one library call: the producer of z leaves the partial sums of the
                    # statistics (yk_gemm_bn_fwd_f32 / yk_dw3x3_bn_fwd_f32), z is not read a second time for them
                    y = self._new(self.B, ho, wo, co)
                    mean, invstd = self._new(co), self._new(co)
                    fa = fused_add.get(i)
                    res = T[fa[1]] if fa is not None else None       # keras Add()([res, this output]) folded into the apply pass
                    bn = (engine._ptr(z), engine._ptr(self.view(self.P, l.bn_name + '/gamma')),
                          engine._ptr(self.view(self.P, l.bn_name + '/beta')), C.c_float(ns.BN_EPS), C.c_int(op['act']),
                          C.c_float(op['alpha']), engine._ptr(y), engine._ptr(mean), engine._ptr(invstd),
                          engine._ptr(self.moving[l.bn_name + '/moving_mean']), engine._ptr(self.moving[l.bn_name + '/moving_variance']),
                          C.c_float(BN_MOMENTUM_V2 if self.spec.name == 'yolo_mobilev2' and not _is_darknet_conv(l.name) else BN_MOMENTUM),
                          engine._ptr(res) if res is not None else None, self._s())
                    if implicit:
                        self._ck(self.L.yk_conv3x3_bn_fwd_f32(engine._ptr(x), engine._ptr(w), *self._geom(op), C.c_int(co), *bn), 'yk_conv3x3_bn_fwd_f32')
                    elif t == ns.OP_CONV:
                        self._ck(self.L.yk_gemm_bn_fwd_f32(C.c_int(M), C.c_int(co), C.c_int(kk), engine._ptr(a), C.c_int(kk), engine._ptr(w),
                                                           C.c_int(kk), *bn), 'yk_gemm_bn_fwd_f32')           # Z = X * W^T
                    else:
                        self._ck(self.L.yk_dw3x3_bn_fwd_f32(engine._ptr(x), engine._ptr(w), *self._geom(op), *bn), 'yk_dw3x3_bn_fwd_f32')
                    S[i] = dict(z=z, mean=mean, invstd=invstd)
                    if fa is not None:                               # y IS the Add's output; the conv's own output tensor is never needed again
                        T[self.spec.ops[fa[0]]['out']] = y
                        done.add(fa[0])
                else:
                    assert op['act'] == ns.ACT_NONE
                    if t == ns.OP_CONV:
                        self.gemm(0, 1, M, co, kk, a, kk, w, kk, z, co)                  # Z = X * W^T
                    else:
                        self._ck(self.L.yk_dw3x3_fwd_f32(engine._ptr(x), engine._ptr(w), *self._geom(op), engine._ptr(z), self._s()),
                                 'yk_dw3x3_fwd_f32')
                    if l.use_bias:
                        self._ck(self.L.yk_bias_add_f32(engine._ptr(z), C.c_longlong(M), C.c_int(co),
                                                        engine._ptr(self.view(self.P, l.name + '/bias')), self._s()), 'yk_bias_add_f32')
                    y = z
            elif t == ns.OP_MAXPOOL:
                hi, wi, ci = self.spec.tensors[op['in0']]
                y = self._new(self.B, ho, wo, co)
                arg = torch.empty((self.B, ho, wo, co), dtype=torch.uint8, device=self.dev)
                self._ck(self.L.yk_maxpool2_fwd_f32(engine._ptr(x), C.c_int(self.B), C.c_int(hi), C.c_int(wi), C.c_int(ci), C.c_int(ho),
                                                    C.c_int(wo), C.c_int(op['stride']), engine._ptr(y), engine._ptr(arg), self._s()),
                         'yk_maxpool2_fwd_f32')
                S[i] = dict(arg=arg)
            elif t == ns.OP_UPSAMPLE:                                                   # nearest x2: pure data movement
                y = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
            elif t == ns.OP_CONCAT:
                y = torch.cat([x, T[op['in1']]], dim=3)
            elif t == ns.OP_ADD:
                y = x.clone()
                self._axpy(1.0, T[op['in1']], y)
            else:
                raise engine.YkError(f'op type {t} not trainable')
            T[op['out']] = y
        self.T, self.saved = T, S
        e = 5 + self.spec.class_num
        return [T[o].view(self.B, *self.spec.tensors[o][:2], self.spec.anchor_num, e) for o in self.spec.outputs]

    # ------------------------------------------------------------------ backward
    def _wstream(self):
        """The weight-gradient stream (created once; its split-K / reduction scratch is keyed by the stream, so it never collides with the
        main chain's)."""
        if self._ws is None:
            self._ws = self.torch.cuda.Stream(device=self.dev)
        return self._ws

    def backward(self, out_grads: Sequence["torch.Tensor"]) -> None:
        """out_grads[i] = dL/d(output i).  Fills self.G (kernel/bias/gamma/beta gradients; regulariser added by step())."""
        torch = self.torch
        self.G.zero_()
        D: Dict[int, "torch.Tensor"] = {}
        main = torch.cuda.current_stream()
        ws = self._wstream() if self.wgrad_stream else None
        keep = []                                                   # operands the side stream reads: alive until the join below
        if ws is not None:
            ws.wait_stream(main)                                    # G is zero, the forward tape is complete

        grouped = []                                                # (dz, x, gw, co, ci, M) of the 1x1 convs: one grouped launch at the end
        grouped_dw = []                                             # (x, dz, gw, geometry) of the depthwise convs: likewise
        pending = []                                                # deferred weight-gradient launches of the current chunk
        n_conv = sum(1 for o in self.spec.ops if o['type'] in (ns.OP_CONV, ns.OP_DWCONV))
        per_chunk = max(1, -(-n_conv // max(1, self.wgrad_chunks)))
        seen = [0]

        def flush():
            """Issue the chunk's deferred launches on the side stream, behind everything the main stream has been given so far."""
            if ws is None or not pending:
                return
            ev = torch.cuda.Event()
            ev.record(main)
            ws.wait_event(ev)
            with torch.cuda.stream(ws):
                for fn in pending:
                    fn()
            pending.clear()

        def on_side(fn, *tensors):
            """fn() only WRITES weight gradients: run it now (no side stream) or defer it to the chunk's flush."""
            if ws is None:
                return fn()
            pending.append(fn)
            keep.extend(tensors)

        def acc(tid, g, own):
            if tid == 0:
                return
            if tid not in D:
                D[tid] = g if own else g.clone()
            else:
                self._axpy(1.0, g, D[tid])

        for o, g in zip(self.spec.outputs, out_grads):
            acc(o, g.reshape(self.B, *self.spec.tensors[o]).contiguous(), False)
        for i in range(len(self.spec.ops) - 1, -1, -1):
            op = self.spec.ops[i]
            t = op['type']
            dy = D.pop(op['out'], None)
            if dy is None:
                continue
            x = self.T.get(op['in0'])                               # (None for an Add whose conv input was folded into the producer: never stored, never read here)
            ho, wo, co = self.spec.tensors[op['out']]
            hi, wi, ci = self.spec.tensors[op['in0']]
            M = self.B * ho * wo
            if t in (ns.OP_CONV, ns.OP_DWCONV):
                seen[0] += 1
                if seen[0] % per_chunk == 0:
                    flush()
                l = self.lay[op['layer']]
                w = self.view(self.P, l.name + '/kernel')
                gw = self.view(self.G, l.name + '/kernel')
                if l.bn_name:
                    sv = self.saved[i]
                    dz = self._new(self.B, ho, wo, co)
                    self._ck(self.L.yk_bn_train_bwd_f32(
                        engine._ptr(sv['z']), engine._ptr(dy), C.c_longlong(M), C.c_int(co),
                        engine._ptr(self.view(self.P, l.bn_name + '/gamma')), engine._ptr(self.view(self.P, l.bn_name + '/beta')),
                        engine._ptr(sv['mean']), engine._ptr(sv['invstd']), C.c_int(op['act']), C.c_float(op['alpha']), engine._ptr(dz),
                        engine._ptr(self.view(self.G, l.bn_name + '/gamma')), engine._ptr(self.view(self.G, l.bn_name + '/beta')),
                        self._s()), 'yk_bn_train_bwd_f32')
                else:
                    dz = dy
                    if l.use_bias:
                        gb = self.view(self.G, l.name + '/bias')
                        on_side(lambda dz=dz, gb=gb, M=M, co=co: self._ck(self.L.yk_colsum_f32(engine._ptr(dz), C.c_longlong(M), C.c_int(co), engine._ptr(gb),
                                                                                                self._s()), 'yk_colsum_f32'), dz)
                need_dx = op['in0'] != 0
                if t == ns.OP_CONV:
                    k = op['k']
                    if k == 1 and op['stride'] == 1:
                        if self.group_wgrad:
                            grouped.append((dz, x, gw, co, ci, M))                       # dW = dZ^T * X, with all the others below
                        else:
                            on_side(lambda dz=dz, x=x, gw=gw, co=co, ci=ci, M=M: self.gemm(1, 0, co, ci, M, dz, co, x, ci, gw, ci), dz)     # dW = dZ^T * X
                        if need_dx:
                            if op['in0'] in D:                                          # a second reader of the input: dX += dZ * W straight into the
                                self.gemm(0, 0, M, ci, co, dz, co, w, ci, D[op['in0']], ci, beta=1.0)   # accumulated gradient (no temporary, no axpy)
                            else:
                                dx = self._new(self.B, hi, wi, ci)
                                self.gemm(0, 0, M, ci, co, dz, co, w, ci, dx, ci)       # dX = dZ * W
                                acc(op['in0'], dx, True)
                    elif self.implicit3x3 and ci % 4 == 0 and co % 4 == 0:
                        geom = self._geom(op)
                        on_side(lambda x=x, dz=dz, gw=gw, geom=geom, co=co: self._ck(self.L.yk_conv3x3_bwd_weight_f32(
                            engine._ptr(x), engine._ptr(dz), *geom, C.c_int(co), engine._ptr(gw), self._s()), 'yk_conv3x3_bwd_weight_f32'), dz)
                        if need_dx:
                            dx = self._new(self.B, hi, wi, ci)
                            if op['stride'] == 1:
                                self._ck(self.L.yk_conv3x3_bwd_data_f32(engine._ptr(dz), engine._ptr(w), *geom, C.c_int(co), engine._ptr(dx), self._s()),
                                         'yk_conv3x3_bwd_data_f32')
                            else:                                           # strided: column matrix of gradients, folded by col2im
                                kk = 9 * ci
                                col = self._new(M, kk)
                                self.gemm(0, 0, M, kk, co, dz, co, w, kk, col, kk)
                                self._ck(self.L.yk_col2im3x3_f32(engine._ptr(col), *geom, engine._ptr(dx), self._s()), 'yk_col2im3x3_f32')
                                del col
                            acc(op['in0'], dx, True)
                    else:
                        kk = 9 * ci
                        col = self._new(M, kk)
                        self._ck(self.L.yk_im2col3x3_f32(engine._ptr(x), *self._geom(op), engine._ptr(col), self._s()), 'yk_im2col3x3_f32')
                        self.gemm(1, 0, co, kk, M, dz, co, col, kk, gw, kk)            # (main stream: `col` is overwritten right below)
                        if need_dx:
                            self.gemm(0, 0, M, kk, co, dz, co, w, kk, col, kk)          # dcol (reuses the buffer)
                            dx = self._new(self.B, hi, wi, ci)
                            self._ck(self.L.yk_col2im3x3_f32(engine._ptr(col), *self._geom(op), engine._ptr(dx), self._s()),
                                     'yk_col2im3x3_f32')
                            acc(op['in0'], dx, True)
                        del col
                else:
                    geom = self._geom(op)
                    if self.group_wgrad:
                        grouped_dw.append((x, dz, gw, [g.value for g in geom]))
                    else:
                        on_side(lambda x=x, dz=dz, gw=gw, geom=geom: self._ck(self.L.yk_dw3x3_bwd_weight_f32(engine._ptr(x), engine._ptr(dz), *geom, engine._ptr(gw),
                                                                                                              self._s()), 'yk_dw3x3_bwd_weight_f32'), dz)
                    if need_dx:
                        dx = self._new(self.B, hi, wi, ci)
                        self._ck(self.L.yk_dw3x3_bwd_data_f32(engine._ptr(dz), engine._ptr(w), *self._geom(op), engine._ptr(dx), self._s()),
                                 'yk_dw3x3_bwd_data_f32')
                        acc(op['in0'], dx, True)
            elif t == ns.OP_MAXPOOL:
                dx = self._new(self.B, hi, wi, ci)
                self._ck(self.L.yk_maxpool2_bwd_f32(engine._ptr(dy), engine._ptr(self.saved[i]['arg']), C.c_int(self.B), C.c_int(hi),
                                                    C.c_int(wi), C.c_int(ci), C.c_int(ho), C.c_int(wo), C.c_int(op['stride']),
                                                    engine._ptr(dx), self._s()), 'yk_maxpool2_bwd_f32')
                acc(op['in0'], dx, True)
            elif t == ns.OP_UPSAMPLE:
                dx = self._new(self.B, hi, wi, ci)
                self._ck(self.L.yk_upsample2x_bwd_f32(engine._ptr(dy), C.c_int(self.B), C.c_int(hi), C.c_int(wi), C.c_int(ci),
                                                      engine._ptr(dx), self._s()), 'yk_upsample2x_bwd_f32')
                acc(op['in0'], dx, True)
            elif t == ns.OP_CONCAT:
                c0 = self.spec.tensors[op['in0']][2]
                acc(op['in0'], dy[..., :c0].contiguous(), True)
                acc(op['in1'], dy[..., c0:].contiguous(), True)
            elif t == ns.OP_ADD:
                # dL/d(in0) = dL/d(in1) = dy.  The two entries may SHARE dy's buffer (no clone) when nothing accumulates into either of them
                # before the other one has been consumed: every other reader of one input precedes the producer of the other input, i.e.
                # the backward loop pops the producer's entry first (the residual blocks of both networks)
                a_, b_ = op['in0'], op['in1']
                pa = max([j for j, q in enumerate(self.spec.ops) if q['out'] == a_], default=-1)
                pb = max([j for j, q in enumerate(self.spec.ops) if q['out'] == b_], default=-1)
                first, second = (a_, b_) if pa > pb else (b_, a_)     # `first` is popped first (its producer comes later in the forward order)
                pf = max(pa, pb)
                others = [j for j, q in enumerate(self.spec.ops) if j != i and (q.get('in0') == second or q.get('in1') == second)]
                lone = not [j for j, q in enumerate(self.spec.ops) if j != i and (q.get('in0') == first or q.get('in1') == first)] and first not in self.spec.outputs
                # ... and `first`'s producer must be a conv WITH BatchNorm: its backward only READS dy and writes a fresh dz.  Another Add would store
                # dy again by reference, a conv without BN hands dy on as dz (and, with a weight-gradient stream, reads it later) - then a later
                # accumulation into D[second] would corrupt a buffer somebody else still holds: clone.
                pq = self.spec.ops[pf] if pf >= 0 else None
                fresh = pq is not None and pq['type'] in (ns.OP_CONV, ns.OP_DWCONV) and bool(self.lay[pq['layer']].bn_name)
                share = a_ != b_ and lone and fresh and first not in D and second not in D and all(j < pf for j in others) and second != 0
                acc(second, dy, share)
                acc(first, dy, True)
        if grouped:
            n = len(grouped)
            ia = lambda v: (C.c_int * n)(*v)
            pa = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            self._ck(self.L.yk_gemm_f32_grouped(C.c_int(n), C.c_int(1), C.c_int(0), ia([g[3] for g in grouped]), ia([g[4] for g in grouped]),
                                                ia([g[5] for g in grouped]), C.c_float(1.0), pa([g[0] for g in grouped]), ia([g[3] for g in grouped]),
                                                pa([g[1] for g in grouped]), ia([g[4] for g in grouped]), C.c_float(0.0), pa([g[2] for g in grouped]),
                                                ia([g[4] for g in grouped]), self._s()), 'yk_gemm_f32_grouped')
        if grouped_dw:
            n = len(grouped_dw)
            pa = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            geo = (C.c_int * (9 * n))(*[v for g in grouped_dw for v in g[3]])
            self._ck(self.L.yk_dw3x3_bwd_weight_grouped_f32(C.c_int(n), pa([g[0] for g in grouped_dw]), pa([g[1] for g in grouped_dw]), geo,
                                                            pa([g[2] for g in grouped_dw]), self._s()), 'yk_dw3x3_bwd_weight_grouped_f32')
        del grouped, grouped_dw
        flush()
        if ws is not None:
            main.wait_stream(ws)                                    # every weight gradient is in G
        del keep

    # ------------------------------------------------------------------ one optimisation step
    def regulariser(self, add_grad: bool, value: bool = True) -> "torch.Tensor":
        """sum over DarknetConv2D kernels of 5e-4 * sum(w^2) (device scalar; value=False skips it); optionally G += 2*5e-4*W.
        One pass over all of those kernels (yk_l2_segments_f32: they are segments of the flat parameter buffer) - two launches instead of
        a dot product and an axpy per layer."""
        torch = self.torch
        if self._l2_seg is None:
            segs = [(self.slots[l.name + '/kernel'][0], int(np.prod(self.slots[l.name + '/kernel'][1])))
                    for l in self.spec.layers if l.kind == 'conv' and _is_darknet_conv(l.name)]
            pre = np.concatenate([[0], np.cumsum([n for _, n in segs])]).astype(np.int64)
            self._l2_seg = (torch.from_numpy(pre).to(self.dev), torch.from_numpy(np.asarray([o for o, _ in segs], np.int64)).to(self.dev), len(segs),
                            int(pre[-1]))
        pre, off, nseg, total = self._l2_seg
        tot = torch.zeros(1, dtype=torch.float32, device=self.dev) if value else None
        if nseg and (value or add_grad):
            self._ck(self.L.yk_l2_segments_f32(engine._ptr(self.P), engine._ptr(self.G), engine._ptr(pre), engine._ptr(off), C.c_int(nseg),
                                               C.c_longlong(total), C.c_float(L2_WEIGHT), C.c_int(1 if value else 0), C.c_int(1 if add_grad else 0),
                                               engine._ptr(tot) if value else None, self._s()), 'yk_l2_segments_f32')
        return tot

    def loss_and_grads(self, x_nhwc, y_true: Sequence["torch.Tensor"]):
        """forward + loss + backward (no all-reduce, no update).  -> dict of device scalars."""
        torch = self.torch
        global_batch = self.B * self.world
        reg = None
        if self.wgrad_stream:
            # the regulariser's value reads the weights only: its ~35 small launches run beside the forward pass (joined by backward())
            main, ws = torch.cuda.current_stream(), self._wstream()
            ws.wait_stream(main)
            with torch.cuda.stream(ws):
                reg = self.regulariser(add_grad=False)
        preds = self.forward(x_nhwc)
        parts, grads = [], []
        for li, (yp, yt) in enumerate(zip(preds, y_true)):
            loss6, g, _ = engine.yolo_loss(yt, yp.contiguous(), self.anchors[li], batch_size=global_batch, counts=self.counts[li],
                                           **self.hyper)
            parts.append(loss6)
            grads.append(g)
        self.backward(grads)
        if reg is None:
            reg = self.regulariser(add_grad=self.world == 1)
        elif self.world == 1:
            self.regulariser(add_grad=True, value=False)
        return dict(layers=parts, reg=reg)

    def invalidate_graph(self) -> None:
        """Forget the captured step (the next two steps run eagerly / re-capture).  Called automatically when something a capture
        bakes in as a launch scalar or a shape changes: `tr.hyper[...]`, the batch shape.  Parameters, Adam state and BatchNorm moving
        statistics live in buffers that are only ever written in place, so loading weights does not need it."""
        self._graph = None
        self._gres = self._gx = self._gy = None
        self._eager_steps = 0

    def _graph_key(self, x_nhwc, y_true):
        return (tuple(sorted(self.hyper.items())), tuple(x_nhwc.shape), tuple(tuple(y.shape) for y in y_true), self.world)

    def _loss_and_grads_replayed(self, x_nhwc, y_true):
        """loss_and_grads through a captured HIP graph (after one eager step that also warms allocator and scratch buffers).
        One Trainer per stream: the library's split-K workspace is keyed by the stream handle and is part of the capture."""
        torch = self.torch
        if not self.use_graph:
            return self.loss_and_grads(x_nhwc, y_true)
        key = self._graph_key(x_nhwc, y_true)
        if self._graph is not None and key != self._captured_key:
            self.invalidate_graph()                                # loss weights / thresholds / shapes are launch arguments of the capture
        if self._graph is None:
            # both the eager first step and the capture run on one dedicated stream: the library's scratch buffers are keyed by
            # stream and must already have their final size when the capture starts (an allocation would invalidate it)
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.dev)
            side, cur = self._side, torch.cuda.current_stream()
            side.wait_stream(cur)
            if self._eager_steps < 1:
                self._eager_steps += 1
                with torch.cuda.stream(side):
                    r = self.loss_and_grads(x_nhwc, y_true)
                cur.wait_stream(side)
                return r
            self._gx = x_nhwc.clone()
            self._gy = [y.clone() for y in y_true]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                g = torch.cuda.CUDAGraph()
                # thread_local: the input pipeline's producer thread allocates pinned / device buffers and copies while this thread
                # captures; in the default (global) mode any such call from another thread invalidates the capture
                with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                    self._gres = self.loss_and_grads(self._gx, self._gy)
            cur.wait_stream(side)
            self._graph = g
            self._captured_key = key
            # the capture itself does not execute: fall through to a replay with the current batch
        self._gx.copy_(x_nhwc)
        for d, s_ in zip(self._gy, y_true):
            d.copy_(s_)
        self._graph.replay()
        return self._gres

    def exchange(self, reduce=None) -> None:
        """The data-parallel exchange of one step: SUM the flat gradient bucket over the ranks, then add the regulariser's gradient
        (identical on every rank) once.  `reduce(flat_grad)` replaces the all-reduce — tests drive several replicas of one process
        through exactly this code with it."""
        import torch.distributed as dist
        from .shard import allreduce_gradients
        # data-term gradients already carry 1/global_batch, so SUM over ranks is the global-batch gradient
        if reduce is not None:
            reduce(self.G)
        else:
            if not dist.is_initialized():
                raise engine.YkError(f'Trainer(world_size={self.world}) needs an initialised torch.distributed process group')
            allreduce_gradients(self.G, dist, self.pg)
        self.regulariser(add_grad=True)

    def apply_update(self) -> None:
        """keras Adam(lr, decay) on the flat buffers (keras_train.py:73-76); one launch."""
        self._ck(self.L.yk_adam_f32(C.c_longlong(self.n_params), engine._ptr(self.P), engine._ptr(self.G), engine._ptr(self.m),
                                    engine._ptr(self.v), C.c_float(self.lr), C.c_float(self.decay), C.c_longlong(self.iterations),
                                    C.c_float(0.9), C.c_float(0.999), C.c_float(1e-7), C.c_float(1.0), self._s()), 'yk_adam_f32')
        self.iterations += 1

    def step(self, x_nhwc, y_true: Sequence["torch.Tensor"], reduce=None, reduce_scalar=None) -> Dict[str, float]:
        """model.fit's inner step (keras_train.py:94).  Returns python floats (one device->host sync)."""
        torch = self.torch
        r = self._loss_and_grads_replayed(x_nhwc, y_true)
        if self.world > 1:
            self.exchange(reduce)
        self.apply_update()
        data = torch.stack([p[0] for p in r['layers']]).sum()
        if self.world > 1:
            if reduce_scalar is not None:
                reduce_scalar(data)
            else:
                import torch.distributed as dist
                dist.all_reduce(data, op=dist.ReduceOp.SUM, group=self.pg)
        vals = torch.cat([data.view(1), r['reg'].view(1)]).cpu().numpy()
        return dict(loss=float(vals[0] + vals[1]), data_loss=float(vals[0]), reg_loss=float(vals[1]))

    def precision_recall(self):
        """Yolo_Precision / Yolo_Recall running values per output layer (tools/custom.py:42-44,74-75)."""
        out = []
        for c in self.counts:
            tp, fp, fn = c.cpu().numpy().tolist()
            out.append((tp / (tp + fp) if tp + fp else 0.0, tp / (tp + fn) if tp + fn else 0.0))
        return out
