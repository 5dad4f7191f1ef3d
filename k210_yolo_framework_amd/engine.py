"""ctypes binding of libyolo_hip.so (include/yolo_hip.h) — the ONLY compute path of this package.

There is no CPU fallback: if the library is missing, or no HIP device is visible, every entry point
raises.  PyTorch-ROCm is used for plumbing only (device buffers, streams, torch.distributed); it is
imported before the library so that both share one HIP runtime instance (libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import netspec as ns

# YK_LIB_PATH: the developer build of the same library (`make -C csrc dev`: tuning switches compiled in), for tools/ only
LIB_PATH = Path(os.environ.get('YK_LIB_PATH') or Path(__file__).resolve().parent / 'csrc' / 'libyolo_hip.so')
YK_MAX_LAYERS, YK_MAX_ANCHORS = 4, 8
PRECISIONS = {'f16': 0, 'f16x2': 1}          # YK_PRECISION_F16 / YK_PRECISION_F16X2
SCHEDULES = {'throughput': 0x000, 'latency': 0x100}      # YK_SCHEDULE_* (include/yolo_hip.h): per-layer launches | per-image cluster launches
_lib: Optional[C.CDLL] = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class YkError(RuntimeError):
    pass


class DecodeCfg(C.Structure):
    """yk_decode_cfg_t"""
    _fields_ = [('n_layers', C.c_int32), ('anchor_num', C.c_int32), ('class_num', C.c_int32),
                ('in_h', C.c_int32), ('in_w', C.c_int32),
                ('out_h', C.c_int32 * YK_MAX_LAYERS), ('out_w', C.c_int32 * YK_MAX_LAYERS),
                ('anchors', (C.c_float * 2) * YK_MAX_ANCHORS * YK_MAX_LAYERS)]


class RegionCfg(C.Structure):
    """yk_region_cfg_t"""
    _fields_ = [('layer_w', C.c_int32), ('layer_h', C.c_int32), ('anchor_num', C.c_int32), ('classes', C.c_int32),
                ('net_w', C.c_int32), ('net_h', C.c_int32), ('image_w', C.c_int32), ('image_h', C.c_int32),
                ('threshold', C.c_float), ('nms_value', C.c_float), ('anchor', C.c_float * (2 * YK_MAX_ANCHORS)),
                ('stride_b', C.c_int64), ('stride_n', C.c_int64), ('stride_e', C.c_int64),
                ('stride_y', C.c_int64), ('stride_x', C.c_int64)]


class LossCfg(C.Structure):
    """yk_loss_cfg_t"""
    _fields_ = [('out_h', C.c_int32), ('out_w', C.c_int32), ('anchor_num', C.c_int32), ('class_num', C.c_int32),
                ('anchors', (C.c_float * 2) * YK_MAX_ANCHORS), ('obj_thresh', C.c_float), ('iou_thresh', C.c_float),
                ('obj_weight', C.c_float), ('noobj_weight', C.c_float), ('wh_weight', C.c_float), ('batch_size', C.c_int32)]


def library_path() -> Path:
    return LIB_PATH


def lib() -> C.CDLL:
    """Load libyolo_hip.so (after torch, so the HIP runtime is shared).  Raises if absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise YkError(f'{LIB_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                          f'(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        import torch  # noqa: F401  (binds libamdhip64.so.7 first)
        L = C.CDLL(str(LIB_PATH))
        L.yk_last_error.restype = C.c_char_p
        L.yk_device_count.restype = C.c_int
        for fn in ('yk_plan_create', 'yk_plan_create_ex', 'yk_run_u8', 'yk_run_f32', 'yk_get_output', 'yk_debug_read_tensor',
                   'yk_plan_launch_count', 'yk_plan_launch_info', 'yk_plan_check', 'yk_plan_peek_error', 'yk_plan_debug_set_error', 'yk_plan_profile', 'yk_decode_py', 'yk_decode_py_ex', 'yk_decode_py_packed',
                   'yk_graph_begin', 'yk_graph_end', 'yk_graph_launch', 'yk_graph_node_count', 'yk_graph_kernel_node_count', 'yk_memcpy_async', 'yk_host_device_ptr', 'yk_stream_create', 'yk_stream_destroy', 'yk_stream_query_priority', 'yk_normalise_u8', 'yk_region_batched', 'yk_yolo_loss', 'yk_letterbox_u8',
                   'region_layer_init', 'yk_gemm_f32', 'yk_gemm_f32_grouped', 'yk_im2col3x3_f32', 'yk_col2im3x3_f32', 'yk_conv3x3_bn_fwd_f32', 'yk_conv3x3_bwd_weight_f32', 'yk_conv3x3_bwd_data_f32', 'yk_dw3x3_fwd_f32',
                   'yk_dw3x3_bwd_data_f32', 'yk_dw3x3_bwd_weight_f32', 'yk_dw3x3_bwd_weight_grouped_f32', 'yk_bn_train_fwd_f32', 'yk_bn_train_fwd_res_f32', 'yk_gemm_bn_fwd_f32', 'yk_dw3x3_bn_fwd_f32', 'yk_l2_segments_f32', 'yk_bn_train_bwd_f32',
                   'yk_bias_add_f32', 'yk_colsum_f32', 'yk_upsample2x_bwd_f32', 'yk_maxpool2_fwd_f32',
                   'yk_maxpool2_bwd_f32', 'yk_axpy_f32', 'yk_adam_f32', 'yk_dot_f32'):
            getattr(L, fn).restype = C.c_int
        L.yk_plan_destroy.restype = None
        L.yk_graph_destroy.restype = None
        L.yk_scratch_generation.restype = C.c_ulonglong
        _lib = L
    return _lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise YkError(f'{what} failed ({rc}): {lib().yk_last_error().decode()}')


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available() or lib().yk_device_count() <= 0:
        raise YkError('no MI355X / HIP device visible: the HIP path is the only path (no CPU fallback)')


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


def _stream(stream=None) -> C.c_void_p:
    import torch
    s = torch.cuda.current_stream() if stream is None else stream
    return C.c_void_p(s.cuda_stream)


class _DevView:
    """Zero-copy torch view of library-owned device memory via __cuda_array_interface__."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        import weakref
        self.__cuda_array_interface__ = {'shape': tuple(int(s) for s in shape), 'typestr': typestr,
                                         'data': (int(ptr), False), 'version': 2}
        # weak: torch keeps this object alive inside the tensor's storage deleter, where Python's GC cannot see it; a strong
        # reference would tie Plan -> views -> storage -> _DevView -> Plan into a cycle that never frees the device memory.
        # The views are BORROWED (kpu_get_output semantics): they are valid until the plan is closed.
        self._owner = weakref.ref(owner)


class Plan:
    """kpu_load_kmodel analogue (main.c:274): a compiled, device-resident network."""

    def __init__(self, spec: ns.NetSpec, weights, max_batch: int = 32, device: Optional[int] = None, precision: str = 'f16x2',
                 schedule: str = 'latency'):
        """precision: 'f16x2' (default; activations stored as fp16 pairs hi + lo, compensated fp16 MFMA operands: the mode that meets
        BASELINE.json's 1e-3 / exact-index tolerance) or 'f16' (fp16 activations, about twice as fast, 5e-3 worst case on the
        scores); see include/yolo_hip.h YK_PRECISION_*.
        schedule (f16x2): 'latency' - the late backbone and the heads as two launches of per-image workgroup clusters, the shortest time
        for ONE batch (a Plan on its own runs one batch at a time, hence the default here) - or 'throughput' - one launch per layer,
        which overlaps better when several plans run on several streams (Pipeline picks it for depth >= 2); YK_SCHEDULE_*."""
        import torch
        require_gpu()
        if precision not in PRECISIONS:
            raise YkError(f'precision {precision!r}: expected one of {sorted(PRECISIONS)}')
        if schedule not in SCHEDULES:
            raise YkError(f'schedule {schedule!r}: expected one of {sorted(SCHEDULES)}')
        self.precision = precision
        self.schedule = schedule
        self.spec = spec
        self.max_batch = int(max_batch)
        self.device = torch.cuda.current_device() if device is None else int(device)
        ops, tens, blob = spec.compile_plan(weights)
        self._ops = np.ascontiguousarray(ops, np.int32)
        self._tens = np.ascontiguousarray(tens, np.int32)
        blob = np.ascontiguousarray(blob, np.float32)
        outs = np.ascontiguousarray(spec.outputs, np.int32)
        self._h = C.c_void_p()
        L = lib()
        _check(L.yk_plan_create_ex(C.byref(self._h), self._ops.ctypes.data_as(i32p), C.c_int(len(ops)),
                                   self._tens.ctypes.data_as(i32p), C.c_int(len(tens)), blob.ctypes.data_as(f32p),
                                   C.c_size_t(blob.size), outs.ctypes.data_as(i32p), C.c_int(len(outs)),
                                   C.c_int(self.max_batch), C.c_int(self.device), C.c_int(PRECISIONS[precision] | SCHEDULES[schedule])), 'yk_plan_create_ex')
        self._out_views = None

    def close(self):
        self._out_views = None
        if getattr(self, '_h', None) and self._h.value:
            lib().yk_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- run ---------------------------------------------------------------
    def run_u8(self, frames, stream=None) -> None:
        """frames: torch.uint8 cuda tensor [B,H,W,3] (kpu_run_kmodel analogue, async)."""
        assert frames.is_cuda and frames.dtype.__str__() == 'torch.uint8' and frames.is_contiguous()
        assert tuple(frames.shape[1:]) == (*self.spec.in_hw, 3), frames.shape
        _check(lib().yk_run_u8(self._h, _ptr(frames), C.c_int(frames.shape[0]), _stream(stream)), 'yk_run_u8')

    def run_f32(self, x, stream=None) -> None:
        assert x.is_cuda and x.dtype.__str__() == 'torch.float32' and x.is_contiguous()
        assert tuple(x.shape[1:]) == (*self.spec.in_hw, 3), x.shape
        _check(lib().yk_run_f32(self._h, _ptr(x), C.c_int(x.shape[0]), _stream(stream)), 'yk_run_f32')

    def check(self) -> None:
        """Wait for the device; raise if an earlier asynchronous run of this plan failed on the device (yk_plan_check)."""
        _check(lib().yk_plan_check(self._h), 'yk_plan_check')

    def raise_if_failed(self) -> None:
        """Raise if a run of this plan that has ALREADY finished failed on the device (yk_plan_peek_error: a read of mapped host memory,
        no synchronisation - call it after waiting for the run's stream or event).  The flag is cleared by the report."""
        e = C.c_uint()
        _check(lib().yk_plan_peek_error(self._h, C.c_int(1), C.byref(e)), 'yk_plan_peek_error')
        if e.value:
            raise YkError('f16x2 cluster launch: a workgroup cluster did not assemble (its workgroups were not co-resident, e.g. another '
                          'kernel held CUs for the whole launch); the results of that run are invalid.  Use schedule=\'throughput\' when the '
                          'GPU is shared.')

    def output_ptrs(self) -> List[Tuple[int, Tuple[int, int, int]]]:
        res = []
        for i in range(len(self.spec.outputs)):
            p = f32p()
            nb = C.c_size_t()
            h, w, c = C.c_int(), C.c_int(), C.c_int()
            _check(lib().yk_get_output(self._h, C.c_int(i), C.byref(p), C.byref(nb), C.byref(h), C.byref(w), C.byref(c)),
                   'yk_get_output')
            res.append((C.cast(p, C.c_void_p).value, (h.value, w.value, c.value)))
        return res

    def outputs(self):
        """Borrowed torch views [max_batch,h,w,A*(5+C)] fp32 of the network outputs (kpu_get_output)."""
        import torch
        if self._out_views is None:
            self._out_views = [torch.as_tensor(_DevView(p, (self.max_batch, *s), '<f4', self), device=f'cuda:{self.device}')
                               for p, s in self.output_ptrs()]
        return self._out_views

    def read_tensor(self, tid: int, batch: int) -> np.ndarray:
        h, w, c = self.spec.tensors[tid]
        out = np.empty((batch, h, w, c), np.float32)
        _check(lib().yk_debug_read_tensor(self._h, C.c_int(tid), C.c_int(batch), out.ctypes.data_as(f32p),
                                          C.c_size_t(out.size)), 'yk_debug_read_tensor')
        return out

    def profile(self, frames, iters: int = 10, stream=None) -> np.ndarray:
        """Average per-launch duration (ms) measured with HIP events on the launch stream."""
        n = lib().yk_plan_launch_count(self._h)
        ms = np.zeros(n, np.float32)
        _check(lib().yk_plan_profile(self._h, _ptr(frames), C.c_int(frames.shape[0]), C.c_int(iters), _stream(stream),
                                     ms.ctypes.data_as(f32p)), 'yk_plan_profile')
        return ms

    def launches(self):
        n = lib().yk_plan_launch_count(self._h)
        res = []
        for i in range(n):
            name = C.create_string_buffer(128)
            fl, by = C.c_double(), C.c_double()
            lib().yk_plan_launch_info(self._h, C.c_int(i), name, C.c_size_t(128), C.byref(fl), C.byref(by))
            res.append((name.value.decode(), fl.value, by.value))
        return res


def make_decode_cfg(anchors: np.ndarray, class_num: int, in_hw, out_hw) -> DecodeCfg:
    anchors = np.asarray(anchors, np.float32)
    L, A = anchors.shape[0], anchors.shape[1]
    if L > YK_MAX_LAYERS or A > YK_MAX_ANCHORS:
        raise YkError('too many layers/anchors')
    cfg = DecodeCfg()
    cfg.n_layers, cfg.anchor_num, cfg.class_num = L, A, int(class_num)
    cfg.in_h, cfg.in_w = int(in_hw[0]), int(in_hw[1])
    for l in range(L):
        cfg.out_h[l], cfg.out_w[l] = int(out_hw[l][0]), int(out_hw[l][1])
        for n in range(A):
            cfg.anchors[l][n][0] = float(anchors[l, n, 0])
            cfg.anchors[l][n][1] = float(anchors[l, n, 1])
    return cfg


def decode_py(cfg: DecodeCfg, preds: Sequence, batch: int, image_hw=None, obj_thresh: float = 0.7,
              iou_thresh: float = 0.5, max_out: int = 30, stream=None, return_index: bool = False):
    """Batched keras_inference.py:94-135 on the GPU.  preds: cuda fp32 tensors [>=batch,h,w,A*(5+C)].
    -> (dets [batch, C*max_out, 6] cuda fp32, counts [batch] cuda int32[, box_index [batch, C*max_out] cuda int32: each row's index
    in the reference's flattened (layer, h, w, anchor) box list])."""
    import torch
    require_gpu()
    dev = preds[0].device
    dets = torch.empty((batch, cfg.class_num * max_out, 6), dtype=torch.float32, device=dev)
    counts = torch.empty((batch,), dtype=torch.int32, device=dev)
    arr = (C.c_void_p * len(preds))(*[C.c_void_p(p.data_ptr()) for p in preds])
    ihw = None
    if image_hw is not None:
        ihw = torch.as_tensor(np.broadcast_to(np.asarray(image_hw, np.float32), (batch, 2)).copy(), device=dev)
    if return_index:
        index = torch.full((batch, cfg.class_num * max_out), -1, dtype=torch.int32, device=dev)
        _check(lib().yk_decode_py_ex(C.byref(cfg), arr, C.c_int(batch), _ptr(ihw) if ihw is not None else None,
                                     C.c_float(obj_thresh), C.c_float(iou_thresh), C.c_int(max_out), _ptr(dets), _ptr(counts),
                                     _ptr(index), _stream(stream)), 'yk_decode_py_ex')
        return dets, counts, index
    _check(lib().yk_decode_py(C.byref(cfg), arr, C.c_int(batch), _ptr(ihw) if ihw is not None else None,
                              C.c_float(obj_thresh), C.c_float(iou_thresh), C.c_int(max_out), _ptr(dets), _ptr(counts),
                              _stream(stream)), 'yk_decode_py')
    return dets, counts


def region_batched(inp, W: int, H: int, A: int, Cn: int, anchor, threshold: float, nms_value: float,
                   net_wh=(320, 224), image_wh=(320, 224), layout: str = 'chw', want_output: bool = True, stream=None):
    """Batched C-mode region layer.  inp: cuda fp32 [B, A*(5+C), H, W] ('chw') or [B,H,W,A*(5+C)] ('hwc').
    -> (output|None, boxes [B,nb,4], probs [B,nb,C+1])."""
    import torch
    require_gpu()
    B = inp.shape[0]
    E, hw, nb = 5 + Cn, W * H, A * W * H
    cfg = RegionCfg()
    cfg.layer_w, cfg.layer_h, cfg.anchor_num, cfg.classes = W, H, A, Cn
    cfg.net_w, cfg.net_h, cfg.image_w, cfg.image_h = net_wh[0], net_wh[1], image_wh[0], image_wh[1]
    cfg.threshold, cfg.nms_value = threshold, nms_value
    for i, v in enumerate(np.asarray(anchor, np.float32).ravel()):
        cfg.anchor[i] = float(v)
    if layout == 'chw':
        cfg.stride_b, cfg.stride_n, cfg.stride_e, cfg.stride_y, cfg.stride_x = A * E * hw, E * hw, hw, W, 1
    else:
        cfg.stride_b, cfg.stride_n, cfg.stride_e, cfg.stride_y, cfg.stride_x = A * E * hw, E, 1, W * A * E, A * E
    out = torch.empty((B, A * E, H, W), dtype=torch.float32, device=inp.device) if want_output else None
    boxes = torch.empty((B, nb, 4), dtype=torch.float32, device=inp.device)
    probs = torch.empty((B, nb, Cn + 1), dtype=torch.float32, device=inp.device)
    _check(lib().yk_region_batched(C.byref(cfg), _ptr(inp), C.c_int(B), _ptr(out) if out is not None else None,
                                   _ptr(boxes), _ptr(probs), _stream(stream)), 'yk_region_batched')
    return out, boxes, probs


def yolo_loss(y_true, y_pred, anchors_l, obj_thresh, iou_thresh, obj_weight, noobj_weight, wh_weight, batch_size=None,
              counts=None, want_grad=True, want_ignore=False, stream=None):
    """tools/utils.py:741-791 loss_fn + custom.py metrics for one layer on the GPU.
    y_true / y_pred: cuda fp32 [B,h,w,A,5+C].  -> (loss[6] = total,xy,wh,obj,noobj,cls ; grad | None ; ignore | None).
    `counts` (cuda fp32 [3], running tp/fp/fn) is updated in place when given."""
    import torch
    require_gpu()
    assert y_true.is_cuda and y_pred.is_cuda and y_true.shape == y_pred.shape and y_pred.dtype == torch.float32
    y_true, y_pred = y_true.contiguous(), y_pred.contiguous()
    B, h, w, A, E = y_pred.shape
    cfg = LossCfg()
    cfg.out_h, cfg.out_w, cfg.anchor_num, cfg.class_num = h, w, A, E - 5
    for n, (aw, ah) in enumerate(np.asarray(anchors_l, np.float32)):
        cfg.anchors[n][0], cfg.anchors[n][1] = float(aw), float(ah)
    cfg.obj_thresh, cfg.iou_thresh = obj_thresh, iou_thresh
    cfg.obj_weight, cfg.noobj_weight, cfg.wh_weight = obj_weight, noobj_weight, wh_weight
    cfg.batch_size = int(batch_size if batch_size else B)
    loss = torch.empty(6, dtype=torch.float32, device=y_pred.device)
    grad = torch.empty_like(y_pred) if want_grad else None
    ign = torch.empty((B, h, w, A), dtype=torch.float32, device=y_pred.device) if want_ignore else None
    _check(lib().yk_yolo_loss(C.byref(cfg), _ptr(y_true), _ptr(y_pred), C.c_int(B), _ptr(loss),
                              _ptr(grad) if grad is not None else None, _ptr(ign) if ign is not None else None,
                              _ptr(counts) if counts is not None else None, _stream(stream)), 'yk_yolo_loss')
    return loss, grad, ign


def letterbox_u8(frames, dst_hw, stream=None, out=None):
    """GPU Helper._process_img letterbox (tools/utils.py:378-399): cuda uint8 [B,h,w,3] -> cuda uint8 [B,H,W,3]."""
    import torch
    require_gpu()
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.is_contiguous() and frames.shape[-1] == 3
    B, sh, sw, _ = frames.shape
    if out is None:
        out = torch.empty((B, int(dst_hw[0]), int(dst_hw[1]), 3), dtype=torch.uint8, device=frames.device)
    assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (B, int(dst_hw[0]), int(dst_hw[1]), 3)
    _check(lib().yk_letterbox_u8(_ptr(frames), C.c_int(B), C.c_int(sh), C.c_int(sw), _ptr(out), C.c_int(out.shape[1]),
                                 C.c_int(out.shape[2]), _stream(stream)), 'yk_letterbox_u8')
    return out


class Graph:
    """A captured step (yk_graph_*, include/yolo_hip.h): one host call replays every launch recorded between begin and end."""

    def __init__(self, handle: C.c_void_p):
        self._h = handle

    @property
    def nodes(self) -> int:
        return int(lib().yk_graph_node_count(self._h)) if self._h else 0

    @property
    def kernel_nodes(self) -> int:
        return int(lib().yk_graph_kernel_node_count(self._h)) if self._h else 0

    def launch(self, stream_handle: C.c_void_p) -> None:
        _check(lib().yk_graph_launch(self._h, stream_handle), 'yk_graph_launch')

    def close(self) -> None:
        if self._h is not None and self._h.value:
            lib().yk_graph_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def capture(stream_handle: C.c_void_p, issue) -> Graph:
    """Record what `issue()` submits to the stream as a Graph.  The stream must have run the same step eagerly before."""
    L = lib()
    _check(L.yk_graph_begin(stream_handle), 'yk_graph_begin')
    try:
        issue()
    except Exception:
        h = C.c_void_p()
        L.yk_graph_end(stream_handle, C.byref(h))                  # leave capture mode whatever happened
        if h.value:
            L.yk_graph_destroy(h)
        raise
    h = C.c_void_p()
    _check(L.yk_graph_end(stream_handle, C.byref(h)), 'yk_graph_end')
    return Graph(h)


def _pinned(shape, dtype):
    """Pinned host tensor + the address a kernel may use for it."""
    import torch
    t = torch.empty(shape, dtype=dtype).pin_memory()
    d = C.c_void_p()
    _check(lib().yk_host_device_ptr(C.c_void_p(t.data_ptr()), C.byref(d)), 'yk_host_device_ptr')
    return t, d


class Ticket:
    """One batch submitted from host memory (Pipeline.submit_host).  `result()` waits for it and returns (rows [n,6] float32 =
    top,left,bottom,right,score,class ; offsets [B+1] int32: image b owns rows[offsets[b]:offsets[b+1]]) - the concatenation of
    keras_inference.py:133-135, for every image of the batch.  Valid until the slot is reused `depth` submits later."""

    def __init__(self, slot, batch, event, with_index):
        self._slot, self.batch, self.event, self._with_index = slot, batch, event, with_index

    def result(self):
        self.event.synchronize()
        s = self._slot
        s.plan.raise_if_failed()
        off = s.h_offsets[:self.batch + 1].numpy().copy()
        n = int(off[-1])
        rows = s.h_rows[:n].numpy().copy()
        if self._with_index:
            return rows, off, s.h_index[:n].numpy().copy()
        return rows, off


class _Slot:
    pass


class Pipeline:
    """Several independent batches in flight on one GPU (the kpu_run_kmodel(async)+callback shape of main.c:303-311, widened).

    One batch alone leaves CUs idle: a step is ~27 dependent launches, most of them a single round of <= 256 workgroups.
    `depth` plans, each with its own outputs / split-K slabs, on `depth` HIP streams (decode scratch is per stream inside the
    library) let the GPU interleave workgroups of different batches; nothing is shared between them but the weight values.

    graph=True (default): a slot's step - (letterbox) -> yk_run_u8 -> decode + per-class NMS -> results - is captured once as a
    hipGraph and REPLAYED: one host call per batch instead of ~30 launches (the host -> device copy of submit_host runs in front of it on the
    pipeline's copy stream, into one of the slot's two device input buffers, under the slot's previous batch).  A captured step is bound to the
    buffers it was captured with: device frames are replayed in place when they live at an address the slot has already captured
    (the slot's own `input(i)` buffer, or up to two caller buffers - a resident ring), and are copied into `input(i)` otherwise.

    submit() returns at once; the returned (dets, counts) tensors belong to that slot and are overwritten when the slot is reused
    `depth` submits later, so consume them (or call `wait`) before that.  submit_host() takes frames from (pinned) host memory and
    delivers the detections to host memory at their live size; see Ticket."""

    def __init__(self, spec: ns.NetSpec, weights, anchors, max_batch: int = 32, depth: int = 3, device: Optional[int] = None,
                 precision: str = 'f16x2', graph: bool = True, src_hw: Optional[Tuple[int, int]] = None, max_out: int = 30,
                 schedule: str = 'auto', streams: str = 'native'):
        """schedule: 'auto' = 'latency' for depth 1 (one batch at a time: the cluster launches finish it soonest), 'throughput' for
        depth >= 2 (batches in flight on several streams: the launch-per-layer form overlaps better); see Plan.
        streams: 'native' = the slots' HIP streams are created by the library, back to back (yk_stream_create): they land on
        consecutive hardware queues whatever streams the process created before; 'torch' = streams from torch's pool."""
        import torch
        require_gpu()
        if streams not in ('native', 'torch'):
            raise YkError(f'streams {streams!r}: expected native or torch')
        self.depth = max(1, int(depth))
        self.schedule = ('latency' if self.depth == 1 else 'throughput') if schedule == 'auto' else schedule
        self.spec, self.max_batch, self.graph = spec, int(max_batch), bool(graph)
        self.src_hw = None if src_hw is None else (int(src_hw[0]), int(src_hw[1]))     # frames of this size are letterboxed on the GPU
        self.max_out = int(max_out)
        self.plans = [Plan(spec, weights, max_batch=max_batch, device=device, precision=precision, schedule=self.schedule) for _ in range(self.depth)]
        self.outs = [p.outputs() for p in self.plans]
        dev = torch.device(f'cuda:{self.plans[0].device}')
        cur = torch.cuda.current_stream()
        self._own_streams = []
        if streams == 'native':
            with torch.cuda.device(dev):
                for _ in range(self.depth):
                    h = C.c_void_p()
                    _check(lib().yk_stream_create(C.byref(h), C.c_int(0)), 'yk_stream_create')
                    self._own_streams.append(h)
            self.streams = [torch.cuda.ExternalStream(h.value, device=dev) for h in self._own_streams]
        else:
            self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)]
        for s in self.streams:
            s.wait_stream(cur)
        self.cfg = make_decode_cfg(anchors, spec.class_num, spec.in_hw, spec.out_hw())
        B, H, W = self.max_batch, *spec.in_hw
        nrow = spec.class_num * self.max_out
        self.slots = []
        for i in range(self.depth):
            s = _Slot()
            s.plan, s.stream, s.st = self.plans[i], self.streams[i], C.c_void_p(self.streams[i].cuda_stream)
            s.preds = (C.c_void_p * len(self.outs[i]))(*[C.c_void_p(o.data_ptr()) for o in self.outs[i]])
            s.frames = torch.full((B, H, W, 3), 127, dtype=torch.uint8, device=dev)      # (defined bytes: the warm-up step reads them)
            s.src = torch.full((B, *self.src_hw, 3), 127, dtype=torch.uint8, device=dev) if self.src_hw else s.frames
            s.image_hw = torch.empty((B, 2), dtype=torch.float32, device=dev)
            s.dets = torch.zeros((B, nrow, 6), dtype=torch.float32, device=dev)
            s.counts = torch.zeros((B,), dtype=torch.int32, device=dev)
            s.index = torch.full((B, nrow), -1, dtype=torch.int32, device=dev)
            s.h_src = s.h_rows = s.h_offsets = s.h_index = None                        # pinned side, built on first submit_host
            s.graphs = {}
            s.scratch_gen, s.warm_dev, s.warm_host = 0, False, False
            s.foreign = []                                                              # caller buffers this slot has captured
            self.slots.append(s)
        self._n = 0
        self.host_us = 0.0
        self.host_wait_us = 0.0                                                         # submit_host: time the submit thread spent BLOCKED on a slot's input buffer (cumulative)
        self._copy_stream = None                                                        # submit_host: the H2D leg's own stream (created on first use)

    # -- buffers ------------------------------------------------------------------------------------
    def input(self, i: int):
        """Slot i's own device input [max_batch, h, w, 3] uint8 (camera size when src_hw is set): fill it, then submit(None)."""
        return self.slots[i % self.depth].src

    def host_input(self, i: int):
        """Slot i's pinned host input (same shape as input(i)): fill it, then submit_host(None)."""
        s = self.slots[i % self.depth]
        self._host_side(s)
        return s.h_src

    def next_slot(self) -> int:
        return self._n % self.depth

    def _host_side(self, s):
        import torch
        if s.h_src is None:
            nrow = self.max_batch * self.spec.class_num * self.max_out
            s.h_src, _ = _pinned(tuple(s.src.shape), torch.uint8)
            s.h_rows, s.d_rows = _pinned((nrow, 6), torch.float32)
            s.h_offsets, s.d_offsets = _pinned((self.max_batch + 1,), torch.int32)
            s.h_index, s.d_index = _pinned((nrow,), torch.int32)
            s.h_offsets.zero_()
            # the frames of a host batch land in one of TWO device buffers, copied on the pipeline's copy stream: the copy of this slot's
            # next batch runs under its current batch's kernels instead of in front of them on the same stream (round 5: the copy leg is
            # 125 us of a 330 us step, and a stream that copies is a stream that does not compute - from host 0.87 of resident before)
            s.h2d_bufs = [s.src, torch.full_like(s.src, 127)]
            s.h2d_next = 0
            s.h2d_copied = [torch.cuda.Event(), torch.cuda.Event()]
            s.staged = None                                         # (buffer index, batch) of a copy stage_host() has started
            s.h2d_free = [None, None]
            if self._copy_stream is None:
                if self._own_streams:                                   # library-created, like the slots' streams
                    h = C.c_void_p()
                    with torch.cuda.device(s.src.device):
                        _check(lib().yk_stream_create(C.byref(h), C.c_int(0)), 'yk_stream_create')
                    self._own_streams.append(h)
                    self._copy_stream = torch.cuda.ExternalStream(h.value, device=s.src.device)
                else:
                    self._copy_stream = torch.cuda.Stream(device=s.src.device)

    # -- the step, as the library calls it is made of (eager, or recorded by capture()) ---------------------------------
    def _h2d_start(self, s, B):
        """Issue the host -> device copy of slot `s`'s pinned frames into its other device input buffer, on the pipeline's COPY stream (the copy of
        a slot's next batch runs under its current batch's kernels).  -> the buffer's index; `s.h2d_copied[b]` fires when the frames are there."""
        import time
        b = s.h2d_next
        s.h2d_next ^= 1
        dst, cs = s.h2d_bufs[b], self._copy_stream
        if s.h2d_free[b] is not None and not s.h2d_free[b].query():
            # the batch that last read this buffer (two submits of this slot ago) - waited for on the HOST, where it is over long ago in steady
            # state: a wait inside the copy stream would put barrier packets into a fifth hardware queue, and a fifth active queue costs the
            # four compute streams a quarter of their rate (measured 85 -> 61 k images/s)
            t0 = time.perf_counter()
            s.h2d_free[b].synchronize()
            self.host_wait_us += (time.perf_counter() - t0) * 1e6      # blocked, not busy: bench.py reports the two apart
        _check(lib().yk_memcpy_async(C.c_void_p(dst.data_ptr()), C.c_void_p(s.h_src.data_ptr()), C.c_size_t(B * s.src[0].numel()),
                                     C.c_void_p(cs.cuda_stream)), 'yk_memcpy_async')
        s.h2d_copied[b].record(cs)
        return b

    def stage_host(self, i: Optional[int] = None, batch: Optional[int] = None) -> None:
        """The frames in host_input(i) are final (default: the slot of the next submit): start their copy to the device NOW, so that it runs
        while earlier batches compute and `submit_host(None)` of that slot finds it done.  A producer calls this when it has filled a slot;
        bench.py calls it one submit ahead.  Optional: without it submit_host issues the copy itself."""
        s = self.slots[(self._n if i is None else int(i)) % self.depth]
        self._host_side(s)
        B = self.max_batch if batch is None else int(batch)
        if getattr(s, 'staged', None) is None:
            s.staged = (self._h2d_start(s, B), B)

    def _h2d(self, s, B):
        """The host -> device leg, issued EAGERLY in front of the replay (measured, images/s from host with four batches in flight: copy
        engine in front of the replay 66 k; as a copy node inside the captured step 55 k; as a kernel inside it that reads the pinned
        frames through their device alias 54 k - profiles/r04_schedules.txt) on the pipeline's COPY stream into the slot's other input buffer.
        The step is launched when the HOST has seen the copy finish (round 6: tools/from_host_split.py, one box, resident 99.7 k: the slot's
        stream waiting for the copy's event 91.4 k - a cross-queue barrier packet per step; the host waiting 92.5 k; the host waiting for a copy
        that stage_host() started one submit earlier 94.8 k = what the copy traffic alone costs).  -> the device address the step reads."""
        import time
        st = getattr(s, 'staged', None)
        s.staged = None
        b = st[0] if st is not None and st[1] == B else self._h2d_start(s, B)
        if not s.h2d_copied[b].query():
            t0 = time.perf_counter()
            s.h2d_copied[b].synchronize()
            self.host_wait_us += (time.perf_counter() - t0) * 1e6
        s.h2d_last = b
        return s.h2d_bufs[b].data_ptr()

    def _h2d_done(self, s):
        """After the step has been given to the slot's stream: its input buffer is free once the stream gets here."""
        import torch
        ev = torch.cuda.Event()
        ev.record(s.stream)
        s.h2d_free[s.h2d_last] = ev

    def _issue(self, s, B, src_ptr, host, use_hw, obj, iou, max_out, want_index):
        L = lib()
        H, W = self.spec.in_hw
        x = src_ptr
        if self.src_hw:
            _check(L.yk_letterbox_u8(C.c_void_p(x), C.c_int(B), C.c_int(self.src_hw[0]), C.c_int(self.src_hw[1]), _ptr(s.frames),
                                     C.c_int(H), C.c_int(W), s.st), 'yk_letterbox_u8')
            x = s.frames.data_ptr()
        _check(L.yk_run_u8(s.plan._h, C.c_void_p(x), C.c_int(B), s.st), 'yk_run_u8')
        ihw = _ptr(s.image_hw) if use_hw else None
        if host:
            _check(L.yk_decode_py_packed(C.byref(self.cfg), s.preds, C.c_int(B), ihw, C.c_float(obj), C.c_float(iou), C.c_int(max_out),
                                         s.d_rows, s.d_offsets, s.d_index if want_index else None, None, None, s.st), 'yk_decode_py_packed')
        else:
            _check(L.yk_decode_py_ex(C.byref(self.cfg), s.preds, C.c_int(B), ihw, C.c_float(obj), C.c_float(iou), C.c_int(max_out),
                                     _ptr(s.dets), _ptr(s.counts), _ptr(s.index) if want_index else None, s.st), 'yk_decode_py_ex')

    GRAPH_CACHE = 8                     # captured steps kept per slot (one per distinct (batch, source, thresholds, ...) combination)

    def _drop_graphs(self, s):
        for g in s.graphs.values():
            g.close()
        s.graphs.clear()

    def _run(self, s, B, src_ptr, host, use_hw, obj, iou, max_out, want_index):
        if max_out > self.max_out:
            raise YkError(f'max_out {max_out} > the pipeline\'s max_out {self.max_out}')
        if host:
            src_ptr = self._h2d(s, B)
        args = (B, src_ptr, host, use_hw, float(obj), float(iou), int(max_out), bool(want_index))
        if not self.graph:
            self._issue(s, *args)
            if host:
                self._h2d_done(s)
            return
        # a captured step holds the address of the stream's decode scratch: if that buffer has moved since (it cannot once the slot is
        # warm - __init__ sizes it for max_batch x max_out - but another user of the same stream could grow it), every capture is stale
        gen = int(lib().yk_scratch_generation(s.st))
        if gen != s.scratch_gen:
            s.stream.synchronize()
            self._drop_graphs(s)
            s.scratch_gen = gen
        g = s.graphs.pop(args, None)
        if g is None:
            self._warm(s, host)
            if len(s.graphs) >= self.GRAPH_CACHE:                  # least recently used first (dicts keep insertion order)
                s.stream.synchronize()
                s.graphs.pop(next(iter(s.graphs))).close()
            g = capture(s.st, lambda: self._issue(s, *args))
        s.graphs[args] = g                                         # (re-)inserted last = most recently used
        g.launch(s.st)
        if host:
            self._h2d_done(s)

    def _warm(self, s, host):
        """One eager step at the slot's LARGEST shape (max_batch images, max_out rows per class, box indices on) for each result form the
        slot uses: sizes the library's per-stream scratch once, so that no later batch size or threshold can make it grow - and move -
        under a captured step."""
        B, did = self.max_batch, False
        if not s.warm_dev:
            self._issue(s, B, s.src.data_ptr(), False, False, 0.7, 0.5, self.max_out, True)
            s.warm_dev = did = True
        if host and not s.warm_host:
            self._issue(s, B, s.src.data_ptr(), True, False, 0.7, 0.5, self.max_out, True)
            s.warm_host = did = True
        if did:
            s.stream.synchronize()
            gen = int(lib().yk_scratch_generation(s.st))
            if gen != s.scratch_gen:                               # the warm-up itself moved a buffer older captures point into
                self._drop_graphs(s)
                s.scratch_gen = gen

    def _set_hw(self, s, B, image_hw):
        import torch
        if image_hw is None:
            return False
        hw = torch.as_tensor(np.broadcast_to(np.asarray(image_hw, np.float32), (B, 2)).copy())
        with torch.cuda.stream(s.stream):
            s.image_hw[:B].copy_(hw, non_blocking=False)
        return True

    def submit(self, frames_u8=None, image_hw=None, obj_thresh: float = 0.7, iou_thresh: float = 0.5, max_out: int = 30,
               return_index: bool = False, batch: Optional[int] = None, sync_input: bool = True):
        """frames_u8: cuda uint8 [B,h,w,3] that stays alive until the results are consumed (None: the slot's own input(i) buffer, `batch`
        images of it).  -> (dets [B, C*max_out, 6], counts [B], stream[, box_index [B, C*max_out]]) - the slot's tensors, sliced to B."""
        import time
        import torch
        t0 = time.perf_counter()
        s = self.slots[self._n % self.depth]
        self._n += 1
        if frames_u8 is None:
            B = self.max_batch if batch is None else int(batch)
            ptr = s.src.data_ptr()
        else:
            assert frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
            assert tuple(frames_u8.shape[1:]) == tuple(s.src.shape[1:]), (frames_u8.shape, s.src.shape)
            B = int(frames_u8.shape[0])
            if sync_input:
                s.stream.wait_stream(torch.cuda.current_stream())       # the frames may have been produced on the caller's stream
            ptr = frames_u8.data_ptr()
            if self.graph and ptr != s.src.data_ptr() and ptr not in s.foreign:
                if len(s.foreign) < 2:
                    s.foreign.append(ptr)                               # a resident caller buffer: replayed in place from now on
                else:
                    with torch.cuda.stream(s.stream):
                        s.src[:B].copy_(frames_u8, non_blocking=True)
                    ptr = s.src.data_ptr()
        assert 0 < B <= self.max_batch
        use_hw = self._set_hw(s, B, image_hw)
        self._run(s, B, ptr, False, use_hw, obj_thresh, iou_thresh, max_out, return_index)
        self.host_us = (time.perf_counter() - t0) * 1e6
        if return_index:
            return s.dets[:B], s.counts[:B], s.stream, s.index[:B]
        return s.dets[:B], s.counts[:B], s.stream

    def submit_host(self, frames_u8=None, image_hw=None, obj_thresh: float = 0.7, iou_thresh: float = 0.5, max_out: int = 30,
                    return_index: bool = False, batch: Optional[int] = None) -> Ticket:
        """frames_u8: host uint8 [B,h,w,3] (numpy or torch; copied into the slot's pinned buffer) or None = `batch` images already
        written to host_input(i).  The frames cross PCIe, the detections come back to pinned host memory at their live size."""
        import time
        import torch
        t0 = time.perf_counter()
        s = self.slots[self._n % self.depth]
        self._n += 1
        self._host_side(s)
        if frames_u8 is None:
            B = self.max_batch if batch is None else int(batch)
        else:
            f = torch.as_tensor(frames_u8)
            assert f.dtype == torch.uint8 and tuple(f.shape[1:]) == tuple(s.src.shape[1:]), f.shape
            B = int(f.shape[0])
            for ev in s.h2d_copied:                                     # the slot's previous copies may still be reading h_src
                ev.synchronize()
            s.staged = None                                             # (a copy staged from the buffer's previous contents is not this batch)
            s.h_src[:B].copy_(f)
        assert 0 < B <= self.max_batch
        use_hw = self._set_hw(s, B, image_hw)
        self._run(s, B, None, True, use_hw, obj_thresh, iou_thresh, max_out, return_index)
        ev = torch.cuda.Event()
        ev.record(s.stream)
        self.host_us = (time.perf_counter() - t0) * 1e6
        return Ticket(s, B, ev, return_index)

    def wait(self):
        for s in self.streams:
            s.synchronize()
        for p in self.plans:
            p.raise_if_failed()

    def close(self):
        if self._copy_stream is not None:
            self._copy_stream.synchronize()
            self._copy_stream = None
        for s in self.streams:
            s.synchronize()
        for s in self.slots:
            self._drop_graphs(s)
        for p in self.plans:
            p.close()
        self.plans = []
        self.slots = []
        self.streams = []
        for h in self._own_streams:
            lib().yk_stream_destroy(h)
        self._own_streams = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
