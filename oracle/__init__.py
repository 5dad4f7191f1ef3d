"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (k210_yolo_framework_amd + libyolo_hip.so) never does and fails
loudly when the HIP library or a GPU is missing.

Contents
  liboracle.so          region_layer_ref.c + yolo_net_ref.c (this repo's restatements)
  _ref/libregion_ref.so the reference's own region_layer.c, built by build_ref.sh where
                        /root/reference exists (travels prebuilt to the GPU box)
  decode_ref.py         numpy restatement of keras_inference.py:94-135 (python-mode decode)
  helper_ref.py         (none) — Helper's numpy members are host code of the product itself
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

HERE = Path(__file__).resolve().parent
_LIB = None
_REF = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


def build(force: bool = False) -> None:
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    if force or not (HERE / 'liboracle.so').exists():
        subprocess.check_call(['make', '-C', str(HERE), 'liboracle.so'], stdout=subprocess.DEVNULL)
    if force or not (HERE / '_ref' / 'libregion_ref.so').exists():
        subprocess.check_call(['sh', str(HERE / 'build_ref.sh')], stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        build()
        _LIB = C.CDLL(str(HERE / 'liboracle.so'))
        _LIB.rlref_iou.restype = C.c_float
        _LIB.rlref_draw.restype = C.c_int
        _LIB.yk_ref_forward.restype = C.c_int
        _LIB.yk_ref_forward_ex.restype = C.c_int
        _LIB.yk_ref_f16_round.restype = C.c_float
        _LIB.yk_ref_f16_round.argtypes = [C.c_float]
        _LIB.yk_ref_set_threads.restype = C.c_int
        _LIB.yk_ref_set_threads(int(os.environ.get('ORACLE_THREADS', min(32, os.cpu_count() or 1))))
    return _LIB


def set_threads(n: int) -> int:
    """OpenMP team size of the C oracle's conv loops (default min(32, cpus): a team of every logical CPU spin-waits at each loop's
    barrier and slows down 10x on a host shared with other jobs).  Returns the value in effect."""
    return int(lib().yk_ref_set_threads(int(n)))


def have_ref() -> bool:
    return (HERE / '_ref' / 'libregion_ref.so').exists()


# --------------------------------------------------------------------------- #
# C-mode region layer (restatement)
# --------------------------------------------------------------------------- #
def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def region_run(inp: np.ndarray, anchor: Sequence[float], W: int, H: int, A: int, C_: int, threshold: float,
               nms_value: float, net_wh=(320, 224), image_wh=(320, 224)):
    """-> (output, boxes[nb,4], probs[nb,C+1]) of rlref_run on a CHW fp32 tensor."""
    L = lib()
    inp = np.ascontiguousarray(inp, np.float32).ravel()
    nb = A * W * H
    assert inp.size == nb * (5 + C_)
    out = np.empty_like(inp)
    boxes = np.empty((nb, 4), np.float32)
    probs = np.empty((nb, C_ + 1), np.float32)
    anc = np.ascontiguousarray(anchor, np.float32)
    L.rlref_run(_p(inp, f32p), _p(out, f32p), _p(anc, f32p), C.c_int(W), C.c_int(H), C.c_int(A), C.c_int(C_),
                C.c_float(threshold), C.c_float(nms_value), C.c_uint32(net_wh[0]), C.c_uint32(net_wh[1]),
                C.c_uint32(image_wh[0]), C.c_uint32(image_wh[1]), _p(boxes, f32p), _p(probs, f32p))
    return out, boxes, probs


def region_draw(boxes: np.ndarray, probs: np.ndarray, threshold: float, image_wh=(320, 224)) -> np.ndarray:
    """-> uint32 [n,6] rows (x1,y1,x2,y2,class,prob bits) in callback order."""
    L = lib()
    nb, c1 = probs.shape
    dets = np.zeros((nb, 6), np.uint32)
    n = L.rlref_draw(_p(np.ascontiguousarray(boxes, np.float32), f32p),
                     _p(np.ascontiguousarray(probs, np.float32), f32p), C.c_int(nb), C.c_int(c1 - 1),
                     C.c_float(threshold), C.c_uint32(image_wh[0]), C.c_uint32(image_wh[1]), _p(dets, u32p),
                     C.c_int(nb))
    return dets[:n].copy()


# --------------------------------------------------------------------------- #
# the reference's own region_layer.c (oracle/_ref), driven through its real ABI
# --------------------------------------------------------------------------- #
class RegionLayerT(C.Structure):
    """region_layer_t, region_layer.h:19-39."""
    _fields_ = [('threshold', C.c_float), ('nms_value', C.c_float), ('coords', C.c_uint32),
                ('anchor_number', C.c_uint32), ('anchor', f32p), ('image_width', C.c_uint32),
                ('image_height', C.c_uint32), ('classes', C.c_uint32), ('net_width', C.c_uint32),
                ('net_height', C.c_uint32), ('layer_width', C.c_uint32), ('layer_height', C.c_uint32),
                ('boxes_number', C.c_uint32), ('output_number', C.c_uint32), ('boxes', C.c_void_p),
                ('input', f32p), ('output', f32p), ('probs_buf', f32p), ('probs', C.POINTER(f32p))]


DRAW_CB = C.CFUNCTYPE(None, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float)


def drive_region_abi(dll: C.CDLL, inp: np.ndarray, anchor: Sequence[float], W: int, H: int, A: int, C_: int,
                     threshold: float, nms_value: float, net_wh=(320, 224)):
    """Run init/run/draw/deinit of a region_layer.h-compatible library exactly as main.c:278-324 does.
    Works for oracle/_ref AND for libyolo_hip.so (same ABI).  -> (output, boxes, probs, dets)."""
    dll.region_layer_init.restype = C.c_int
    rl = RegionLayerT()
    anc = np.ascontiguousarray(anchor, np.float32)
    rl.anchor_number = A
    rl.anchor = _p(anc, f32p)
    rl.threshold = threshold
    rl.nms_value = nms_value
    rc = dll.region_layer_init(C.byref(rl), C.c_int(W), C.c_int(H), C.c_int(A * (5 + C_)), C.c_int(net_wh[0]),
                               C.c_int(net_wh[1]))
    if rc != 0:
        raise RuntimeError(f'region_layer_init -> {rc}')
    x = np.ascontiguousarray(inp, np.float32).ravel()
    rl.input = _p(x, f32p)
    dll.region_layer_run(C.byref(rl), None)
    nb = int(rl.boxes_number)
    out = np.ctypeslib.as_array(rl.output, (int(rl.output_number),)).copy()
    boxes = np.ctypeslib.as_array(C.cast(rl.boxes, f32p), (nb, 4)).copy()
    probs = np.ctypeslib.as_array(rl.probs_buf, (nb, int(rl.classes) + 1)).copy()
    dets: List[Tuple] = []

    def cb(x1, y1, x2, y2, cls, prob):
        dets.append((x1, y1, x2, y2, cls, np.float32(prob).view(np.uint32)))
    cbf = DRAW_CB(cb)
    dll.region_layer_draw_boxes(C.byref(rl), cbf)
    dll.region_layer_deinit(C.byref(rl))
    d = np.array(dets, dtype=np.uint32).reshape(-1, 6)
    return out, boxes, probs, d


def ref_region_run(inp, anchor, W, H, A, C_, threshold, nms_value, net_wh=(320, 224)):
    """The reference's compiled region_layer.c (requires oracle/_ref)."""
    global _REF
    if _REF is None:
        _REF = C.CDLL(str(HERE / '_ref' / 'libregion_ref.so'))
    return drive_region_abi(_REF, inp, anchor, W, H, A, C_, threshold, nms_value, net_wh)


# --------------------------------------------------------------------------- #
# conv net interpreter
# --------------------------------------------------------------------------- #
def net_forward(plan, x: np.ndarray, emulate_f16: bool = False, out_ids: Optional[Sequence[int]] = None,
                dump_id: int = -1):
    """plan = (ops, tensors, blob) from NetSpec.compile_plan; x fp32 [B,H,W,3] normalised.
    -> list of fp32 NHWC outputs (and the dumped tensor if dump_id >= 0)."""
    L = lib()
    ops, tens, blob = plan
    ops = np.ascontiguousarray(ops, np.int32)
    tens = np.ascontiguousarray(tens, np.int32)
    blob = np.ascontiguousarray(blob, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    B = x.shape[0]
    assert tuple(x.shape[1:]) == tuple(tens[0, :3]), (x.shape, tens[0])
    ids = np.ascontiguousarray(out_ids, np.int32)
    outs = [np.empty((B, *tens[i, :3]), np.float32) for i in ids]
    ptrs = (f32p * len(outs))(*[_p(o, f32p) for o in outs])
    dump = np.empty((B, *tens[dump_id, :3]), np.float32) if dump_id >= 0 else np.empty(1, np.float32)
    rc = L.yk_ref_forward(_p(ops, i32p), C.c_int(len(ops)), _p(tens, i32p), C.c_int(len(tens)), _p(blob, f32p),
                          C.c_size_t(blob.size), _p(x, f32p), C.c_int(B), C.c_int(1 if emulate_f16 else 0),
                          _p(ids, i32p), C.c_int(len(outs)), ptrs, C.c_int(dump_id), _p(dump, f32p))
    if rc != 0:
        raise RuntimeError(f'yk_ref_forward -> {rc}')
    return (outs, dump) if dump_id >= 0 else outs


def net_forward_ex(plan, inputs: dict, op_rows: Sequence[int], out_ids: Sequence[int], emulate_f16: bool = True):
    """Run only the ops `op_rows` of the plan with the tensors in `inputs` ({tensor id: fp32 NHWC}) pre-filled.
    Used for layer-at-a-time parity (the GPU's own layer inputs go in, its layer output is compared)."""
    L = lib()
    ops, tens, blob = plan
    sub = np.ascontiguousarray(np.asarray(ops, np.int32)[list(op_rows)], np.int32)
    tens = np.ascontiguousarray(tens, np.int32)
    blob = np.ascontiguousarray(blob, np.float32)
    ids = np.ascontiguousarray(list(inputs.keys()), np.int32)
    arrs = [np.ascontiguousarray(inputs[int(i)], np.float32) for i in ids]
    B = arrs[0].shape[0]
    in_ptrs = (f32p * len(arrs))(*[_p(a, f32p) for a in arrs])
    oid = np.ascontiguousarray(out_ids, np.int32)
    outs = [np.empty((B, *tens[i, :3]), np.float32) for i in oid]
    out_ptrs = (f32p * len(outs))(*[_p(o, f32p) for o in outs])
    dummy = np.empty(1, np.float32)
    rc = L.yk_ref_forward_ex(_p(sub, i32p), C.c_int(len(sub)), _p(tens, i32p), C.c_int(len(tens)), _p(blob, f32p),
                             C.c_size_t(blob.size), C.c_int(len(arrs)), _p(ids, i32p), in_ptrs, C.c_int(B),
                             C.c_int(1 if emulate_f16 else 0), _p(oid, i32p), C.c_int(len(outs)), out_ptrs, C.c_int(-1),
                             _p(dummy, f32p))
    if rc != 0:
        raise RuntimeError(f'yk_ref_forward_ex -> {rc}')
    return outs


def normalise_u8(frames: np.ndarray) -> np.ndarray:
    """tools/utils.py:405 `img / np.max(img)` per image, -> fp32."""
    L = lib()
    frames = np.ascontiguousarray(frames, np.uint8)
    out = np.empty(frames.shape, np.float32)
    per = int(np.prod(frames.shape[1:]))
    L.yk_ref_normalise_u8(_p(frames, u8p), C.c_int(frames.shape[0]), C.c_size_t(per), _p(out, f32p))
    return out


def f16_round(a: np.ndarray) -> np.ndarray:
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
