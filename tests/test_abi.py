"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/yolo_hip.h declares; struct layouts match region_layer.h:7-39; no compute calls (no GPU here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / 'include' / 'yolo_hip.h'
LIB = ROOT / 'k210_yolo_framework_amd' / 'csrc' / 'libyolo_hip.so'


def declared_functions():
    txt = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)
    txt = re.sub(r'typedef\s+void\s*\(\*\w+\)\s*\([^;]*?\);', '', txt, flags=re.S)   # callback typedef is not an export
    names = re.findall(r'^\s*(?:const\s+)?(?:int|void|char|size_t)\s*\*?\s*(\w+)\s*\(', txt, flags=re.M)
    return sorted(set(names))


@pytest.fixture(scope='module')
def dll():
    if not LIB.exists():
        import __graft_entry__ as g
        g.build()
    import torch  # noqa: F401  (first, as engine.lib() does: the library must bind the HIP runtime torch ships, or this process loses its device)
    return C.CDLL(str(LIB))


def test_header_declares_the_reference_abi():
    fns = declared_functions()
    for f in ('region_layer_init', 'region_layer_deinit', 'region_layer_run', 'region_layer_draw_boxes',   # region_layer.h:44-48
              'yk_plan_create', 'yk_run_u8', 'yk_run_f32', 'yk_get_output', 'yk_decode_py', 'yk_region_batched'):
        assert f in fns


def test_library_exports_every_declared_symbol(dll):
    for f in declared_functions():
        assert hasattr(dll, f), f'libyolo_hip.so does not export {f}'


def test_struct_layouts_match_reference_header():
    import oracle
    rl = oracle.RegionLayerT
    # region_layer.h:19-39 on LP64: 2 floats, 2 u32, ptr @16, 9 u32 (+4 pad), 5 pointers @64..96
    assert C.sizeof(rl) == 104
    assert rl.anchor.offset == 16 and rl.image_width.offset == 24 and rl.boxes.offset == 64
    assert rl.input.offset == 72 and rl.probs.offset == 96
    from k210_yolo_framework_amd import engine
    assert C.sizeof(engine.DecodeCfg) == 5 * 4 + 2 * 4 * 4 + 4 * 8 * 2 * 4
    assert C.sizeof(engine.RegionCfg) == 8 * 4 + 2 * 4 + 16 * 4 + 5 * 8


def test_no_gpu_means_loud_failure(dll):
    """Without a HIP device the product must refuse, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    dll.yk_device_count.restype = C.c_int
    assert dll.yk_device_count() == 0
    from k210_yolo_framework_amd import engine, netspec
    with pytest.raises(engine.YkError):
        engine.require_gpu()
    s = netspec.yolo_mobilev1((32, 32, 3), 3, 2, alpha=0.5)
    with pytest.raises(engine.YkError):
        engine.Plan(s, s.init_weights(), max_batch=1)
    rl = __import__('oracle').RegionLayerT()
    anc = (C.c_float * 6)(*[0.5] * 6)
    rl.anchor_number, rl.anchor, rl.threshold, rl.nms_value = 3, anc, 0.6, 0.3
    dll.region_layer_init.restype = C.c_int
    assert dll.region_layer_init(C.byref(rl), 10, 7, 75, 320, 224) == -5


def test_boundary_default_is_the_conforming_mode():
    """A caller that follows INTEGRATION.md section 2 (yk_plan_create / engine.Plan / the plugin classes with no precision argument)
    gets the mode whose results meet BASELINE.json's tolerance (f16x2); the faster fp16-storage mode is opt-in."""
    import inspect
    from k210_yolo_framework_amd import engine, yolonet
    assert inspect.signature(engine.Plan.__init__).parameters['precision'].default == 'f16x2'
    assert inspect.signature(engine.Pipeline.__init__).parameters['precision'].default == 'f16x2'
    assert yolonet.YoloModel(None, {}, False).precision == 'f16x2'
    src = (ROOT / 'k210_yolo_framework_amd' / 'csrc' / 'yk_engine.hip').read_text()
    body = src[src.index('extern "C" int yk_plan_create('):src.index('extern "C" int yk_plan_create_ex(')]
    assert 'YK_PRECISION_F16X2' in body and 'YK_PRECISION_F16)' not in body


@pytest.mark.gpu
def test_yk_plan_create_through_ctypes_gives_fp32_class_results(dll):
    """The C entry point itself (not the Python default): outputs within 1e-4 of max|logit| of the FP32 oracle."""
    import numpy as np
    import torch
    import oracle
    from k210_yolo_framework_amd import engine, netspec
    spec = netspec.yolo_mobilev1((64, 96, 3), 3, 20, alpha=0.75)
    w = spec.init_weights(seed=2)
    ops, tens, blob = spec.compile_plan(w)
    ops, tens = np.ascontiguousarray(ops, np.int32), np.ascontiguousarray(tens, np.int32)
    blob, outs = np.ascontiguousarray(blob, np.float32), np.ascontiguousarray(spec.outputs, np.int32)
    h = C.c_void_p()
    vp = C.c_void_p
    dll.yk_last_error.restype = C.c_char_p
    rc = dll.yk_plan_create(C.byref(h), ops.ctypes.data_as(vp), C.c_int(len(ops)), tens.ctypes.data_as(vp), C.c_int(len(tens)),
                            blob.ctypes.data_as(vp), C.c_size_t(blob.size), outs.ctypes.data_as(vp), C.c_int(len(outs)), C.c_int(2), C.c_int(0))
    assert rc == 0, dll.yk_last_error()
    frames = np.random.default_rng(1).integers(0, 256, (2, 64, 96, 3), dtype=np.uint8)
    fd = torch.from_numpy(frames).cuda()
    assert dll.yk_run_u8(h, vp(fd.data_ptr()), C.c_int(2), vp(torch.cuda.current_stream().cuda_stream)) == 0
    torch.cuda.synchronize()
    ref = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames), emulate_f16=False, out_ids=spec.outputs)
    for i, r in enumerate(ref):
        ptr, nb, hh, ww, cc = vp(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        assert dll.yk_get_output(h, C.c_int(i), C.byref(ptr), C.byref(nb), C.byref(hh), C.byref(ww), C.byref(cc)) == 0
        got = np.empty((2, hh.value, ww.value, cc.value), np.float32)
        got[...] = torch.as_tensor(engine._DevView(ptr.value, got.shape, '<f4', dll), device='cuda:0').cpu().numpy()   # borrowed view
        assert np.abs(got - r.reshape(got.shape)).max() <= 1e-4 * np.abs(r).max()
    dll.yk_plan_destroy.restype = None
    dll.yk_plan_destroy(h)


def test_package_import_asks_for_one_hardware_queue_per_pipeline_stream():
    """engine.Pipeline keeps three batches in flight on three HIP streams; the ROCm default of 4 hardware queues per process (shared with
    the null stream and torch's pooled streams) can put two of them on one queue.  The package asks for 8 unless the user chose."""
    import subprocess
    import sys
    code = "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import k210_yolo_framework_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=str(ROOT))
    assert out.stdout.strip() == '8', out.stderr
    code = "import os; os.environ['GPU_MAX_HW_QUEUES'] = '2'; import k210_yolo_framework_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=str(ROOT))
    assert out.stdout.strip() == '2', out.stderr


def test_schedule_and_precision_constants_match_the_header():
    """engine.py passes `precision | schedule` to yk_plan_create_ex: the numbers must be the header's."""
    from k210_yolo_framework_amd import engine
    txt = HEADER.read_text()
    val = lambda name: int(re.search(rf'#define\s+{name}\s+(0x[0-9a-fA-F]+|\d+)', txt).group(1), 0)
    assert engine.PRECISIONS == {'f16': val('YK_PRECISION_F16'), 'f16x2': val('YK_PRECISION_F16X2')}
    assert engine.SCHEDULES == {'throughput': val('YK_SCHEDULE_THROUGHPUT'), 'latency': val('YK_SCHEDULE_LATENCY')}
    assert all((v & val('YK_SCHEDULE_MASK')) == v for v in engine.SCHEDULES.values())
    assert all((v & val('YK_SCHEDULE_MASK')) == 0 for v in engine.PRECISIONS.values())
