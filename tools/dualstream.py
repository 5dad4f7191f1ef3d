"""Experiment: B=32 as S independent sub-batches on S streams (co-resident kernels hide each other's latency)."""
import sys, time
sys.path.insert(0, '.')
import torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
for S in (1, 2, 4):
    B = 32 // S
    plans = [engine.Plan(spec, w, max_batch=B, precision='f16') for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    frames = [torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda') for _ in range(S)]
    def step():
        for p, st, f in zip(plans, streams, frames):
            with torch.cuda.stream(st):
                p.run_u8(f, stream=st)
                engine.decode_py(cfg, p.outputs(), B, None, 0.7, 0.5, stream=st)
    for _ in range(10): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'S={S} sub-batches of {B}: {dt*1e6:.1f} us per 32 images -> {32/dt:.0f} img/s')
    for p in plans: p.close()
