"""images/s of engine.Pipeline (B=32, f16x2) at depth 1 and 4 under the current environment (developer switches via YK_LIB_PATH=..._dev.so)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
frames = torch.randint(0, 256, (32, 224, 320, 3), dtype=torch.uint8, device='cuda')
sched = sys.argv[1] if len(sys.argv) > 1 else 'throughput'
out = []
for depth in (4, 1):
    pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=32, depth=depth, precision='f16x2', schedule=sched)
    for _ in range(12):
        pipe.submit(frames, sync_input=False)
    pipe.wait()
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(150):
            pipe.submit(frames, sync_input=False)
        pipe.wait()
        best = max(best, 32 * 150 / (time.perf_counter() - t0))
    out.append(f'depth {depth}: {best:,.0f}')
    if depth == 4:
        ms = pipe.plans[0].profile(frames, iters=5)
        out.append(f'sum {ms.sum() * 1e3:.0f} us / {len(ms)} launches')
    pipe.close()
print(' | '.join(out), flush=True)
