"""Throughput of any of the four networks with `depth` batches in flight (engine.Pipeline), conv stack + python-mode decode/NMS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

name, alpha, H, W, B = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
spec = ns.NETWORKS[name]((H, W, 3), 3, 20, alpha=alpha)
anchors = VOC_ANCHORS if len(spec.outputs) == 2 else np.concatenate([VOC_ANCHORS, VOC_ANCHORS[:1] * 0.5])
w = spec.init_weights(seed=1)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device='cuda')
for depth in (1, 2, 3, 4):
    pipe = engine.Pipeline(spec, w, anchors, max_batch=B, depth=depth)
    for _ in range(3 * depth):
        pipe.submit(frames)
    pipe.wait()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit(frames)
    pipe.wait()
    dt = time.perf_counter() - t0
    print(f'{name} {H}x{W} B={B} depth={depth}: {B * n / dt:9.0f} images/s  {dt / n * 1e3:7.3f} ms/batch', flush=True)
    pipe.close()
