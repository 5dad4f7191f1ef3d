"""The other BASELINE.json configurations on one GPU (they are parity-test cases, not bench lines): images/s of the conv stack + python-mode
decode / NMS for every network of the reference, both precision modes, one and three batches in flight (engine.Pipeline).

    python tools/netbench.py            # prints a markdown table
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

CASES = [('yolo_mobilev1', 0.75, 224, 320, 32, 'configs[1] (the bench line)'),
         ('tiny_yolo', 1.0, 416, 416, 8, 'configs[2]: 64 images over 8 GPUs'),
         ('yolo_mobilev2', 1.0, 224, 320, 16, 'configs[3] network, inference'),
         ('yolo', 1.0, 416, 416, 8, 'configs[4]: Darknet-53'),
         ('yolo', 1.0, 416, 416, 32, 'configs[4]: Darknet-53, 32 images (where the 3x3 layers are MFMA-bound)'),
         ('yolo', 1.0, 416, 416, 64, 'configs[4]: Darknet-53, 64 images')]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
print('| network | input | batch | mode | images/s, 1 in flight | images/s, 4 in flight | ms per batch (1 in flight) | note |')
print('|---|---|---|---|---|---|---|---|')
for name, alpha, H, W, B, note in CASES:
    spec = ns.NETWORKS[name]((H, W, 3), 3, 20, alpha=alpha)
    anchors = VOC_ANCHORS if len(spec.outputs) == 2 else np.concatenate([VOC_ANCHORS, VOC_ANCHORS[:1] * 0.5])
    w = spec.init_weights(seed=1)
    frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device='cuda')
    for prec in ('f16x2', 'f16'):
        rates = {}
        for depth in (1, 4):
            pipe = engine.Pipeline(spec, w, anchors, max_batch=B, depth=depth, precision=prec)
            for _ in range(3 * depth):
                pipe.submit(frames)
            pipe.wait()
            t0 = time.perf_counter()
            for _ in range(30):
                pipe.submit(frames)
            pipe.wait()
            n = max(60, int(0.6 / ((time.perf_counter() - t0) / 30)))     # ~0.6 s of back-to-back submits, no drain in between: a drained
            t0 = time.perf_counter()                                     # pipeline restarts with its streams in phase (the same kernels of
            for _ in range(n):                                           # three batches competing for the same resource: -20 %)
                pipe.submit(frames)
            pipe.wait()
            dt = time.perf_counter() - t0
            rates[depth] = (B * n / dt, dt / n * 1e3)
            pipe.close()
        print(f'| {name}-{alpha:g} | {H}x{W} | {B} | {prec} | {rates[1][0]:,.0f} | {rates[4][0]:,.0f} | {rates[1][1]:.3f} | {note} |', flush=True)
