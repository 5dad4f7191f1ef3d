"""Round-5 diagnostics (run on the GPU box): (a) is the step's low roofline fraction "B=32 under-fills 256 CUs" or kernel inefficiency -
the same throughput plan at max_batch 32 / 64 / 128; (b) stream order / hardware-queue effect: library-created streams vs torch's pool,
depth 3..8; (c) from-host rate on pipelines created first / later.

    python tools/r05_diag1.py [a|b|c ...]
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

what = sys.argv[1:] or ['a', 'b', 'c']
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)


def rate(pipe, frames, host=False, secs=0.5):
    step = (lambda: pipe.submit_host(None)) if host else (lambda: pipe.submit(frames, sync_input=False))
    for _ in range(3 * pipe.depth):
        step()
    pipe.wait()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    pipe.wait()
    n = max(40, int(secs / ((time.perf_counter() - t0) / 20)))
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        pipe.wait()
        best = max(best, frames.shape[0] * n / (time.perf_counter() - t0))
    return best


if 'a' in what:
    print('# (a) throughput plan, one stream: per-launch sum and rate by max_batch', flush=True)
    for B in (32, 64, 128):
        frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
        plan = engine.Plan(spec, w, max_batch=B, precision='f16x2', schedule='throughput')
        ms = plan.profile(frames, iters=10)
        names = [n for n, _, _ in plan.launches()]
        fam = {}
        for n, m in zip(names, ms):
            k = 'fused' if '+conv1x1' in n and 'dw3x3' in n.split('+conv')[0] else ('dw' if n.startswith('x:dw') else ('conv3x3' if 'conv3x3' in n else ('conv1x1' if 'conv1x1' in n else 'other')))
            fam[k] = fam.get(k, 0.0) + m * 1e3
        print(f'B={B}: sum {ms.sum() * 1e3:.1f} us = {ms.sum() * 1e3 / B:.2f} us/image  ' + '  '.join(f'{k} {v:.1f}' for k, v in sorted(fam.items())), flush=True)
        if B == 128:
            for n, m in zip(names, ms):
                print(f'    {n:62s} {m * 1e3:8.1f} us')
        plan.close()
        for depth in ((1, 2, 4) if B == 32 else (1, 2)):
            pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=B, depth=depth, precision='f16x2', schedule='throughput')
            print(f'    B={B} depth={depth}: {rate(pipe, frames):,.0f} images/s', flush=True)
            pipe.close()

if 'b' in what:
    print('# (b) streams: library-created (native) vs torch pool; B=32 throughput schedule', flush=True)
    frames = torch.randint(0, 256, (32, 224, 320, 3), dtype=torch.uint8, device='cuda')
    print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
    for kind in ('native', 'torch', 'native'):
        for depth in (3, 4, 5, 6, 8):
            pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=32, depth=depth, precision='f16x2', streams=kind)
            print(f'    {kind:6s} depth={depth}: {rate(pipe, frames):,.0f} images/s', flush=True)
            pipe.close()

if 'c' in what:
    print('# (c) from host, pipelines created one after another in this process (depth 4)', flush=True)
    frames = torch.randint(0, 256, (32, 224, 320, 3), dtype=torch.uint8, device='cuda')
    for kind in ('native', 'native', 'torch', 'torch', 'native'):
        pipe = engine.Pipeline(spec, w, VOC_ANCHORS, max_batch=32, depth=4, precision='f16x2', streams=kind)
        r0 = rate(pipe, frames)
        h = frames.cpu()
        for i in range(4):
            pipe.host_input(i).copy_(h)
        r1 = rate(pipe, frames, host=True)
        r2 = rate(pipe, frames)
        print(f'    {kind:6s}: resident {r0:,.0f}  from host {r1:,.0f} ({r1 / r0:.3f})  resident again {r2:,.0f}', flush=True)
        pipe.close()
