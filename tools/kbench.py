"""Per-launch HIP-event timing of a plan at several batch sizes (dev tool; not part of the product path)."""
import os
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

name = os.environ.get('NET', 'yolo_mobilev1')
alpha = float(os.environ.get('ALPHA', '0.75'))
shape = tuple(int(v) for v in os.environ.get('SHAPE', '224,320').split(','))
batches = [int(b) for b in (sys.argv[1:] or ['32'])]
spec = ns.NETWORKS[name]((*shape, 3), 3, 20, alpha=alpha)
w = spec.init_weights(seed=1)
cols = {}
for B in batches:
    plan = engine.Plan(spec, w, max_batch=B, precision='f16')
    frames = torch.randint(0, 256, (B, *shape, 3), dtype=torch.uint8, device='cuda')
    for _ in range(5):
        plan.run_u8(frames)
    torch.cuda.synchronize()
    ms = plan.profile(frames, iters=20)
    L = plan.launches()
    cols[B] = (ms, L)
    # wall clock of back-to-back runs
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(50):
        plan.run_u8(frames)
    t1.record(); torch.cuda.synchronize()
    print(f'B={B}: sum of kernels {ms.sum()*1e3:.1f} us, wall/run {t0.elapsed_time(t1)/50*1e3:.1f} us, {B/(t0.elapsed_time(t1)/50/1e3):.0f} img/s')
    plan.close()
L = cols[batches[0]][1]
print('%-52s' % 'kernel' + ''.join(f'{"B=%d us" % b:>12s}{"GB/s":>8s}{"TF/s":>7s}' for b in batches))
for i, (n, fl, by) in enumerate(L):
    row = '%-52s' % n
    for b in batches:
        t = cols[b][0][i] * 1e-3
        row += f'{t*1e6:12.2f}{by*b/t/1e9:8.0f}{fl*b/t/1e12:7.1f}'
    print(row)
