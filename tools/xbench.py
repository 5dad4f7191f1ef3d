"""Dev tool: the f16x2 plan of the headline network — error vs the fp32 oracle on 4 frames, per-launch times at B=32.

    python tools/xbench.py [net] [B]          (environment: YK_FUSE_DWPW, YK_X_FUSE_MAXC, YK_X_BN, YK_X_PATCH, YK_X_SPLITK)
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from k210_yolo_framework_amd import engine, netspec as ns

name = sys.argv[1] if len(sys.argv) > 1 else 'yolo_mobilev1'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
shape = (224, 320, 3)
alpha = 0.75 if name == 'yolo_mobilev1' else 1.0
spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
w = spec.init_weights(seed=1)
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (B, *shape), dtype=np.uint8)
nb = 4
ref = oracle.net_forward(spec.compile_plan(w), oracle.normalise_u8(frames[:nb]), False, spec.outputs)
plan = engine.Plan(spec, w, max_batch=B, precision='f16x2')
t = torch.from_numpy(frames).cuda()
plan.run_u8(t)
torch.cuda.synchronize()
outs = [o[:nb].cpu().numpy() for o in plan.outputs()]
for o, r in zip(outs, ref):
    print(f'max|ref| {np.abs(r).max():9.3g}  err max {np.abs(o - r).max():9.3g}  rel {np.abs(o - r).max() / np.abs(r).max():8.2g}')
ms = plan.profile(t, iters=20)
tot = 0.0
for (nm, fl, by), m in zip(plan.launches(), ms):
    tot += m
    print(f'{nm:60s} {m * 1e3:8.1f} us  {fl * B / m / 1e9:8.1f} TF/s  {by * B / m / 1e6:8.1f} GB/s')
print(f'sum {tot * 1e3:.1f} us -> {B / tot * 1e3:.0f} img/s one batch, back to back')
plan.close()
