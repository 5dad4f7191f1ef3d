"""Dev tool: per-launch times of the Darknet-53 plan (BASELINE configs[4]: 416x416, 8 images), f16 or f16x2.

    python tools/darknet_layers.py [f16|f16x2] [B]
"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

prec = sys.argv[1] if len(sys.argv) > 1 else 'f16'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spec = ns.yolo((416, 416, 3), 3, 20)
w = spec.init_weights(seed=1)
plan = engine.Plan(spec, w, max_batch=B, precision=prec)
x = torch.randint(0, 256, (B, 416, 416, 3), dtype=torch.uint8, device="cuda")
ms = plan.profile(x, iters=10) if hasattr(plan, 'profile') else None
rows = []
tot = 0.0
for (nm, fl, by), m in zip(plan.launches(), ms):
    tot += m
    rows.append((m, nm, fl * B / m / 1e9, by * B / m / 1e6))
print(f'{prec} B={B}: {len(rows)} launches, sum {tot:.3f} ms -> {B / tot * 1e3:.0f} img/s; total {sum(r[2] * r[0] for r in rows) / tot:.1f} TF/s average')
acc = 0.0
for m, nm, tf, gb in sorted(rows, reverse=True)[:28]:
    acc += m
    print(f'{m * 1e3:8.1f} us  {tf:7.1f} TF/s  {gb:8.1f} GB/s  cum {acc / tot * 100:5.1f} %  {nm[:90]}')
