"""Sweep igemm tile config x split-K for the long-K head convs (dev tool).  YK_IGEMM_FORCE / YK_SPLIT_FORCE are read at plan build."""
import os
import sys
sys.path.insert(0, '.')
import torch
from k210_yolo_framework_amd import engine, netspec as ns

spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
B = 32
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
res = {}
for cfg in ['', '4', '0', '8', '9', '10', '5', '3']:
    for split in ['', '2', '4', '6', '8', '12', '16', '24']:
        os.environ['YK_IGEMM_FORCE'], os.environ['YK_SPLIT_FORCE'] = cfg, split
        try:
            plan = engine.Plan(spec, w, max_batch=B, precision='f16')
        except Exception as e:
            print(cfg, split, 'ERR', e)
            continue
        for _ in range(3):
            plan.run_u8(frames)
        torch.cuda.synchronize()
        ms = plan.profile(frames, iters=20)
        L = plan.launches()
        # long-K convs and the reduce launches that follow them
        out = []
        for i, (n, fl, by) in enumerate(L):
            if 'igemm' in n and fl > 4e8:
                t = ms[i] * 1e3 + (ms[i + 1] * 1e3 if i + 1 < len(L) and 'reduce' in L[i + 1][0] else 0.0)
                out.append((n, round(float(t), 1)))
        print(f'cfg={cfg or "auto":>4s} split={split or "auto":>4s} sum={ms.sum()*1e3:7.1f}us  ' + '  '.join(f'{n}:{t}' for n, t in out), flush=True)
        plan.close()
