"""Two-conv plan (stem 3->C1, then conv3x3 C1->C2 at HxW) for counter runs on ONE igemm launch (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k210_yolo_framework_amd import engine, netspec as ns

H, W, C1, C2, B = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (26, 26, 256, 512, 16)))
s = ns.NetSpec('probe', (H, W), anchor_num=3, class_num=20)
x = s._new_tensor(H, W, 3)
x = s.conv(x, 32, 3, act=ns.LEAKY01, name='conv2d_1')
x = s.conv(x, C1, 3, act=ns.LEAKY01, name='conv2d_2')
y = s.conv(x, C2, int(os.environ.get('KS', '3')), act=ns.LEAKY01, name='conv2d_3')
z = s.conv(y, 75, 1, bn=False, bias=True, name='conv2d_4', net_output=True)
s.outputs = [z]
w = s.init_weights(1)
plan = engine.Plan(s, w, max_batch=B, precision='f16')
f = torch.rand(B, H, W, 3, device='cuda')
for _ in range(int(os.environ.get('ITERS', '10'))):
    plan.run_f32(f)
torch.cuda.synchronize()
ms = plan.profile(f, iters=10) if os.environ.get('PROFILE') else None
if ms is not None:
    for (n, fl, by), t in zip(plan.launches(), ms):
        print(f'{n:50s} {t*1e3:8.2f} us {fl*B/t/1e9:8.1f} TF/s')
