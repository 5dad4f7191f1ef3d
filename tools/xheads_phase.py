"""Dev tool (developer build): per-phase timeline of the heads launch of the f16x2 plan (wall_clock64, 10 ns ticks).

    YK_LIB_PATH=.../libyolo_hip_dev.so python tools/xheads_phase.py [B]
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
plan = engine.Plan(spec, spec.init_weights(seed=1), max_batch=B, precision='f16x2')
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    plan.run_u8(frames)
torch.cuda.synchronize()
names = [l[0] for l in plan.launches()]
li = next(i for i, n in enumerate(names) if n.startswith('x:heads'))
print(names[li])
L = engine.lib()
L.yk_debug_phase_stamps.restype = C.c_int
NS = 4 * 24 + 4
nwg = 256
raw = np.zeros((4096, 16), np.int64)
rc = L.yk_debug_phase_stamps(plan._h, C.c_int(li), C.c_void_p(frames.data_ptr()), C.c_int(B), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                             raw.ctypes.data_as(C.c_void_p), C.c_int(4096))
assert rc == 0, rc
v = raw.reshape(-1)[:nwg * NS].reshape(nwg, NS)
v = v[v[:, 0] > 0]
nph = names[li].count('|') + 1
t0 = v[:, 0].min()
print('workgroups', len(v), ' kernel span %.2f us' % ((v[:, 8 * nph].max() - t0) / 100.0), ' start skew max %.2f us' % ((v[:, 0].max() - t0) / 100.0))
med = lambda x: np.median(x) / 100
for pi in range(nph):
    st = [v[:, 8 * pi + k] for k in range(7)] + [v[:, 8 * (pi + 1)]]
    d = [med(st[k + 1] - st[k]) for k in range(7)]
    print('phase %d   total %6.2f us   fill %5.2f   K loop %6.2f (p90 %6.2f)   partial stores %5.2f   barrier %5.2f (min %5.2f)   reduce %5.2f   sync+max %5.2f   tail %5.2f' % (
        pi, med(st[7] - st[0]), d[0], d[1], np.percentile(st[2] - st[1], 90) / 100, d[2], d[3], (st[4] - st[3]).min() / 100, d[4], d[5], d[6]))
plan.close()
