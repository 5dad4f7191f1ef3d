"""Dev tool (developer build): per-phase timeline of the persistent late-backbone launch of the f16x2 plan (wall_clock64, 10 ns ticks).

    YK_LIB_PATH=.../libyolo_hip_dev.so python tools/xpersist_phase.py [B]
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
li = next(i for i, n in enumerate(names) if 'x:persist' in n)
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
nph = int(names[li].split(',')[1].split()[0])
t0 = v[:, 0].min()
print('workgroups', len(v), ' kernel span %.2f us' % ((v[:, 4 * nph].max() - t0) / 100.0), ' start skew max %.2f us' % ((v[:, 0].max() - t0) / 100.0))
for pi in range(nph):
    start, nxt = v[:, 4 * pi], v[:, 4 * (pi + 1)]
    s1, s2, s3 = v[:, 4 * pi + 1], v[:, 4 * pi + 2], v[:, 4 * pi + 3]
    med = lambda x: np.median(x) / 100
    if s2.max() > 0:            # pointwise: prologue | K loop | drain | epilogue
        print('phase %2d  pw   total %6.2f us   prologue %5.2f   loop %6.2f   drain %5.2f   epilogue %5.2f' % (pi, med(nxt - start), med(s1 - start), med(s2 - s1), med(s3 - s2), med(nxt - s3)))
    elif s1.max() > 0:          # depthwise: work until the arrival, then the barrier
        print('phase %2d  dw   work %6.2f us (p90 %6.2f)   barrier %6.2f us (p90 %6.2f, min %5.2f)' % (
            pi, med(s1 - start), np.percentile(s1 - start, 90) / 100, med(nxt - s1), np.percentile(nxt - s1, 90) / 100, (nxt - s1).min() / 100))
    else:
        print('phase %2d       work %6.2f us (p90 %6.2f)' % (pi, med(nxt - start), np.percentile(nxt - start, 90) / 100))
plan.close()
