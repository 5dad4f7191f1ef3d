"""Dev tool: per-workgroup phase timestamps of one fused launch (wall_clock64 ticks, 100 MHz -> 10 ns)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

li = int(sys.argv[1]) if len(sys.argv) > 1 else 9
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
plan = engine.Plan(spec, spec.init_weights(seed=1), max_batch=B, precision='f16')
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    plan.run_u8(frames)
torch.cuda.synchronize()
name = plan.launches()[li][0]
nwg = 8192
out = np.zeros((nwg, 8), np.int64)
L = engine.lib()
L.yk_debug_phase_stamps.restype = C.c_int
rc = L.yk_debug_phase_stamps(plan._h, C.c_int(li), C.c_void_p(frames.data_ptr()), C.c_int(B),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream), out.ctypes.data_as(C.c_void_p), C.c_int(nwg))
assert rc == 0
v = out[out[:, 0] > 0]
t0 = v[:, 0].min()
print(name, 'workgroups', len(v), 'kernel span %.2f us' % ((v[:, 6].max() - t0) / 100.0))
d = (v - v[:, :1]) / 100.0
lab = ['start', 'Ws+bar', 'dw done', 'barrier', 'gemm+C', 'barrier', 'stored']
for k in range(1, 7):
    print('  %-12s median %6.2f us  p90 %6.2f  max %6.2f' % (lab[k], np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max()))
st = (v[:, 0] - t0) / 100.0
print('  WG start times: median %.2f p90 %.2f max %.2f us' % (np.median(st), np.percentile(st, 90), st.max()))
# per-CU residency reconstruction
hw = v[:, 7]
xcc = (hw >> 32) & 0xf
hwid = hw & 0xffffffff
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 10 + cu
ids = np.unique(key)
print('  distinct (xcc,se,sh,cu):', len(ids))
conc, gaps, per_cu = [], [], []
for k in ids:
    w = v[key == k]
    w = w[np.argsort(w[:, 0])]
    per_cu.append(len(w))
    ev = sorted([(t, 1) for t in w[:, 0]] + [(t, -1) for t in w[:, 6]])
    c = m = 0
    area = 0.0
    last = ev[0][0]
    for t, d in ev:
        area += c * (t - last)
        last = t
        c += d
        m = max(m, c)
    conc.append((m, area / max(1, (ev[-1][0] - ev[0][0]))))
    g = w[1:, 0] - w[:-1, 6]
    gaps.extend(g.tolist())
conc = np.array(conc)
print('  per-CU: WGs %.1f, max concurrent WGs median %d, time-avg concurrent %.2f' % (np.mean(per_cu), np.median(conc[:, 0]), conc[:, 1].mean()))
gaps = np.array(gaps) / 100.0
if len(gaps):
    print('  gap between a WG end (stamp6) and the next WG start on the same CU: median %.2f us, p10 %.2f, p90 %.2f' % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90)))
