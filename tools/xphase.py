"""Dev tool: per-workgroup phase timestamps of one fused block launch of the f16x2 plan (wall_clock64 ticks, 100 MHz -> 10 ns).

    python tools/xphase.py <launch index>[,<launch index>...] [B]        (XPH_SCHEDULE=throughput: the plan bench.py's `value` runs)
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
plan = engine.Plan(spec, spec.init_weights(seed=1), max_batch=B, precision='f16x2', schedule=os.environ.get('XPH_SCHEDULE', 'latency'))
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    plan.run_u8(frames)
torch.cuda.synchronize()
L = engine.lib()
L.yk_debug_phase_stamps.restype = C.c_int
lab = ['start', 'patch0 issued', 'prep done', 'patch0 landed (bar1)', 'dw0 done', 'bar2', 'loop end', 'drained (bar)', 'pass1 staged', 'copied out', 'end']
for li in [int(v) for v in sys.argv[1].split(',')]:
    name = plan.launches()[li][0]
    nwg = 16384
    out = np.zeros((nwg, 16), np.int64)
    rc = L.yk_debug_phase_stamps(plan._h, C.c_int(li), C.c_void_p(frames.data_ptr()), C.c_int(B),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream), out.ctypes.data_as(C.c_void_p), C.c_int(nwg))
    assert rc == 0, rc
    v = out[out[:, 0] > 0]
    t0 = v[:, 0].min()
    print(name, 'workgroups', len(v), 'kernel span %.2f us' % ((v[:, 10].max() - t0) / 100.0))
    d = (v - v[:, :1]) / 100.0
    for k in range(1, 11):
        print('  %-22s median %6.2f us  p90 %6.2f  max %6.2f   (step %+5.2f)' % (lab[k], np.median(d[:, k]), np.percentile(d[:, k], 90), d[:, k].max(),
                                                                               np.median(d[:, k] - d[:, k - 1])))
    if v[:, 11].max() > 0:
        print('  stem: window loads issued+stored %.2f us, barrier %.2f us, patch done (stamp 1) %.2f us' % (np.median(d[:, 11]), np.median(d[:, 12]), np.median(d[:, 1])))
    if v[:, 13].max() > 0:                                        # steady-state step (the second one)
        print('  step 1: top (bar1) %.2f us | dw %.2f | bar2 %.2f   (step 0: bar1 %.2f | dw %.2f | bar2 %.2f; step 0 bar2 -> step 1 bar1 = mma + wait %.2f)' % (
            np.median(d[:, 13]), np.median(d[:, 14] - d[:, 13]), np.median(d[:, 15] - d[:, 14]), np.median(d[:, 3]), np.median(d[:, 4] - d[:, 3]),
            np.median(d[:, 5] - d[:, 4]), np.median(d[:, 13] - d[:, 5])))
    st = (v[:, 0] - t0) / 100.0
    print('  WG start times: median %.2f p90 %.2f max %.2f us' % (np.median(st), np.percentile(st, 90), st.max()))
plan.close()
