"""Developer build: where a wave's k-loop time goes in igemm_pipe_kernel (cycles of wave 0 of every workgroup, summed over the k-steps).
    YK_LIB_PATH=.../libyolo_hip_dev.so python tools/r05_igemm_phase.py H W C1 C2 B   (env: YK_IGEMM_FORCE, YK_NS, YK_PIPE_IL, YK_SPLIT_FORCE, KS)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

H, W, C1, C2, B = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (52, 52, 128, 256, 32)))
s = ns.NetSpec('probe', (H, W), anchor_num=3, class_num=20)
x = s._new_tensor(H, W, 3)
x = s.conv(x, 32, 3, act=ns.LEAKY01, name='conv2d_1')
x = s.conv(x, C1, 3, act=ns.LEAKY01, name='conv2d_2')
y = s.conv(x, C2, int(os.environ.get('KS', '3')), act=ns.LEAKY01, name='conv2d_3')
z = s.conv(y, 75, 1, bn=False, bias=True, name='conv2d_4', net_output=True)
s.outputs = [z]
plan = engine.Plan(s, s.init_weights(1), max_batch=B, precision='f16')
f = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device='cuda')
for _ in range(3):
    plan.run_u8(f)
torch.cuda.synchronize()
ms = plan.profile(f, iters=5)
li = [i for i, (n, _, _) in enumerate(plan.launches()) if f'_{C1}to{C2}[' in n][0]
name, fl, _ = plan.launches()[li]
nwg = 16384
out = np.zeros((nwg, 8), np.int64)
L = engine.lib()
L.yk_debug_phase_stamps.restype = C.c_int
rc = L.yk_debug_phase_stamps(plan._h, C.c_int(li), C.c_void_p(f.data_ptr()), C.c_int(B), C.c_void_p(torch.cuda.current_stream().cuda_stream),
                             out.ctypes.data_as(C.c_void_p), C.c_int(nwg))
assert rc == 0, rc
v = out[out[:, 0] > 0]
nk = v[:, 7].astype(float)
tag = f"cfg={os.environ.get('YK_IGEMM_FORCE', 'A')} ns={os.environ.get('YK_NS', '-')} il={os.environ.get('YK_PIPE_IL', '1')}"
print(f'{name} {tag}: {ms[li] * 1e3:.1f} us ({fl * B / ms[li] / 1e9:.0f} TF/s; with stamps on the kernel is slower), {len(v)} workgroups, {nk.mean():.0f} k-steps')
span = (v[:, 6].max() - v[:, 0].min()) / 100.0
print(f'   stamped span {span:.1f} us; per workgroup: prologue+loop {np.median(v[:, 5] - v[:, 0]) / 100.0:.2f} us, epilogue {np.median(v[:, 6] - v[:, 5]) / 100.0:.2f} us')
for k, lab in ((1, 'wait vmcnt'), (2, 'barrier'), (3, 'dma issue'), (4, 'frags+mfma')):
    per = v[:, k] / nk
    print(f'   {lab:11s} median {np.median(per):7.0f} cycles per k-step   p90 {np.percentile(per, 90):7.0f}')
tot = (v[:, 1] + v[:, 2] + v[:, 3] + v[:, 4]) / nk
print(f'   sum         median {np.median(tot):7.0f} cycles per k-step')
