"""Round 5: tile / ring sweep of the f16 implicit-GEMM over Darknet-53 conv shapes at B=32, one process (developer build: YK_IGEMM_FORCE,
YK_NS and YK_SPLIT_FORCE are read at plan creation / per launch).
    YK_LIB_PATH=.../libyolo_hip_dev.so python tools/r05_igemm_sweep.py [B]
cfg ids (yk_conv.h): A = the shipped pick, 9 = 64x128k64, 10 = 64x192, 11 = 256x128 on 8 waves, 12 = 256x128 on 4 waves, 13 = 128x256 on 4 waves,
14 = 128x128 ring kernel, 15 = 256x256 on 8 waves, 16 / 17 / 18 = loader + consumer waves (yk_igemm_lc.h) 256x128 / 128x128 / 128x256; /n = ring depth.
Every variant's network output is compared with the first variant's (max |difference|: the K order is the same, so 0 is expected)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k210_yolo_framework_amd import engine, netspec as ns

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(104, 104, 64, 128, 3), (52, 52, 128, 256, 3), (26, 26, 256, 512, 3), (13, 13, 512, 1024, 3), (52, 52, 256, 128, 1), (26, 26, 512, 256, 1)]
variants = [('', ''), ('14', '2'), ('11', '2'), ('11', '3'), ('13', '2'), ('15', '2')]
if len(sys.argv) > 2:
    variants = [tuple(v.split('/')) for v in sys.argv[2].split(',')]
os.environ['YK_FORCE_MINK'] = '64'
for (h, w, c1, c2, k) in shapes:
    s = ns.NetSpec('probe', (h, w), anchor_num=3, class_num=20)
    x = s._new_tensor(h, w, 3)
    x = s.conv(x, 32, 3, act=ns.LEAKY01, name='conv2d_1')
    x = s.conv(x, c1, 3, act=ns.LEAKY01, name='conv2d_2')
    y = s.conv(x, c2, k, act=ns.LEAKY01, name='conv2d_3')
    z = s.conv(y, 75, 1, bn=False, bias=True, name='conv2d_4', net_output=True)
    s.outputs = [z]
    wts = s.init_weights(1)
    f = torch.rand(B, h, w, 3, device='cuda')
    line = f'{h}x{w} {c1}->{c2} k{k} B={B} (TF/s): '
    ref_out = None
    for cfg, nsd in variants:
        for key, val in (('YK_IGEMM_FORCE', cfg), ('YK_NS', nsd), ('YK_SPLIT_FORCE', '1' if cfg else '')):
            if val:
                os.environ[key] = val
            else:
                os.environ.pop(key, None)
        try:
            plan = engine.Plan(s, wts, max_batch=B, precision='f16')
            plan.run_f32(f)
            torch.cuda.synchronize()
            out = plan.outputs()[0][:B].clone()
            if ref_out is None:
                ref_out = out
            dmax = float((out - ref_out).abs().max())
            ms = plan.profile(f, iters=5)
            hit = [(n, fl, t) for (n, fl, by), t in zip(plan.launches(), ms) if f'_{c1}to{c2}[' in n]
            n, fl, t = hit[0]
            line += f" {cfg or 'A'}/{nsd or '-'}={fl * B / t / 1e9:.0f}" + (f"[{n.split('[')[1].split(']')[0]}]" if not cfg else '') + (f'(!d={dmax:.3g})' if dmax != 0.0 else '')
            plan.close()
        except Exception as e:
            line += f" {cfg or 'A'}/{nsd or '-'}=ERR({str(e)[:60]})"
    print(line, flush=True)
