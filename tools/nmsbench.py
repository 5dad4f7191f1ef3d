import sys; sys.path.insert(0,'.')
import numpy as np, torch
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
plan = engine.Plan(spec, spec.init_weights(seed=1), max_batch=32, precision='f16')
frames = torch.randint(0, 256, (32, 224, 320, 3), dtype=torch.uint8, device='cuda')
plan.run_u8(frames); torch.cuda.synchronize()
outs = plan.outputs()
cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
o = [x.cpu().numpy() for x in outs]
sc = np.concatenate([(1/(1+np.exp(-x.reshape(32,-1,3,25)[...,5:])))*(1/(1+np.exp(-x.reshape(32,-1,3,25)[...,4:5]))) for x in o],1).reshape(32,-1,20)
for thr in (0.7, 0.9, 0.99):
    n = (sc >= thr).sum(1)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    for _ in range(3): engine.decode_py(cfg, outs, 32, None, thr, 0.5)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): d,c = engine.decode_py(cfg, outs, 32, None, thr, 0.5)
    e1.record(); torch.cuda.synchronize()
    print(f'thr {thr}: candidates per (img,class) mean {n.mean():.0f} max {n.max()} ; dets/img {c.float().mean().item():.0f}; decode+nms+compact {e0.elapsed_time(e1)/20*1e3:.1f} us')
