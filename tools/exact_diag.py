"""Dev diagnostic: f16x2 plan vs the fp32 oracle (logit error, decoded detections) for the four networks."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import oracle
from oracle import decode_ref as dr
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

cases = [('yolo_mobilev1', (224, 320, 3), 0.75, 8), ('yolo_mobilev2', (224, 320, 3), 1.0, 2), ('tiny_yolo', (224, 320, 3), 1.0, 2),
         ('yolo', (96, 128, 3), 1.0, 2)]
for name, shape, alpha, B in cases:
    spec = ns.NETWORKS[name](shape, 3, 20, alpha=alpha)
    w = spec.init_weights(seed=1)
    frames = np.random.default_rng(0).integers(0, 256, (B, *shape), dtype=np.uint8)
    x = oracle.normalise_u8(frames)
    cp = spec.compile_plan(w)
    ref = oracle.net_forward(cp, x, False, spec.outputs)
    for prec in ('f16x2', 'f16'):
        plan = engine.Plan(spec, w, max_batch=B, precision=prec)
        t = torch.from_numpy(frames).cuda()
        plan.run_u8(t)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            plan.run_u8(t)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        outs = [o[:B].cpu().numpy() for o in plan.outputs()]
        msg = f'{name:14s} {prec:6s} {dt*1e3:7.2f} ms'
        for o, r in zip(outs, ref):
            msg += f' | max|ref| {np.abs(r).max():9.3g} err max {np.abs(o - r).max():9.3g} rel {np.abs(o - r).max() / np.abs(r).max():8.2g}'
        print(msg, flush=True)
        if len(spec.outputs) == 2 and shape[:2] == (224, 320):
            e = 25
            rd = dr.decode_batch([r.reshape(B, r.shape[1], r.shape[2], 3, e) for r in ref], VOC_ANCHORS, shape[:2], shape[:2], 0.7, 0.5)
            gd = dr.decode_batch([o.reshape(B, o.shape[1], o.shape[2], 3, e) for o in outs], VOC_ANCHORS, shape[:2], shape[:2], 0.7, 0.5)
            same = all(len(a[0]) == len(b[0]) and np.array_equal(a[0][:, 5], b[0][:, 5]) for a, b in zip(rd, gd))
            nd = sum(len(a[0]) for a in rd)
            if same and nd:
                es = max(np.abs(a[0][:, 4] - b[0][:, 4]).max() for a, b in zip(rd, gd) if len(a[0]))
                eb = max(np.abs(a[0][:, :4] - b[0][:, :4]).max() for a, b in zip(rd, gd) if len(a[0]))
                print(f'    detections {nd}: identical class sequence; max score err {es:.3g}, max coord err {eb:.3g} px')
            else:
                print(f'    detections ref {nd} got {sum(len(a[0]) for a in gd)}: class sequences differ')
        plan.close()
