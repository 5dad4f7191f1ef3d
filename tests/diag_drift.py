"""Diagnostic (not a test): per-tensor error statistics GPU vs oracle, and decoded-quantity drift."""
import sys
import numpy as np
sys.path.insert(0, '.')
import torch
import oracle
from oracle import decode_ref as dr
from k210_yolo_framework_amd import engine, netspec as ns
from k210_yolo_framework_amd.helper import VOC_ANCHORS

name = sys.argv[1] if len(sys.argv) > 1 else 'yolo_mobilev1'
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 0.75
spec = ns.NETWORKS[name]((224, 320, 3), 3, 20, alpha=alpha)
w = spec.init_weights(seed=1)
B = 4
frames = np.random.default_rng(0).integers(0, 256, (B, 224, 320, 3), dtype=np.uint8)
plan = engine.Plan(spec, w, max_batch=B, precision='f16')
plan.run_u8(torch.from_numpy(frames).cuda())
torch.cuda.synchronize()
x = oracle.normalise_u8(frames)
cp = spec.compile_plan(w)
for op in spec.ops:
    if op['type'] not in (ns.OP_CONV, ns.OP_DWCONV, ns.OP_ADD):
        continue
    t = op['out']
    try:
        got = plan.read_tensor(t, B)
    except engine.YkError:
        continue
    _, r16 = oracle.net_forward(cp, x, True, spec.outputs, dump_id=t)
    _, r32 = oracle.net_forward(cp, x, False, spec.outputs, dump_id=t)
    s = np.abs(r32).max()
    e16, e32 = np.abs(got - r16), np.abs(got - r32)
    print(f't{t:3d} {op.get("layer")!s:22s} scale {s:8.3f} rms {np.sqrt((r32**2).mean()):7.3f} | vs emu: max {e16.max()/s:.2e} rms {np.sqrt((e16**2).mean())/s:.2e} '
          f'| vs f32: max {e32.max()/s:.2e} rms {np.sqrt((e32**2).mean())/s:.2e}')
outs = [o[:B].cpu().numpy() for o in plan.outputs()]
r32 = oracle.net_forward(cp, x, False, spec.outputs)
for o, r in zip(outs, r32):
    print('logits: rms', np.sqrt((r**2).mean()), 'max', np.abs(r).max(), 'abs err max', np.abs(o - r).max(), 'rms', np.sqrt(((o - r)**2).mean()))
sg = [1 / (1 + np.exp(-o)) for o in outs]
sr = [1 / (1 + np.exp(-r)) for r in r32]
for a, b in zip(sg, sr):
    print('sigmoid err max', np.abs(a - b).max())
