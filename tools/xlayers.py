"""Dev tool: every materialised tensor of the f16x2 plan vs the fp32 oracle, in op order (first bad layer is the bug).

    python tools/xlayers.py [net] [H] [W] [B] [alpha]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from k210_yolo_framework_amd import engine, netspec as ns

ALL = '--all' in sys.argv
sys.argv = [v for v in sys.argv if v != '--all']
name = sys.argv[1] if len(sys.argv) > 1 else 'yolo_mobilev1'
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 96
B = int(sys.argv[4]) if len(sys.argv) > 4 else 3
alpha = float(sys.argv[5]) if len(sys.argv) > 5 else (0.75 if name == 'yolo_mobilev1' else 1.0)
spec = ns.NETWORKS[name]((H, W, 3), 3, 20, alpha=alpha)
w = spec.init_weights(seed=1)
frames = np.random.default_rng(0).integers(0, 256, (B, H, W, 3), dtype=np.uint8)
x = oracle.normalise_u8(frames)
plan = engine.Plan(spec, w, max_batch=B, precision='f16x2')
plan.run_u8(torch.from_numpy(frames).cuda())
torch.cuda.synchronize()
cp = spec.compile_plan(w)
names = [l[0] for l in plan.launches()]
print('\n'.join(names))
worst = 0.0
for op in spec.ops:
    if op['type'] not in (ns.OP_CONV, ns.OP_DWCONV, ns.OP_ADD, ns.OP_MAXPOOL):
        continue
    t = op['out']
    try:
        got = plan.read_tensor(t, B)
    except engine.YkError:
        continue
    if t in spec.outputs:
        ref = oracle.net_forward(cp, x, False, spec.outputs)[spec.outputs.index(t)]
    else:
        _, ref = oracle.net_forward(cp, x, False, spec.outputs, dump_id=t)
    got = got.reshape(ref.shape)
    scale = float(np.abs(ref).max())
    err = float(np.abs(got - ref).max())
    rel = err / max(scale, 1e-30)
    worst = max(worst, rel)
    flag = '' if rel < 1e-4 else '   <<<<<<'
    print(f'tensor {t:3d} type {op["type"]} {op.get("layer", ""):24s} shape {tuple(ref.shape)} max|ref| {scale:10.4g} err {err:10.3g} rel {rel:9.2g} finite {bool(np.isfinite(got).all())}{flag}')
    if rel > 1e-5 and '--detail' in os.environ.get('XL', ''):
        e = np.abs(got - ref)
        print('   per-channel max err (first 48):', np.array2string(e.reshape(-1, e.shape[-1]).max(0)[:48], precision=1, max_line_width=400))
        pe = e.max(-1)[0]
        print('   per-pixel max err image 0, rows 0-3:', np.array2string(pe[:4, :24], precision=1, max_line_width=400))
    if not (rel <= 1e-2) and not ALL:
        bad = np.argwhere(np.abs(got - ref) > 1e-3 * scale)
        print('   first bad indices', bad[:8].tolist(), ' count', len(bad), 'of', got.size)
        print('   got', got[tuple(bad[0])], 'ref', ref[tuple(bad[0])])
        break
print('worst rel', worst)
plan.close()
