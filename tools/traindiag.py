"""Per-layer gradient error + activation-gate flips of the HIP training step vs oracle/train_ref.py (diagnostic)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import tests.test_gpu_train as t
from oracle import train_ref
from k210_yolo_framework_amd.train import Trainer
from k210_yolo_framework_amd import netspec as ns

name, hw, B, alpha, seed = sys.argv[1], (int(sys.argv[2]), int(sys.argv[3])), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])
spec, w, h, x, yt = t._case(name, hw, B, alpha, seed)
d, r, g, st, pr = train_ref.loss_and_grads(spec, w, x, yt, h.anchors, want_pre=True)
tr = Trainer(spec, w, h.anchors, B)
out = tr.loss_and_grads(t._cu(x), [t._cu(y) for y in yt])
got = tr.grads()
gmax = max(np.abs(v).max() for v in g.values())
for i, op in enumerate(spec.ops):
    if op['type'] not in (ns.OP_CONV, ns.OP_DWCONV):
        continue
    l = tr.lay[op['layer']]
    line = f"{i:3d} {l.name:28s}"
    for suf in ['/kernel'] + ([l.bn_name + '/gamma'] if l.bn_name else []):
        k = l.name + suf if suf == '/kernel' else suf
        e = np.abs(got[k] - g[k]).max() / max(np.abs(g[k]).max(), 1e-5 * gmax)
        line += f" {suf.split('/')[-1]}:{e:.2e}"
    if l.bn_name and i in tr.saved:
        sv = tr.saved[i]
        z = sv['z'].cpu().numpy().astype(np.float64)
        co = z.shape[-1]
        pre = (z - sv['mean'].cpu().numpy()) * sv['invstd'].cpu().numpy() * tr.view(tr.P, l.bn_name + '/gamma').cpu().numpy() \
            + tr.view(tr.P, l.bn_name + '/beta').cpu().numpy()
        rp = st[l.name + '/pre']
        flips = 0
        if op['act'] != ns.ACT_NONE:
            flips = int(((pre > 0) != (rp > 0)).sum())
            if op['act'] == ns.ACT_RELU6:
                flips += int(((pre < 6) != (rp < 6)).sum())
        line += f" pre_err:{np.abs(pre - rp).max():.1e} flips:{flips}"
    print(line)
