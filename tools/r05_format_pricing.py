#!/usr/bin/env python
"""price the early-layer byte problem of the f16x2 mode (4 bytes per stored element).  CPU only.

For yolo_mobilev1-0.75 (configs[1]) the tensors BETWEEN the six fused blocks are 64 % of the step's stored bytes.  Options for them:
  A  (hi | lo) fp16 pair / fp32 planes, 4 B per element (shipped)          22 significant bits
  B  fp16 hi + 8-bit residual, 3 B per element                             ~19 bits
  C  plain fp16, 2 B per element (what the f16 mode stores everywhere)      11 bits
This script measures what each costs in ACCURACY on the north-star bar (float64 torch build of the Keras graph, the chosen tensors rounded
to p significant bits when stored, everything else exact): logit error, decoded score error, and how many (class, box) detections change
against the exact run.  The time side is arithmetic on measured numbers (profiles/r05_x2_kernel_trace_per_launch.csv) and is in DESIGN.md.

    python tools/r05_format_pricing.py [n_images]
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
from oracle import decode_ref, torch_net_ref  # noqa: E402
from k210_yolo_framework_amd import netspec as ns  # noqa: E402
from k210_yolo_framework_amd.helper import VOC_ANCHORS  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
frames = np.random.default_rng(0).integers(0, 256, (n, 224, 320, 3), dtype=np.uint8)
x = oracle.normalise_u8(frames)
# tensors whose only reader is a depthwise conv of a fused block = outputs of the pointwise convs of blocks 1..6 (>= 128 pixels per image rule)
lay = {l.name: l for l in spec.layers}
between = []
for i, op in enumerate(spec.ops):
    if op['type'] == ns.OP_CONV and op['k'] == 1 and i + 1 < len(spec.ops) and spec.ops[i + 1]['type'] == ns.OP_DWCONV and spec.ops[i + 1]['in0'] == op['out']:
        h, ww, c = spec.tensors[op['out']]
        if c <= 384 and h * ww >= 280:
            between.append(op['out'])
between = between[:6]
elems = {t: int(np.prod(spec.tensors[t])) for t in between}
total_stored = sum(int(np.prod(spec.tensors[op['out']])) for op in spec.ops if op['type'] in (ns.OP_CONV, ns.OP_DWCONV))


def round_bits(y, p):
    m, e = torch.frexp(y)
    return torch.ldexp(torch.round(m * (2.0 ** p)) / (2.0 ** p), e)


def run(p, which):
    hook = None if p is None else (lambda tid, y: round_bits(y, p) if tid in which else y)
    outs = torch_net_ref.forward(spec, w, x, dtype=torch.float64, store_hook=hook)
    preds = [outs[o] for o in spec.outputs]
    det = decode_ref.decode_batch_fast([q.reshape(n, q.shape[1], q.shape[2], 3, 25) for q in preds], VOC_ANCHORS, (224, 320), (224, 320), 0.7, 0.5, threads=4)
    return preds, det


ref, ref_det = run(None, ())
rows = []
for name, p, which in (('A: (hi|lo) fp16 pair, 4 B (shipped)', 22, between), ('B: fp16 + 8-bit residual, 3 B', 19, between), ('C: plain fp16, 2 B', 11, between),
                       ('C on the first two tensors only (112x160x48, 56x80x96)', 11, between[:2])):
    preds, det = run(p, set(which))
    lerr = max(float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(preds, ref))
    changed, serr, nref = 0, 0.0, 0
    for (d, ix), (rd, rix) in zip(det, ref_det):
        a = set(zip(d[:, 5].astype(int).tolist(), ix.tolist()))
        b = set(zip(rd[:, 5].astype(int).tolist(), rix.tolist()))
        changed += len(a ^ b)
        nref += len(b)
        sc = {(int(c), int(i)): s for c, i, s in zip(rd[:, 5], rix, rd[:, 4])}
        for c, i, s in zip(d[:, 5], ix, d[:, 4]):
            if (int(c), int(i)) in sc:
                serr = max(serr, abs(float(s) - float(sc[(int(c), int(i))])))
    bytes_img = sum(elems[t] for t in which) * {22: 4, 19: 3, 11: 2}[p]
    rows.append({'format': name, 'bits': p, 'logit_err_of_max': lerr, 'max_score_err': serr, 'detections_changed': changed, 'of': nref,
                 'stored_MB_per_image_these_tensors': round(bytes_img / 1e6, 3)})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({'tensors': between, 'elems_per_image': elems, 'share_of_stored_elements': round(sum(elems.values()) / total_stored, 3), 'images': n}))
