"""Dev tool (developer build): sweep the fused-block geometry (TM, TN, tile width, stages) of ONE block at a time and print the measured
time of that launch per configuration.   python tools/xbsweep.py [first_block] [last_block] [net] [alpha]"""
import itertools
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
last = int(sys.argv[2]) if len(sys.argv) > 2 else 10
net = sys.argv[3] if len(sys.argv) > 3 else 'yolo_mobilev1'
alpha = float(sys.argv[4]) if len(sys.argv) > 4 else 0.75
spec = ns.NETWORKS[net]((224, 320, 3), 3, 20, alpha=alpha)
w = spec.init_weights(seed=1)
frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (32, 224, 320, 3), dtype=np.uint8)).cuda()


def run():
    plan = engine.Plan(spec, w, max_batch=32, precision='f16x2')
    ms = plan.profile(frames, iters=10)
    names = [l[0] for l in plan.launches()]
    plan.close()
    return names, ms


os.environ['YK_X_NOFUSE'] = '1'
names0, ms0 = run()
del os.environ['YK_X_NOFUSE']
dw0 = [i for i, n in enumerate(names0) if n.startswith('x:dw3x3')]
tns = tuple(int(v) for v in os.environ.get('XBS_TN', '1,2,3,6').split(','))
dbs = tuple(int(v) for v in os.environ.get('XBS_DB', '0,1').split(','))
for li in range(first, last + 1):
    if li >= len(dw0):
        break
    unf = (ms0[dw0[li]] + ms0[dw0[li] + 1]) * 1e3
    rows = []
    for tm, tn, tw, db in itertools.product((2, 3, 4, 5, 8), tns, (4, 5, 8, 10, 16, 20), dbs):
        if tm * tn > 24:
            continue
        os.environ.update(YK_XB_LAYER=str(li), YK_XB_TM=str(tm), YK_XB_TN=str(tn), YK_XB_TW=str(tw), YK_XB_DB=str(db), YK_XB_ALWAYS='1')
        names, ms = run()
        blocks = [i for i, n in enumerate(names) if 'dw3x3' in n]
        i = blocks[li]
        if '+conv1x1' in names[i] and f',{2 if db else 1}stage' in names[i]:
            rows.append((float(ms[i]) * 1e3, names[i], tm, tn, tw, db))
    rows.sort()
    if os.environ.get('XBS_ALL'):
        for t, n, tm, tn, tw, db in rows:
            print(f'   {t:6.1f} us  tm{tm} tn{tn} tw{tw} db{db}  {n.split("[")[1][:-1]}')
    print(f'block {li} {names0[dw0[li]]} + {names0[dw0[li] + 1]}: unfused {unf:.1f} us | ' +
          ' | '.join(f'{t:.1f} tm{tm} tn{tn} {n.split("[")[1][:-1]}' for t, n, tm, tn, tw, db in rows[:5]), flush=True)
