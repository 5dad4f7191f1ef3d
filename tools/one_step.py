"""A few bench steps for profilers (rocprofv3 --stats / --pmc): same plan, frames and decode call as bench.py.
Writes the launch names of one step (dispatch order) to gpurun_out/launch_names.json so counter rows can be matched."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from k210_yolo_framework_amd import engine, netspec
from k210_yolo_framework_amd.helper import VOC_ANCHORS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
precision = sys.argv[2] if len(sys.argv) > 2 else 'f16x2'
schedule = sys.argv[3] if len(sys.argv) > 3 else 'throughput'          # the plan behind bench.py's `value` | 'latency': its one-batch plan
spec = netspec.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
plan = engine.Plan(spec, spec.init_weights(seed=1), max_batch=32, precision=precision, schedule=schedule)
cfg = engine.make_decode_cfg(VOC_ANCHORS, 20, spec.in_hw, spec.out_hw())
frames = torch.randint(0, 256, (32, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
outs = plan.outputs()
torch.cuda.synchronize()
for _ in range(steps):
    plan.run_u8(frames)
    engine.decode_py(cfg, outs, 32, None, 0.7, 0.5)
torch.cuda.synchronize()
names = [l[0] for l in plan.launches()] + ['decode_py', 'nms_py', 'compact_py']
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
json.dump({'launches': names, 'alg_bytes_per_image': [l[2] for l in plan.launches()], 'steps': steps, 'precision': precision, 'schedule': schedule,
           'kernels_per_launch': [2 if ('splitk' in n and not (n.startswith('x:conv') and '+conv1x1_' in n)) else 1 for n in names]},   # (a fused head is ONE kernel)
          open(os.path.join(root, 'gpurun_out', f'launch_names_{schedule}.json'), 'w'))
