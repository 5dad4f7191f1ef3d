"""The bench regime for profilers: engine.Pipeline with D batches in flight (what bench.py's `value` times), K steps, nothing else.
  python tools/inflight.py [steps=80] [depth=4] [graph=1] [precision=f16x2]
Writes gpurun_out/launch_names_inflight.json (launch names of one step in dispatch order) so counter rows can be matched."""
import json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
from k210_yolo_framework_amd import engine, netspec
from k210_yolo_framework_amd.helper import VOC_ANCHORS

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
graph = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
precision = sys.argv[4] if len(sys.argv) > 4 else 'f16x2'
B = int(os.environ.get('YK_BENCH_BATCH', '32'))
import numpy as np
NET = os.environ.get('YK_NET', 'yolo_mobilev1')                      # YK_NET=yolo|tiny_yolo|yolo_mobilev2: the other BASELINE configs
H, W, ALPHA = {'yolo_mobilev1': (224, 320, 0.75), 'yolo_mobilev2': (224, 320, 1.0), 'tiny_yolo': (416, 416, 1.0), 'yolo': (416, 416, 1.0)}[NET]
spec = netspec.NETWORKS[NET]((H, W, 3), 3, 20, alpha=ALPHA)
anchors = VOC_ANCHORS if len(spec.outputs) == 2 else np.concatenate([VOC_ANCHORS, VOC_ANCHORS[:1] * 0.5])
pipe = engine.Pipeline(spec, spec.init_weights(seed=1), anchors, max_batch=B, depth=depth, precision=precision, graph=graph)
frames = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))
for _ in range(3 * depth):
    pipe.submit(frames, sync_input=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.submit(frames, sync_input=False)
torch.cuda.synchronize()
el = time.perf_counter() - t0
names = [l[0] for l in pipe.plans[0].launches()] + ['decode_py', 'nms_py', 'compact_py']
os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
json.dump({'launches': names, 'steps': steps, 'depth': depth, 'graph': graph, 'precision': precision, 'warm_steps': 3 * depth,
           'images_per_sec': round(B * steps / el, 1), 'alg_bytes_per_image': [l[2] for l in pipe.plans[0].launches()],
           'alg_flops_per_image': [l[1] for l in pipe.plans[0].launches()]},
          open(os.path.join(root, 'gpurun_out', 'launch_names_inflight.json'), 'w'))
print(f'inflight {NET} depth={depth} graph={graph} {precision}: {B * steps / el:.0f} images/s over {steps} steps, {el / steps * 1e3:.3f} ms/step')
pipe.close()
