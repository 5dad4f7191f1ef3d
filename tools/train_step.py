"""configs[3] training step alone (what bench.py reports as secondary.train), timed three ways:
  step_ms      tr.step() as bench.py does (ends in a device->host read of the loss)
  replay_ms    the captured forward+loss+backward graph alone, back to back (device time per replay)
  launches     kernels per captured step (graph nodes)
  python tools/train_step.py [steps=30]"""
import os, sys, time, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tr, x, y = bench._train_setup(16, 0, 1, 0)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(x, y)
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / steps * 1e3
g = tr._graph
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(steps):
    g.replay()
e1.record(); torch.cuda.synchronize()
replay_ms = e0.elapsed_time(e1) / steps
out = dict(step_ms=round(step_ms, 3), replay_ms=round(replay_ms, 3), lib=os.environ.get('YK_LIB_PATH', 'in-tree'))
print(json.dumps(out))
