"""Dev tool: three runs of the f16x2 plan of the headline network (for rocprofv3 passes)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from k210_yolo_framework_amd import engine, netspec as ns
spec = ns.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
w = spec.init_weights(seed=1)
frames = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (32, 224, 320, 3), dtype=np.uint8)).cuda()
plan = engine.Plan(spec, w, max_batch=32, precision='f16x2')
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    plan.run_u8(frames)
torch.cuda.synchronize()
plan.close()
