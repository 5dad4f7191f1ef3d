"""Dev probe: is the training step host-bound?  Times eager steps vs a HIP-graph replay of the same launches."""
import sys, time
sys.path.insert(0, '.')
import torch
from bench import _train_setup
tr, x, y = _train_setup(16, 0, 1, 0)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(x, y)
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t0) / 10 * 1e3)
import ctypes as C
from k210_yolo_framework_amd import engine
def body():
    r = tr.loss_and_grads(x, y)
    tr._ck(tr.L.yk_adam_f32(C.c_longlong(tr.n_params), engine._ptr(tr.P), engine._ptr(tr.G), engine._ptr(tr.m), engine._ptr(tr.v),
                            C.c_float(tr.lr), C.c_float(tr.decay), C.c_longlong(tr.iterations), C.c_float(0.9), C.c_float(0.999),
                            C.c_float(1e-7), C.c_float(1.0), tr._s()), 'adam')
    return r
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        r = body()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
print('graph ms/step', (time.perf_counter() - t0) / 10 * 1e3)
