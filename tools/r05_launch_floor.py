"""What a small dependent kernel costs inside a replayed hipGraph on this box (no profiler attached): N x yk_axpy_f32 on a tiny vector (each
launch depends on the previous one through y), captured once, replayed.  The training step is ~690 launches; rocprofv3 shows every small kernel at
~5 us - this says what they cost when nobody is watching."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k210_yolo_framework_amd import engine
L = engine.lib()
x = torch.ones(4096, device='cuda'); y = torch.zeros(4096, device='cuda')
big_x = torch.ones(16 * 280 * 384, device='cuda'); big_y = torch.zeros_like(big_x)
side = torch.cuda.Stream()
for name, xs, ys in (('4 K floats', x, y), ('1.7 M floats (14x20x384 x 16 images)', big_x, big_y)):
    for n in (100, 400):
        with torch.cuda.stream(side):
            st = C.c_void_p(side.cuda_stream)
            for _ in range(3):
                L.yk_axpy_f32(C.c_longlong(xs.numel()), C.c_float(1e-6), C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()), st)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(n):
                    L.yk_axpy_f32(C.c_longlong(xs.numel()), C.c_float(1e-6), C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f'{name}: graph of {n} dependent axpy launches: {dt * 1e6:8.1f} us per replay = {dt * 1e6 / n:5.2f} us per launch', flush=True)
