"""Dev tool: what the host -> device copy path of this box delivers on its own - pinned host memory, 6.9 MB per copy (32 frames of 224x320x3),
1 / 2 / 4 streams, nothing else running - against the rate `value_from_host` needs (images/s x 215 KB).

    python tools/r05_h2d.py
"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from k210_yolo_framework_amd import engine

L = engine.lib()
n = 32 * 224 * 320 * 3
for S in (1, 2, 4):
    hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(S)]
    ds = [torch.empty(n, dtype=torch.uint8, device='cuda') for _ in range(S)]
    st = [torch.cuda.Stream() for _ in range(S)]
    for via in ('torch', 'yk_memcpy_async'):
        def go(k):
            for i in range(k):
                j = i % S
                if via == 'torch':
                    with torch.cuda.stream(st[j]):
                        ds[j].copy_(hs[j], non_blocking=True)
                else:
                    engine._check(L.yk_memcpy_async(C.c_void_p(ds[j].data_ptr()), C.c_void_p(hs[j].data_ptr()), C.c_size_t(n), C.c_void_p(st[j].cuda_stream)), 'copy')
        go(8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 200
        go(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'{S} stream(s), {via:16s}: {K * n / dt / 1e9:6.1f} GB/s  = {K * 32 / dt / 1e3:6.1f} k images/s of frames')
