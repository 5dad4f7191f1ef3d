"""Calibration only (NOT part of the product path, which never calls a BLAS): what the vendor's tuned GEMM (torch.matmul -> hipBLASLt) reaches on
this box for the plain-GEMM equivalents of the Darknet-53 layers the implicit-GEMM kernels are measured on - the practical ceiling of this part
for these shapes, to read profiles/r05_igemm_sweep.txt against."""
import sys, time
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [(104, 104, 64, 128, 3), (52, 52, 128, 256, 3), (26, 26, 256, 512, 3), (13, 13, 512, 1024, 3), (52, 52, 256, 128, 1), (26, 26, 512, 256, 1)]
for h, w, c1, c2, k in shapes:
    M, N, K = B * h * w, c2, c1 * k * k
    a = torch.randn(M, K, device='cuda', dtype=torch.float16)
    b = torch.randn(N, K, device='cuda', dtype=torch.float16)
    for _ in range(5):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'{h}x{w} {c1}->{c2} k{k} B={B}: plain GEMM {M}x{N}x{K} fp16 via hipBLASLt {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s', flush=True)
