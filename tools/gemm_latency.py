"""tn_gemm on the shapes of a sweep's census (TN_GEMM_TRACE=1) and on the mid-K attach shapes: time per call of back-to-back
launches on one stream, checked against torch.  TN_GEMM_BK=16 / 32 forces the K step of the kernel (default: by grid size).
Usage: gemm_latency.py [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = 'cuda'
# M N K batch transA transB  (transA: A is stored K-major, i.e. the view is a transpose of a row-major K x M array)
SHAPES = [
    (64, 64, 416, 2, 1, 1), (64, 64, 544, 3, 1, 1), (64, 544, 64, 2, 0, 0), (64, 736, 64, 3, 0, 0),
    (32, 32, 992, 1, 0, 0), (32, 64, 1024, 1, 0, 0), (32, 32, 992, 1, 1, 1), (992, 32, 32, 1, 0, 0), (1024, 64, 32, 1, 1, 0),
    (64, 64, 1024, 1, 1, 0), (128, 128, 1024, 1, 1, 0), (192, 192, 512, 1, 0, 0), (256, 256, 256, 1, 0, 0), (512, 64, 512, 1, 1, 0),
    (1024, 64, 1024, 1, 0, 0), (64, 1024, 1024, 1, 1, 0), (1024, 1024, 64, 1, 0, 0), (2048, 32, 2048, 1, 1, 0), (32, 992, 2048, 1, 1, 0),
    (2048, 992, 32, 1, 0, 0), (8192, 32, 32, 1, 0, 0),
    (256, 64, 256, 1024, 1, 0), (256, 1024, 256, 64, 1, 0),
    (64, 16384, 1024, 1, 0, 0), (16384, 1024, 64, 1, 1, 0), (16384, 544, 1024, 1, 1, 0),
    (16384, 128, 1024, 1, 0, 0), (16384, 256, 1024, 1, 0, 0), (16384, 1024, 128, 1, 0, 0), (16384, 1024, 256, 1, 0, 0),
    (1024, 1024, 16384, 1, 1, 0), (16384, 1024, 1024, 1, 0, 0),
]
# the rank-32 trailing update of a column-major matrix (A -= W X: C, A column-major): M N K
COLMAJOR = [(8704, 1024, 32), (16384, 992, 32), (2048, 64, 32), (8704, 512, 128)]

print('%6s %6s %6s %5s tA tB : %9s %8s   max|err|' % ('M', 'N', 'K', 'batch', 'us/call', 'TFLOP/s'))
for (M, N, K, b, ta, tb) in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M * 31 + N * 7 + K)
    A = (torch.randn((b, K, M), generator=g, dtype=torch.float64, device=dev).transpose(1, 2) if ta
         else torch.randn((b, M, K), generator=g, dtype=torch.float64, device=dev))
    B = (torch.randn((b, N, K), generator=g, dtype=torch.float64, device=dev).transpose(1, 2) if tb
         else torch.randn((b, K, N), generator=g, dtype=torch.float64, device=dev))
    out = torch.empty((b, M, N), dtype=torch.float64, device=dev)
    if b == 1:
        fn = lambda: ops.mm(A[0], B[0], out=out[0])
    else:
        fn = lambda: ops.bmm(A, B, out=out)
    fn()
    ref = torch.matmul(A, B)
    err = float((out - ref).abs().max())
    n = reps if M * N * K * b < 1 << 32 else max(5, reps // 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%6d %6d %6d %5d  %d  %d : %9.2f %8.2f   %.2e' % (M, N, K, b, ta, tb, 1e6 * dt, 2.0 * M * N * K * b / dt / 1e12, err), flush=True)

print('column-major C and A (the QR of a transposed site):')
for (M, N, K) in COLMAJOR:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    A = torch.randn((K, M), generator=g, dtype=torch.float64, device=dev).t()             # M x K, column-major
    B = torch.randn((K, N), generator=g, dtype=torch.float64, device=dev)
    Cc = torch.randn((N, M), generator=g, dtype=torch.float64, device=dev).t()            # M x N, column-major
    ref = Cc - A @ B
    out = Cc.clone().t().contiguous().t()
    ops.mm(A, B, out=out, alpha=-1.0, beta=1.0)
    err = float((out - ref).abs().max())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.mm(A, B, out=out, alpha=-1.0, beta=1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('%6d %6d %6d : %9.2f us  %6.2f TFLOP/s  %7.1f GB/s (C read + written)   %.2e' % (M, N, K, 1e6 * dt, 2.0 * M * N * K / dt / 1e12, 16.0 * M * N / dt / 1e9, err), flush=True)
