import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
def t(f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for (M, N, K) in [(16384, 1024, 1024), (4096, 4096, 4096), (16384, 1024, 32), (32, 1024, 16384), (1024, 16384, 64), (4096, 256, 1024), (1024, 64, 1024), (16384, 896, 128), (128, 896, 16384), (16384, 96, 32), (32, 96, 16384), (128, 128, 16384)]:
    A = torch.randn(M, K, dtype=torch.float64, device='cuda'); B = torch.randn(K, N, dtype=torch.float64, device='cuda')
    C = torch.empty(M, N, dtype=torch.float64, device='cuda')
    mine = t(lambda: ops.mm(A, B, out=C)); ref = t(lambda: torch.matmul(A, B, out=C))
    fl = 2.0 * M * N * K
    print('%6d x %6d x %6d  tn_gemm %8.3f ms %6.2f TF | rocBLAS %8.3f ms %6.2f TF' % (M, N, K, mine * 1e3, fl / mine / 1e12, ref * 1e3, fl / ref / 1e12))
