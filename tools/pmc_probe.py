"""Tiny workload for PMC collection (absorb at the bulk shape, one large GEMM, one big QR panel chain)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
A = torch.randn(64, 16, 64, dtype=torch.float64, device='cuda'); W = torch.randn(16, 16, 16, 16, dtype=torch.float64, device='cuda')
for _ in range(5):
    T = ops.absorb(A, W, True)
X = torch.randn(16384, 1024, dtype=torch.float64, device='cuda'); Y = torch.randn(1024, 1024, dtype=torch.float64, device='cuda')
for _ in range(3):
    Z = ops.mm(X, Y)
torch.cuda.synchronize()
print('done', float(T.sum()), float(Z.sum()))
