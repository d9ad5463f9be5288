"""Tiny workload for PMC collection: absorb at the bulk shape, one large GEMM, one QR of an absorbed-bulk-site shape
(16384 x 1024, nb = 32: tsqr_factor / tsqr_apply / lu_reconstruct / rows_times_small3 / trailing GEMMs) and one
truncated SVD (1024 x 1024: eig_small + pair GEMMs).  Run under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
separate passes (tools/pmc_summary.py aggregates the counter CSV per kernel)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
torch.manual_seed(0)
A = torch.randn(64, 16, 64, dtype=torch.float64, device='cuda'); W = torch.randn(16, 16, 16, 16, dtype=torch.float64, device='cuda')
for _ in range(5):
    T = ops.absorb(A, W, True)
X = torch.randn(16384, 1024, dtype=torch.float64, device='cuda'); Y = torch.randn(1024, 1024, dtype=torch.float64, device='cuda')
for _ in range(3):
    Z = ops.mm(X, Y)
Q, Rr = ops.qr(X.clone())
# graded spectrum like a centre matrix of the sweep
U0, _ = torch.linalg.qr(torch.randn(1024, 1024, dtype=torch.float64, device='cuda'))
S0 = torch.logspace(0, -14, 1024, dtype=torch.float64, device='cuda')
C = (U0 * S0) @ Y
out = ops.svd_trunc(C, 256, 1e-16)
torch.cuda.synchronize()
print('done', float(T.sum()), float(Z.sum()), float(Rr.abs().sum()), out[1][:2])
