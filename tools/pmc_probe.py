"""Tiny workload for PMC collection: absorb at the bulk shape (x5), one large GEMM (x3), ONE QR of an absorbed-bulk-site shape
(16384 x 1024, nb = 32: the Cholesky-QR panel chain cq_gram / cq_pass / cq_post and the trailing GEMMs) and ONE truncated SVD
of a centre matrix as the sweep produces them (the leading 320 rows of the triangular factor of a graded rank-300 matrix,
i.e. what the rank-revealing QR of a truncating pass hands over: eig_small + pair GEMMs).  Inputs are prepared with torch
(rocSOLVER / rocBLAS kernels, not counted).  Run under rocprofv3 --pmc in separate passes (tools/collect_profiles.sh);
tools/pmc_summary.py aggregates the counter CSV per kernel and per SEGMENT: a torch bitwise_xor launch (used nowhere else) marks
the boundaries -- segment 1 = the 16384 x 1024 QR, 2 = the 4096 x 512 QR, 4 = the truncated SVD (320 vectors: rounds as separate launches), 6 = sixteen one-launch 1024 x 64 QRs, 8 = a 192 x 900 truncated SVD (all rounds in one launch) -- so that whole-call traffic is the
sum over everything launched inside the call.  Prints the un-profiled timings of the QR and the SVD."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from tnac4o_amd import ops
torch.manual_seed(0)
_mk = torch.zeros(64, dtype=torch.int32, device='cuda')


def marker():
    torch.bitwise_xor(_mk, _mk, out=_mk)


A = torch.randn(64, 16, 64, dtype=torch.float64, device='cuda'); W = torch.randn(16, 16, 16, 16, dtype=torch.float64, device='cuda')
for _ in range(5):
    T = ops.absorb(A, W, True)
X = torch.randn(16384, 1024, dtype=torch.float64, device='cuda'); Y = torch.randn(1024, 1024, dtype=torch.float64, device='cuda')
for _ in range(3):
    Z = ops.mm(X, Y)
k = 1024
Q = torch.empty((16384, k), dtype=torch.float64, device='cuda'); Rr = torch.empty((k, 1024), dtype=torch.float64, device='cuda')
Xq = X.clone()
marker()
torch.cuda.synchronize()
t0 = time.perf_counter()
ops.qr_into(Xq, Q, Rr, overwrite=True)
torch.cuda.synchronize()
t_qr = time.perf_counter() - t0
marker()
# one QR whose panels take the single-launch form (<= 4096 rows)
X2 = torch.randn(4096, 512, dtype=torch.float64, device='cuda')
Q2 = torch.empty((4096, 512), dtype=torch.float64, device='cuda'); R2 = torch.empty((512, 512), dtype=torch.float64, device='cuda')
ops.qr_into(X2, Q2, R2, overwrite=True)
marker()
G0 = torch.randn(4096, 300, dtype=torch.float64, device='cuda') * (10.0 ** (-torch.arange(300, dtype=torch.float64, device='cuda') / 20.0))
_, Rfull = torch.linalg.qr(G0 @ torch.randn(300, 1024, dtype=torch.float64, device='cuda'))
C = Rfull[:320].contiguous()
marker()
torch.cuda.synchronize()
t0 = time.perf_counter()
out = ops.svd_trunc(C, 256, 1e-17)
torch.cuda.synchronize()
t_svd = time.perf_counter() - t0
marker()
# sixteen one-launch factorisations of the 1024 x 64 class (csrc/smallqr.hip; segment 5)
X3 = torch.randn(1024, 64, dtype=torch.float64, device='cuda')
Q3 = torch.empty((1024, 64), dtype=torch.float64, device='cuda'); R3 = torch.empty((64, 64), dtype=torch.float64, device='cuda')
ops.qr_into(X3, Q3, R3, overwrite=True)
torch.cuda.synchronize()
marker()
t0 = time.perf_counter()
for _ in range(16):
    ops.qr_into(X3, Q3, R3, overwrite=True)
torch.cuda.synchronize()
t_sq = (time.perf_counter() - t0) / 16
marker()
# a truncated SVD of the class the sweep runs most (192 x 900, chi = 64): all Jacobi rounds in ONE launch with the vectors resident in
# LDS (svdl_kernel; segment 8)
_, Rf2 = torch.linalg.qr(G0[:, :200] @ torch.randn(200, 900, dtype=torch.float64, device='cuda'))
C2 = Rf2[:192].contiguous()
out2 = ops.svd_trunc(C2, 64, 1e-16)
torch.cuda.synchronize()
marker()
t0 = time.perf_counter()
out2 = ops.svd_trunc(C2, 64, 1e-16)
torch.cuda.synchronize()
t_svd2 = time.perf_counter() - t0
marker()
print('done', float(T.sum()), float(Z.sum()), float(Rr.abs().sum()), out[1][:2])
print('PROBE_TIMES ' + json.dumps({'svd_trunc_320x1024_ms': 1e3 * t_svd, 'svd_sweeps': out[5]['sweeps'], 'svd_keep': out[3],
                                   'svd_preconditioned': bool(out[5].get('preconditioned')), 'qr_16384x1024_ms': 1e3 * t_qr, 'qr_1024x64_one_launch_us': 1e6 * t_sq,
                                   'svd_trunc_192x900_ms': 1e3 * t_svd2, 'svd_192x900_sweeps': out2[5]['sweeps'], 'svd_192x900_keep': out2[3]}))
