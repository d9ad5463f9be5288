"""One-launch factorisation (csrc/smallqr.hip) against the blocked path: per case passes / fallbacks, time per call of a dependent
chain of calls (TN_QR_SMALL=0/1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from tnac4o_amd import ops
import test_gpu_kernels as tk

for name, Th in tk._smallqr_cases():
    T = Th.cuda()
    ops.smallqr_stats(reset=True)
    Q, R = ops.qr(T)
    st = ops.smallqr_stats()
    print('%-18s %5d x %3d  passes %d fallback %d' % (name, T.shape[0], T.shape[1], st['passes'], st['householder_fallbacks']))

g = torch.Generator(device='cpu').manual_seed(1)
for m, n in ((1024, 64), (1024, 32), (512, 64), (256, 64), (128, 64), (2048, 64), (4096, 64), (300, 20), (2048, 32), (8192, 32), (720, 43)):
    T = torch.randn(m, n, dtype=torch.float64, generator=g).cuda()
    Q = torch.empty(m, n, dtype=torch.float64, device='cuda')
    R = torch.empty(n, n, dtype=torch.float64, device='cuda')
    res = []
    for small in ('1', '0'):
        os.environ['TN_QR_SMALL'] = small
        for _ in range(5):
            ops.qr_into(T, Q, R, overwrite=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ops.qr_into(T, Q, R, overwrite=False)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    print('%5d x %3d   one launch %7.1f us   blocked %7.1f us' % (m, n, res[0], res[1]))
