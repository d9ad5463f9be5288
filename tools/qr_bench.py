"""Time tn_qr on the pass-1 shape: single-level blocking vs two-level (outer 128 / 256) (ms per call, nominal TFLOP/s)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops

for (m, n) in [(16384, 1024), (8192, 1024)]:
    T = torch.randn((m, n), dtype=torch.float64, device='cuda')
    k = min(m, n)
    Q = torch.empty((m, k), dtype=torch.float64, device='cuda')
    R = torch.empty((k, n), dtype=torch.float64, device='cuda')
    for nbo in (0, 128, 256):
        os.environ['TN_QR_NBO'] = str(nbo)
        work = [T.clone() for _ in range(6)]
        ops.qr_into(work[0], Q, R, overwrite=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for w in work[1:]:
            ops.qr_into(w, Q, R, overwrite=True)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / 5
        fl = 4.0 * m * n * n - 4.0 / 3.0 * n ** 3
        print('tn_qr %6d x %5d outer block %3d  %.3f ms  %.2f nominal TFLOP/s' % (m, n, nbo, ms, fl / ms / 1e9), flush=True)
