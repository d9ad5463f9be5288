"""Per-kernel-family time of one tn_qr 16384 x 1024 for the single-level and the two-level (outer 128 / 256) factorisation."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops, _lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

L = _lib.lib()
m, n = 16384, 1024
T = torch.randn((m, n), dtype=torch.float64, device='cuda')
Q = torch.empty((m, n), dtype=torch.float64, device='cuda'); R = torch.empty((n, n), dtype=torch.float64, device='cuda')
for nbo in (0, 128, 256):
    os.environ['TN_QR_NBO'] = str(nbo)
    ops.qr_into(T.clone(), Q, R, overwrite=True)
    torch.cuda.synchronize()
    L.tn_profile_reset(); L.tn_profile_enable((1 << len(bench.FAMILIES)) - 1)
    ops.qr_into(T.clone(), Q, R, overwrite=True)
    torch.cuda.synchronize()
    tot = bench.profile_totals(L)
    L.tn_profile_enable(0)
    print('outer block %d: total kernel ms %.2f, launches %d' % (nbo, sum(t['ms'] for t in tot), sum(t['calls'] for t in tot)))
    for t in tot:
        if t['calls']:
            extra = ''
            if t['flops'] and t['ms']:
                extra = '  %.1f TFLOP/s' % (t['flops'] / t['ms'] / 1e9)
            print('   %-60s %6d launches %8.3f ms%s' % (t['kernel'], t['calls'], t['ms'], extra))
