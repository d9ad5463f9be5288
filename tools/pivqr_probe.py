"""The pivoted, truncating site QR of the weighted first pass (tn_site_qr side 1, pivot, Frobenius exit) on a synthetic site whose
rows are graded like a weighted site's (numerical rank ~ keep): time per call, accepted rank, per-family kernel time.
Usage: pivqr_probe.py [Dl p r keep]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops, _lib
import bench

Dl, p, r, keep = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (1024, 136, 64, 160)
L = _lib.lib()
g = torch.Generator(device='cuda').manual_seed(5)
m = p * r
U = torch.linalg.qr(torch.randn((Dl, 320), generator=g, dtype=torch.float64, device='cuda'))[0]
V = torch.linalg.qr(torch.randn((m, 320), generator=g, dtype=torch.float64, device='cuda'))[0]
s = torch.logspace(0, -20, 320, dtype=torch.float64, device='cuda')          # one decade per 16 values
B = ((U * s[None, :]) @ V.t()).contiguous()
rows = B.norm(dim=1)
B = B[torch.argsort(rows, descending=True)].contiguous().view(Dl, p, r)
tol = float(s[keep])
for rep in range(3):
    info = {}
    torch.cuda.synchronize()
    if rep == 2:
        L.tn_profile_reset(); L.tn_profile_enable((1 << len(bench.FAMILIES)) - 1)
    t0 = time.perf_counter()
    Q, R, k, nf = ops.site_qr(1, B.clone(), None, rank_tol=tol, normalise=False, info=info, frobenius_exit=True, pivot=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('site (%d, %d, %d): m = %d, n = %d, accepted rank %d, %.3f ms (incl. the clone), dropped %.2e' % (Dl, p, r, m, Dl, k, 1e3 * dt, info['dropped2'] ** 0.5), flush=True)
tot = bench.profile_totals(L)
L.tn_profile_enable(0)
print('kernel ms %.3f, launches %d' % (sum(t['ms'] for t in tot), sum(t['calls'] for t in tot)))
for t in tot:
    if t['calls']:
        extra = ''
        if t['flops'] and t['ms']:
            extra = '  %.1f TFLOP/s' % (t['flops'] / t['ms'] / 1e9)
        if t.get('bytes') and t['ms']:
            extra += '  %.0f GB/s' % (t['bytes'] / t['ms'] / 1e6)
        print('   %-60s %6d launches %8.3f ms%s' % (t['kernel'], t['calls'], t['ms'], extra))
err = (Q @ Q.t() - torch.eye(k, dtype=torch.float64, device="cuda")).abs().max()
print('orthonormality of the basis: %.2e' % float(err))
