"""Single-launch panel step (cq_fused_kernel) against the six-launch chain: time per panel of tn_panel_orth (orthonormalisation
only) and of tn_qr per panel (orthonormalisation + reconstruction + trailing update) at the row counts of the truncating passes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


def main():
    dev = 'cuda'
    for (n, b) in [(4096, 32), (2048, 32), (1024, 32), (512, 32), (256, 32), (64, 32)]:
        X = torch.randn(n, b, dtype=torch.float64, device=dev)
        Xi = X @ torch.diag(torch.logspace(0, -6, b, dtype=torch.float64, device=dev)) @ torch.randn(b, b, dtype=torch.float64, device=dev)
        Y = torch.empty_like(X)
        for name, M in (('well-conditioned (1 pass)', X), ('kappa 1e6 (2-3 passes)', Xi)):
            res = {}
            for fused in ('0', '1'):
                os.environ['TN_PANEL_FUSED'] = fused
                res[fused] = timeit(lambda: ops.panel_orth(M, 0, out=Y), 300)
            print('panel_orth %5d x %2d %-28s chain %6.1f us   single launch %6.1f us' % (n, b, name, res['0'], res['1']), flush=True)
    for (m, n) in [(4096, 512), (4096, 256), (2048, 256), (1024, 128), (4096, 1024)]:
        T = torch.randn((m, n), dtype=torch.float64, device=dev)
        k = min(m, n)
        Q = torch.empty((m, k), dtype=torch.float64, device=dev)
        R = torch.empty((k, n), dtype=torch.float64, device=dev)
        res = {}
        for fused in ('0', '1'):
            os.environ['TN_PANEL_FUSED'] = fused
            res[fused] = timeit(lambda: ops.qr_into(T.clone(), Q, R, overwrite=True), 20)
        npan = (k + 31) // 32
        print('tn_qr %5d x %4d (%2d panels)  chain %8.1f us  single launch %8.1f us  -> %.1f us saved per panel'
              % (m, n, npan, res['0'], res['1'], (res['0'] - res['1']) / npan), flush=True)


if __name__ == '__main__':
    main()
