"""The panel step of tn_qr on its own (tn_panel_orth): iterated Cholesky-QR (method 0) against the Householder TSQR (method 1)
on well-conditioned, graded, nearly dependent, rank-deficient and badly scaled panels: orthonormality, span residual, passes,
time per panel."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tnac4o_amd import ops
from tnac4o_amd._lib import lib, check


panel_orth = ops.panel_orth


def quality(X, Y):
    b = X.shape[1]
    G = Y.t() @ Y
    orth = (G - torch.eye(b, dtype=torch.float64, device=X.device)).abs().max().item()
    Res = X - Y @ (Y.t() @ X)
    cn = X.norm(dim=0)
    res = (Res.norm(dim=0) / torch.where(cn > 0, cn, torch.ones_like(cn))).max().item()
    return orth, res


def cases(dev):
    g = torch.Generator(device='cpu').manual_seed(7)
    rn = lambda *s: torch.randn(*s, dtype=torch.float64, generator=g).to(dev)
    out = []
    out.append(('randn 16384x32 row-major', rn(16384, 32)))
    out.append(('randn 16384x32 col-major', rn(32, 16384).t()))
    out.append(('randn 4096x32', rn(4096, 32)))
    out.append(('randn 300x32 (ragged)', rn(300, 32)))
    out.append(('randn 40x32', rn(40, 32)))
    out.append(('randn 32x32', rn(32, 32)))
    out.append(('randn 1000x17 (narrow)', rn(1000, 17)))
    out.append(('randn 5000x1', rn(5000, 1)))
    X = rn(16384, 32) * torch.logspace(0, -30, 32, dtype=torch.float64, device=dev)[None, :]
    out.append(('graded columns 1e0..1e-30', X))
    for kappa in (1e3, 1e6, 1e9, 1e12, 1e15):
        U, _ = torch.linalg.qr(rn(8192, 32))
        V, _ = torch.linalg.qr(rn(32, 32))
        S = torch.logspace(0, -torch.log10(torch.tensor(kappa)).item(), 32, dtype=torch.float64, device=dev)
        out.append(('kappa %.0e' % kappa, (U * S[None, :]) @ V.t()))
    X = rn(8192, 32)
    X[:, 5] = 0.0
    X[:, 9] = X[:, 2]
    X[:, 20] = 2.0 * X[:, 3]
    X[:, 31] = X[:, 0] + 1e-13 * X[:, 31]
    out.append(('zero / duplicate / dependent columns', X))
    out.append(('all zero', torch.zeros(2048, 32, dtype=torch.float64, device=dev)))
    out.append(('rank 3 of 32', rn(4096, 3) @ rn(3, 32)))
    out.append(('scaled 1e-200', rn(4096, 32) * 1e-200))
    out.append(('scaled 1e+200', rn(4096, 32) * 1e200))
    X = rn(16384, 32)
    X[:256] *= 1e150
    out.append(('one block 1e150 larger', X))
    # Krylov-like: columns converge to the dominant direction
    A = rn(2048, 2048) / 45.0
    v = rn(2048, 1)
    cols = []
    for _ in range(32):
        v = A @ v
        cols.append(v / v.norm())
    out.append(('Krylov 2048x32', torch.cat(cols, 1)))
    return out


def main():
    dev = 'cuda'
    worst = 0.0
    for name, X in cases(dev):
        Xc = X.clone()
        Y, st, devh = panel_orth(X, 0, state=True)
        assert torch.equal(X, Xc), 'input modified'
        o0, r0 = quality(X, Y)
        Y1 = panel_orth(X, 1)
        o1, r1 = quality(X, Y1)
        print('%-40s cholqr orth %.1e res %.1e passes %d defer %d refill %d fallback %d dev %s | tsqr orth %.1e res %.1e'
              % (name, o0, r0, st[3], st[6], st[7], st[8], ' '.join('%.0e' % d for d in devh[:st[3] + 1]), o1, r1), flush=True)
        worst = max(worst, o0, r0 if 'zero' not in name else 0.0)
    print('worst', worst)
    # timing
    for (n, b) in [(16384, 32), (8192, 32), (4096, 32), (1024, 32), (256, 32)]:
        X = torch.randn(n, b, dtype=torch.float64, device=dev)
        Y = torch.empty_like(X)
        for method in (0, 1):
            for _ in range(3):
                panel_orth(X, method, out=Y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                panel_orth(X, method, out=Y)
            torch.cuda.synchronize()
            print('%6d x %2d  %s  %.1f us / panel' % (n, b, ('cholqr', 'tsqr')[method], 1e6 * (time.perf_counter() - t0) / 200), flush=True)
    st = (C.c_uint64 * 8)()
    check(lib().tn_panel_stats(st, 0))
    print('stats', list(st))


if __name__ == '__main__':
    main()
