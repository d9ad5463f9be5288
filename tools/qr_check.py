"""tn_qr with the fused Cholesky-QR panel step against the Householder TSQR panel step and numpy: factorisation residual, orthogonality,
|R| agreement; plain / graded / rank-deficient inputs; also with TN_PANEL_MAXPASS=1 (drives the Householder fallback)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tnac4o_amd import ops


def run(T, rank_tol=0.0):
    m, n = T.shape
    k = min(m, n)
    Q = torch.empty((m, k), dtype=torch.float64, device='cuda')
    R = torch.empty((k, n), dtype=torch.float64, device='cuda')
    _, _, keff = ops.qr_into(T.clone(), Q, R, overwrite=True, rank_tol=rank_tol)
    return Q[:, :keff], R[:keff], keff


def check(name, T, rank_tol=0.0):
    out = {}
    for panel in ('chol', 'tsqr'):
        os.environ['TN_PANEL'] = panel
        Q, R, keff = run(T, rank_tol)
        res = ((Q @ R - T).norm(dim=0) / T.norm(dim=0).clamp_min(1e-300)).max().item()
        orth = (Q.t() @ Q - torch.eye(keff, dtype=torch.float64, device='cuda')).abs().max().item()
        out[panel] = (res, orth, keff, R)
    os.environ['TN_PANEL'] = 'chol'
    dR = (out['chol'][3].abs()[:min(out['chol'][2], out['tsqr'][2])] - out['tsqr'][3].abs()[:min(out['chol'][2], out['tsqr'][2])]).abs().max().item() / T.abs().max().item()
    print('%-44s chol res %.1e orth %.1e keff %d | tsqr res %.1e orth %.1e keff %d | d|R| %.1e' %
          (name, out['chol'][0], out['chol'][1], out['chol'][2], out['tsqr'][0], out['tsqr'][1], out['tsqr'][2], dR), flush=True)
    return max(out['chol'][0], out['chol'][1])


g = torch.Generator().manual_seed(11)
rn = lambda *s: torch.randn(*s, dtype=torch.float64, generator=g).cuda()
worst = 0.0
worst = max(worst, check('randn 4096x256', rn(4096, 256)))
worst = max(worst, check('randn 16384x1024 (two-level)', rn(16384, 1024)))
worst = max(worst, check('randn 1000x300 col-major', rn(300, 1000).t()))
worst = max(worst, check('randn 300x1000 (wide)', rn(300, 1000)))
worst = max(worst, check('randn 100x100', rn(100, 100)))
worst = max(worst, check('randn 50x7', rn(50, 7)))
A = rn(4096, 64) @ rn(64, 512)
worst = max(worst, check('rank 64 of 512', A))
worst = max(worst, check('rank 64 of 512, rank_tol 2^-56', A, rank_tol=2.0 ** -56))
A = (rn(8192, 256) * torch.logspace(0, -20, 256, dtype=torch.float64).cuda()[None, :]) @ torch.linalg.qr(rn(256, 256))[0]
worst = max(worst, check('graded spectrum 1e0..1e-20', A))
worst = max(worst, check('graded spectrum, rank_tol 2^-56', A, rank_tol=2.0 ** -56))
print('worst', worst)
assert worst < 1e-12
