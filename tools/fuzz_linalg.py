"""Randomised robustness run of tn_qr / tn_svd_trunc / tn_svdvals against numpy (not part of the test suite: prints the
worst residuals over many random shapes and structures, exits non-zero when a tolerance is exceeded)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import torch
from tnac4o_amd import ops

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = dict(qr_res=0.0, qr_orth=0.0, svd_S=0.0, svd_res=0.0, svd_orth=0.0, vals=0.0)
bad = []


def make(m, n, kind):
    A = rng.standard_normal((m, n))
    if kind == 'graded':
        A = A * (10.0 ** (-rng.uniform(0, 14) * np.arange(n) / max(n - 1, 1)))[None, :]
    elif kind == 'rankdef':
        r = max(1, min(m, n) // 3)
        A = rng.standard_normal((m, r)) @ rng.standard_normal((r, n))
    elif kind == 'zerocols':
        A[:, rng.random(n) < 0.3] = 0.0
    elif kind == 'scaled':
        A = A * 10.0 ** rng.uniform(-120, 120)
    elif kind == 'ones':
        A = np.ones((m, n))
    return A


for case in range(ncase):
    kind = ['plain', 'graded', 'rankdef', 'zerocols', 'scaled', 'ones'][case % 6]
    m = int(rng.integers(1, 700)); n = int(rng.integers(1, 260))
    if case % 7 == 0:
        m, n = int(rng.integers(1000, 5000)), int(rng.integers(33, 200))
    A = make(m, n, kind)
    T = torch.tensor(A, device='cuda')
    view = T if case % 2 == 0 else torch.tensor(np.ascontiguousarray(A.T), device='cuda').t()      # both layouts
    Q, Rm = ops.qr(view.clone() if case % 2 == 0 else view)
    Qh, Rh = Q.cpu().numpy(), Rm.cpu().numpy()
    cn = np.linalg.norm(A, axis=0); cn[cn == 0] = 1.0
    res = (np.abs(Qh @ Rh - A) / cn[None, :]).max() if A.size else 0.0
    orth = np.abs(Qh.T @ Qh - np.eye(Qh.shape[1])).max()
    if not (np.diag(Rh) >= 0).all():
        bad.append(('diagR<0', case, m, n, kind))
    worst['qr_res'] = max(worst['qr_res'], res); worst['qr_orth'] = max(worst['qr_orth'], orth)
    if res > 2e-13 or orth > 1e-13 or not np.isfinite(res + orth):
        bad.append(('qr', case, m, n, kind, res, orth))
    # SVD on a smaller matrix
    k, l = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    C = make(k, l, kind)
    if kind == 'scaled':
        C = C / np.abs(C).max()
    Sref = np.linalg.svd(C, compute_uv=False)
    U, S, Vt, keep, disc, info = ops.svd_trunc(torch.tensor(C, device='cuda'), 10 ** 6, 1e-16)
    Uh, Sh, Vh = U.cpu().numpy(), S.cpu().numpy(), Vt.cpu().numpy()
    s0 = max(Sref[0], 1e-300)
    eS = np.abs(Sh - Sref[:keep]).max() / s0 if keep else 0.0
    eR = np.abs((Uh * Sh) @ Vh - C).max() / s0 if keep else 0.0
    eO = max(np.abs(Uh.T @ Uh - np.eye(keep)).max(), np.abs(Vh @ Vh.T - np.eye(keep)).max()) if keep else 0.0
    worst['svd_S'] = max(worst['svd_S'], eS); worst['svd_res'] = max(worst['svd_res'], eR); worst['svd_orth'] = max(worst['svd_orth'], eO)
    if eS > 2e-13 or eR > 5e-13 or eO > 5e-12:
        bad.append(('svd', case, k, l, kind, eS, eR, eO))
    Sv = ops.svdvals(torch.tensor(C, device='cuda'))
    Sv = Sv.cpu().numpy() if torch.is_tensor(Sv) else np.asarray(Sv)
    eV = np.abs(Sv[:len(Sref)] - Sref).max() / s0
    worst['vals'] = max(worst['vals'], eV)
    if eV > 2e-13:
        bad.append(('svdvals', case, k, l, kind, eV))
print('cases', ncase, 'worst', {k: float('%.3g' % v) for k, v in worst.items()})
for b in bad[:20]:
    print('BAD', b)
sys.exit(1 if bad else 0)
