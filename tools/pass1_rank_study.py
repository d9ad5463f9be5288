"""Experiment (analysis only, torch used for dense linear algebra): how low is the numerical rank of the matrix that the first
canonisation pass factors at a bulk site, (a) as it is, (b) with its left-bond index scaled by the norm of the unfactored
left part (column norms of L = sqrt(diag G_L)), (c) in the exact metric of the left part (the true Schmidt spectrum)?
Decides whether a rank-revealing pass 1 with a diagonal gauge can be made rigorous."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tnac4o_amd
from tnac4o_amd import ops, mps
from tnac4o_amd.auxx import synthetic_chimera

n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)
ny = 8
psi = s.rhoT[ny + 1].copy()
psi.apply_mpo(s._row_mpo(ny), Hconj=True)
T = [a.clone() for a in psi.A]                       # absorbed sites (Dl a, i, Dr b)
# exact left environments G_L(site) = Gram of the part left of `site` w.r.t. its right bond
G = [torch.ones((1, 1), dtype=torch.float64, device='cuda')]
for k in range(n - 1):
    Dl, p, Dr = T[k].shape
    X = (G[-1] @ T[k].reshape(Dl, p * Dr)).reshape(Dl * p, Dr)
    G.append(T[k].reshape(Dl * p, Dr).t() @ X)
    G[-1] = G[-1] / G[-1].abs().max()                # overall scale is irrelevant
# pass 1 down to the site under study, capturing what is factored there
site = 8
cap = {}
orig = ops.site_qr


def spy(side, A, Cm=None, rank_tol=0.0):
    if side == 1 and A.shape[0] == T[site].shape[0] and 'M' not in cap and spy.count == n - 1 - site:
        cap['M'] = ops.mm(A.reshape(-1, A.shape[2]), Cm).reshape(A.shape[0], -1).clone()      # (left bond, i k)
    spy.count += 1
    return orig(side, A, Cm, rank_tol)


spy.count = 0
ops.site_qr = spy
psi.canonise_right()
ops.site_qr = orig
M = cap['M']
print('site', site, 'M shape', tuple(M.shape))
GL = G[site]
d = torch.sqrt(torch.diagonal(GL).clamp_min(0))
sv_plain = torch.linalg.svdvals(M.cpu())
Ms = M * d[:, None]
sv_scaled = torch.linalg.svdvals(Ms.cpu())
lam, V = torch.linalg.eigh(GL.cpu())
lam = lam.clamp_min(0)
S = (V * torch.sqrt(lam)).t()                        # G_L = S^T S
sv_true = torch.linalg.svdvals(S @ M.cpu())


def rank(sv, thr):
    return int((sv > sv[0] * thr).sum())


for name, sv in (('plain', sv_plain), ('diag-scaled', sv_scaled), ('exact metric (Schmidt)', sv_true)):
    print('%-24s rank@2^-56 %4d  rank@1e-20 %4d  rank@1e-12 %4d   sv[0] %.2e' % (name, rank(sv, 2.0 ** -56), rank(sv, 1e-20), rank(sv, 1e-12), sv[0]))
# state norm vs matrix norm in the scaled gauge (what the a-posteriori bound divides by), and the bound ||L'|| <= sqrt(n_cols)
Lp = S / d.cpu()[None, :].clamp_min(1e-300)          # L' = L D^-1 in the S representation (columns of unit norm)
print('||L\'||_2 = %.3f  (bound sqrt(%d) = %.1f)' % (torch.linalg.matrix_norm(Lp, 2), Lp.shape[1], Lp.shape[1] ** 0.5))
N = torch.linalg.matrix_norm(S @ M.cpu())
print('state norm / ||M_scaled||_F = %.3e   state norm / ||M_plain||_F = %.3e' % (N / torch.linalg.matrix_norm(Ms.cpu()), N / torch.linalg.matrix_norm(M.cpu())))
# unpivoted QR of the scaled matrix with columns (left-bond indices) sorted by decreasing scaled norm: residual after k columns
cn = torch.linalg.vector_norm(Ms, dim=1)
order = torch.argsort(cn, descending=True)
B = Ms[order].t().cpu()                               # columns = left-bond indices in sorted order
Q, R = torch.linalg.qr(B)
diag = R.diagonal().abs()
tail = torch.sqrt(torch.flip(torch.cumsum(torch.flip((R ** 2).sum(dim=0) - 0, [0]), 0), [0]))   # not exact residuals; use R blocks
res = []
for k in (64, 128, 192, 256, 320, 384, 448, 512, 640, 768):
    res.append((k, float(torch.linalg.matrix_norm(R[k:, k:])) / float(N)))
print('sorted scaled QR: ||trailing block|| / state norm after k columns:', ['%d: %.1e' % r for r in res])
B2 = M.t().cpu()
Q2, R2 = torch.linalg.qr(B2)
print('plain unsorted QR: ||trailing|| / ||M||:', ['%d: %.1e' % (k, float(torch.linalg.matrix_norm(R2[k:, k:])) / float(torch.linalg.matrix_norm(M.cpu()))) for k in (64, 128, 256, 384, 512, 768)])
