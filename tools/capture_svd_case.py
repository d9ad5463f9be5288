"""Capture the centre matrices of the headline sweep on which the block-Jacobi SVD needs the most sweeps (for offline
convergence studies).  Writes gpurun_out/svd_case_<rank>.npy (at most 3 matrices) and prints the sweep histogram."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tnac4o_amd
from tnac4o_amd import ops
from tnac4o_amd.auxx import synthetic_chimera

orig = ops.svd_trunc
cases = []


def spy(Cm, Dmax, tol):
    keepC = Cm.clone()
    out = orig(Cm, Dmax, tol)
    cases.append((out[5]['sweeps'], tuple(Cm.shape), int(Dmax), float(tol), out[3], keepC if out[5]['sweeps'] >= 8 else None))
    return out


ops.svd_trunc = spy
s = tnac4o_amd.tnac4o(mode='Ising', Nx=16, Ny=16, Nc=8, J=synthetic_chimera(16, 16, 20260004), beta=3.0)
s._setup_rhoT(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
hist = {}
for c in cases:
    hist[c[0]] = hist.get(c[0], 0) + 1
print('sweep histogram', dict(sorted(hist.items())))
big = sorted([c for c in cases if c[5] is not None], key=lambda c: -c[0])[:3]
os.makedirs('gpurun_out', exist_ok=True)
for i, c in enumerate(big):
    np.save('gpurun_out/svd_case_%d.npy' % i, c[5].cpu().numpy())
    print('case', i, 'sweeps', c[0], 'shape', c[1], 'Dmax', c[2], 'tol', c[3], 'keep', c[4])
