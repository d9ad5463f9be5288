"""How much of its error budget does the weighted rank-revealing first pass use?  One sweep of the bench instance: per row the a-posteriori
bound (accepted while <= 2^-56), the bonds before / after the pass, and the peak of the arena."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import mps, ops
from tnac4o_amd.auxx import synthetic_chimera
n, chi = 16, 64
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20, graduate_truncation=True)
orig = ops.compress_mps_native
rows = []
def spy(*a, **k):
    r = orig(*a, **k)
    rows.append(r['info'])
    return r
ops.compress_mps_native = spy
mps.ops.compress_mps_native = spy
s._setup_rhoT(**kw)
for i, inf in enumerate(rows):
    print('row %2d bound %.3e (limit 1.39e-17) bonds %d -> %d arena peak %.2f GB of %.2f' % (n - 1 - i, inf['reveal_error_bound'], inf['bonds_before'], inf['bonds_after'],
          inf['arena_peak'] / 1e9, inf['arena_bytes'] / 1e9))
