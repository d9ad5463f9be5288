"""Wall time of every row of one boundary-MPS sweep of the bench instance (single chain): where a sweep's 2 s go by row, with
the bond dimensions the row ends with."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tnac4o_amd
from tnac4o_amd import mps
from tnac4o_amd.auxx import synthetic_chimera

n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw)                                    # warm-up
torch.cuda.synchronize()
orig = mps.MPS.apply_mpo_compress
rows = []


def timed(self, *a, **k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = orig(self, *a, **k)
    torch.cuda.synchronize()
    rows.append((1e3 * (time.perf_counter() - t0), list(self.D)))
    return out


mps.MPS.apply_mpo_compress = timed
t0 = time.perf_counter()
s._setup_rhoT(**kw)
torch.cuda.synchronize()
tot = 1e3 * (time.perf_counter() - t0)
for i, (ms, D) in enumerate(rows):
    print('row %2d (ny = %2d): %7.1f ms   D = %s' % (i, n - 1 - i, ms, D))
print('sum of rows %.1f ms, sweep %.1f ms (the difference: MPO tables, copies, Python between the rows)' % (sum(r[0] for r in rows), tot))
