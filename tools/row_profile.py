import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import tnac4o_amd
from tnac4o_amd import mps
from tnac4o_amd.auxx import synthetic_chimera
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
s._setup_rhoT(**kw); torch.cuda.synchronize()
T = dict(mpo=0.0, copy=0.0, call=0.0, total=0.0)
t_all = time.perf_counter()
s.rhoT = [None] * (n + 1); s.rhoT_overlap = [1] * (n + 1); s.rhoT_discarded = [0] * (n + 1)
s.rhoT[n] = mps.MPS(d=1, L=n, Dmax=1, initial='X')
for ny in range(n - 1, -1, -1):
    t0 = time.perf_counter(); M = s._row_mpo(ny); t1 = time.perf_counter()
    psi = s.rhoT[ny + 1].copy(); t2 = time.perf_counter()
    ov = psi.apply_mpo_compress(M, Hconj=True, **kw); t3 = time.perf_counter()
    s.rhoT[ny] = psi
    T['mpo'] += t1 - t0; T['copy'] += t2 - t1; T['call'] += t3 - t2
torch.cuda.synchronize()
T['total'] = time.perf_counter() - t_all
print({k: round(1e3 * v, 1) for k, v in T.items()})
# inside the call: time in the library vs wrapper
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
s._setup_rhoT(**kw); torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(18)
