"""Host-side profile of the top-down sweep of BASELINE config 5 (RMF 64 x 64, d = 8, chi = 128), single chain: where a row's wall time
goes (library call vs wrapper vs MPO build), cProfile of the sweep, info fields of the chain driver.  Usage: python tools/rmf_row_profile.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile, pstats
import torch
import tnac4o_amd
from tnac4o_amd import mps, ops
from tnac4o_amd.auxx import synthetic_rmf
n = 64
J = synthetic_rmf(n, n, 8, 20260005)
s = tnac4o_amd.tnac4o(mode='RMF', Nx=n, Ny=n, J=J, beta=1.0)
kw = dict(graduate_truncation=True, Dmax=128, tolS=1e-16, tolV=1e-10, max_sweeps=20)
nrows = int(sys.argv[1]) if len(sys.argv) > 1 else 12
s.rhoT = [None] * (n + 1); s.rhoT_overlap = [1] * (n + 1); s.rhoT_discarded = [0] * (n + 1)
s.rhoT[n] = mps.MPS(d=1, L=n, Dmax=1, initial='X')
pr = cProfile.Profile()
for ny in range(n - 1, n - 1 - nrows, -1):
    t0 = time.perf_counter(); M = s._row_mpo(ny); t1 = time.perf_counter()
    psi = s.rhoT[ny + 1].copy(); t2 = time.perf_counter()
    if ny == n - nrows: pr.enable()
    ov = psi.apply_mpo_compress(M, Hconj=True, **kw); torch.cuda.synchronize(); t3 = time.perf_counter()
    if ny == n - nrows: pr.disable()
    s.rhoT[ny] = psi
    info = getattr(psi, '_last_native_info', None)
    print('row %2d: mpo %.1f copy %.1f call %.1f ms  bonds max %d  info %s' % (ny, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), max(a.shape[2] for a in psi.A), info), flush=True)
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
