import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 16
s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=synthetic_chimera(n, n, 20260004), beta=3.0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for ny in range(n):
        m = s._row_mpo(ny)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print('row MPO build: %.2f ms per row, %.1f ms per sweep' % (1e3 * (t1 - t0) / n, 1e3 * (t1 - t0)))
t0 = time.perf_counter()
for ny in range(n):
    for nx in range(n):
        s._site_tables(ny, nx)
print('host tables only: %.1f ms per sweep' % (1e3 * (time.perf_counter() - t0)))
