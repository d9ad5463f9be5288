"""Throughput experiment: G independent sweeps (lattice rotations) driven by G Python threads on G HIP streams."""
import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
chi = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = {128: 4, 512: 8, 2048: 16}[L]
J = synthetic_chimera(n, n, 20260004)
kw = dict(graduate_truncation=True, Dmax=chi, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
make(0)._setup_rhoT(**dict(kw, Dmax=8))      # warm the library
torch.cuda.synchronize()
for G in (4, 8):
    solvers = [make(r % 4) for r in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    res = [None] * G
    def work(i):
        with torch.cuda.stream(streams[i]):
            solvers[i]._setup_rhoT(**kw)
            streams[i].synchronize()
            res[i] = min(solvers[i].rhoT_overlap)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i,)) for i in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('G=%d  wall %.2f s  -> %.2f s/sweep   overlaps %s' % (G, dt, dt / G, ['%.12f' % r for r in res]), flush=True)
