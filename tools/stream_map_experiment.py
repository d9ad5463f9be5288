"""Which torch streams interleave well?  Runs 2 single-chain sweeps concurrently on stream pairs (i, j) drawn from a pool
of freshly created streams and prints the wall time per pair (diagnosis of HIP stream -> hardware-queue mapping)."""
import os, sys, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import tnac4o_amd
from tnac4o_amd.auxx import synthetic_chimera
n = 8      # L = 512 keeps the experiment short
J = synthetic_chimera(n, n, 20260003)
kw = dict(graduate_truncation=True, Dmax=64, tolS=1e-16, tolV=1e-10, max_sweeps=20)
def make(rot):
    s = tnac4o_amd.tnac4o(mode='Ising', Nx=n, Ny=n, Nc=8, J=J, beta=3.0)
    if rot: s.rotate_graph(rot)
    return s
pool = [torch.cuda.Stream() for _ in range(8)]
solvers = [make(r) for r in range(4)]
def run(streams):
    def work(i):
        with torch.cuda.stream(streams[i]):
            solvers[i]._setup_rhoT(**kw)
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(streams))]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize(); return time.perf_counter() - t0
run([pool[0]]); 
print('single', round(run([pool[0]]), 3))
for pair in [(0, 1), (0, 2), (0, 3), (0, 4), (1, 2), (1, 3), (2, 3)]:
    print('pair', pair, round(run([pool[pair[0]], pool[pair[1]]]), 3))
print('four 0-3', round(run(pool[:4]), 3), ' four 0,2,4,6', round(run([pool[0], pool[2], pool[4], pool[6]]), 3), ' four 4-7', round(run(pool[4:8]), 3))
